"""A/B of two plan settings of the persistent ring on the step's large NT products (default tile choice), results
compared bit for bit.   python tools/bench_pring_ab.py [codeA codeB]     (on the GPU box; default 51 52)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zero_amd import ops

CODES = [int(x) for x in sys.argv[1:3]] or [51, 52]
SHAPES = [("dec qkv", 6144, 5120, 3072), ("dec o", 6144, 3072, 3072), ("dec gateup", 6144, 16384, 3072),
          ("dec down", 6144, 3072, 8192), ("dec d_down", 6144, 8192, 3072), ("dec d_gu", 6144, 3072, 16384),
          ("dec d_qkv", 6144, 3072, 5120), ("c4 qkv", 6144, 6144, 4096), ("c4 gateup", 6144, 28672, 4096),
          ("square 8k", 8192, 8192, 8192), ("c2 fc1", 36864, 4096, 1024), ("lm head", 1024, 128256, 3072)]


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    torch.manual_seed(0)
    for name, m, n, k in SHAPES:
        a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        b = (torch.randn(n, k, device="cuda") * 0.05).to(torch.bfloat16)
        outs, line = [], f"{name:11s} M={m:6d} N={n:6d} K={k:6d}:"
        for rep in range(2):                       # A B A B: box drift shows as a difference between the repeats
            for code in CODES:
                ops.set_gemm_tile(code)
                out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
                ops.gemm(a, b, out=out)
                plan = ops.lib.zk_gemm_last_plan()
                ms = timed(lambda: ops.gemm(a, b, out=out))
                outs.append(out)
                line += f"  {code}[{plan}] {2.0 * m * n * k / ms / 1e9:6.0f} TF {ms * 1e3:5.0f} us"
        line += f"  same={torch.equal(outs[0], outs[1])}"
        if n % 64 == 0 and n >= 8192:              # fused SwiGLU epilogues (interleaved gate|up)
            res = []
            wd = b[: n // 2].contiguous()          # stands in for down_proj^T [F, d]
            for code in CODES:
                ops.set_gemm_tile(code)
                gu, h = ops.gemm_swiglu(a, b)
                ms = timed(lambda: ops.gemm_swiglu(a, b))
                dgu = ops.gemm_swiglu_bwd(a, wd, gu)
                msb = timed(lambda: ops.gemm_swiglu_bwd(a, wd, gu))
                res.append((gu, h, dgu))
                line += f"  swiglu {code}: {ms * 1e3:5.0f} / bwd {msb * 1e3:5.0f} us"
            line += f"  same={all(torch.equal(x, y) for x, y in zip(res[0], res[1]))}"
        print(line, flush=True)
    ops.set_gemm_tile(51)


if __name__ == "__main__":
    main()
