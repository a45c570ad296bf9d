"""GEMM micro-benchmark on the shapes of the C3 step (random bf16 data), per tile config.
    python tools/bench_gemm.py            (on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zero_amd import ops

M = 6144
SHAPES = [  # (name, M, N, K, trans_a, trans_b)
    ("dec qkv    NT", M, 5120, 3072, False, False),
    ("dec o      NT", M, 3072, 3072, False, False),
    ("dec gateup NT", M, 16384, 3072, False, False),
    ("dec down   NT", M, 3072, 8192, False, False),
    ("dec d_down NT", M, 8192, 3072, False, False),
    ("dec d_gu   NT", M, 3072, 16384, False, False),
    ("dec d_qkv  NT", M, 3072, 5120, False, False),
    ("dec d_down NN", M, 8192, 3072, False, True),
    ("dec d_gu   NN", M, 3072, 16384, False, True),
    ("dec d_qkv  NN", M, 3072, 5120, False, True),
    ("enc qkv    NT", 4616, 3072, 1024, False, False),
    ("enc fc1    NT", 4616, 4096, 1024, False, False),
    ("enc fc2    NT", 4616, 1024, 4096, False, False),
    ("enc wg fc1 TN", 4096, 1024, 4616, True, True),
    ("enc wg qkv TN", 3072, 1024, 4616, True, True),
    ("enc wg o   TN", 1024, 1024, 4616, True, True),
    ("enc o      NT", 4616, 1024, 1024, False, False),
    ("ad fc1     NT", 4608, 3072, 1024, False, False),
    ("ad fc2     NT", 4608, 3072, 3072, False, False),
    ("c4 qkv     NT", M, 6144, 4096, False, False),
    ("c4 o       NT", M, 4096, 4096, False, False),
    ("c4 gateup  NT", M, 28672, 4096, False, False),
    ("c4 down    NT", M, 4096, 14336, False, False),
    ("c4 d_down  NT", M, 14336, 4096, False, False),
    ("c4 d_gu    NT", M, 4096, 28672, False, False),
    ("c4 d_qkv   NT", M, 4096, 6144, False, False),
    ("c4 enc qkv NT", 584, 3072, 1024, False, False),
    ("c4 enc fc1 NT", 584, 4096, 1024, False, False),
    ("lm head    NT", 1024, 128256, 3072, False, False),
    ("lm dgrad   NN", 1024, 3072, 128256, False, True),
    ("square 4k  NT", 4096, 4096, 4096, False, False),
    ("square 8k  NT", 8192, 8192, 8192, False, False),
    ("c2 fc1     NT", 36928, 4096, 1024, False, False),
    ("c2 fc2     NT", 36928, 1024, 4096, False, False),
    ("c2 qkv     NT", 36928, 3072, 1024, False, False),
    ("c2 o       NT", 36928, 1024, 1024, False, False),
    ("c2 d_qkv   NT", 36928, 1024, 3072, False, False),
]


def main():
    torch.manual_seed(0)
    only = os.environ.get("ZERO_AMD_GEMM_FILTER")      # comma-separated substrings of shape names
    for name, m, n, k, ta, tb in SHAPES:
        if only and not any(f in name for f in only.split(",")):
            continue
        a = torch.randn((k, m) if ta else (m, k), device="cuda").to(torch.bfloat16)
        b = torch.randn((k, n) if tb else (n, k), device="cuda").to(torch.bfloat16)
        out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        line = f"{name}  M={m:6d} N={n:6d} K={k:6d} :"
        for cfg, stg, pr in ((0, 11, 1), (1, 11, 1), (2, 11, 0), (2, 11, 1), (3, 11, 1), (4, 11, 0), (4, 11, 1), (5, 11, 1), (-1, 11, 1)):
            if (stg == 22 or cfg in (4, 5) or pr == 0) and (ta or tb):
                continue
            ops.set_gemm_tile(21)
            ops.set_gemm_tile(stg)
            ops.set_gemm_tile(50 + pr)      # 51 = persistent ring (default), 50 = one workgroup per tile
            ops.set_gemm_tile(cfg)
            for _ in range(3):
                ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 20
            e0.record()
            for _ in range(iters):
                ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            line += f"  t{cfg:2d}{'p' if pr else 'r'}: {2.0 * m * n * k / ms / 1e9:7.1f} TF ({ms * 1e3:6.0f} us)"
        ops.set_gemm_tile(101)
        for _ in range(3):
            ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        line += f"  nosplit: {2.0 * m * n * k / ms / 1e9:7.1f} TF ({ms * 1e3:6.0f} us)"
        ops.set_gemm_tile(100)
        print(line, flush=True)
    ops.set_gemm_tile(-1)
    ops.set_gemm_tile(11)
    ops.set_gemm_tile(21)
    ops.set_gemm_tile(51)


if __name__ == "__main__":
    main()
