"""Split-K sweep on the small products of the step (two-stage 128x128 kernel + splitk_finish): microseconds per
forced split count, and what the library's own choice gives.   python tools/bench_splitk.py   (on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zero_amd import ops

SHAPES = [  # (name, M, N, K, trans_a, trans_b)
    ("c3 wg fc1", 4096, 1024, 4616, True, True), ("c3 wg fc2", 1024, 4096, 4616, True, True),
    ("c3 wg qkv", 3072, 1024, 4616, True, True), ("c3 wg o", 1024, 1024, 4616, True, True),
    ("c3 wg ad2", 3072, 3072, 4608, True, True), ("c3 fc2", 4616, 1024, 4096, False, False),
    ("c3 o", 4616, 1024, 1024, False, False), ("c3 dg fc1", 4616, 1024, 4096, False, True),
    ("c3 dg qkv", 4616, 1024, 3072, False, True), ("c3 dg o", 4616, 1024, 1024, False, True),
    ("c4 wg fc1", 4096, 1024, 584, True, True), ("c4 fc2", 584, 1024, 4096, False, False),
    ("c4 dg fc1", 584, 1024, 4096, False, True),
    ("c2 wg fc1", 4096, 1024, 36928, True, True), ("c2 wg o", 1024, 1024, 36928, True, True),
    ("c2 wg qkv", 3072, 1024, 36928, True, True), ("c2 wg ad2", 3072, 3072, 36864, True, True),
    ("lm dgrad", 1024, 3072, 128256, False, True),
]


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
    for name, m, n, k, ta, tb in SHAPES:
        a = torch.randn((k, m) if ta else (m, k), device="cuda").to(torch.bfloat16)
        b = torch.randn((k, n) if tb else (n, k), device="cuda").to(torch.bfloat16)
        out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        line = f"{name:10s} M={m:5d} N={n:5d} K={k:6d}:"
        ops.set_gemm_tile(100)
        ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out)
        plan = ops.lib.zk_gemm_last_plan()
        line += f"  auto[{plan}] {timed(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out)) * 1e3:5.0f}"
        for sp in (1, 2, 3, 4, 5, 6, 8, 12, 16):
            ops.set_gemm_tile(100 + sp)
            ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out)
            got = ops.lib.zk_gemm_last_plan()
            ms = timed(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out))
            line += f"  s{sp}[{got % 100 if got >= 4000 else 1}] {ms * 1e3:5.0f}"
        print(line, flush=True)
    ops.set_gemm_tile(100)


if __name__ == "__main__":
    main()
