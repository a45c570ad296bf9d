"""Persistent GEMM: band height of the tile order (XCD L2 footprint) A/B on the decoder shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zero_amd import ops

SHAPES = [(6144, 3072, 8192, 4), (6144, 3072, 16384, 4), (6144, 3072, 3072, 4), (6144, 16384, 3072, 2), (6144, 8192, 3072, 2),
          (6144, 5120, 3072, 2), (1024, 128256, 3072, 2), (36928, 4096, 1024, 2)]
for m, n, k, tile in SHAPES:
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    b = torch.randn(n, k, device="cuda").to(torch.bfloat16)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    ops.set_gemm_tile(tile)
    line = f"M={m} N={n} K={k} tile{tile}:"
    for rounds in range(2):
        for band in (1, 2, 4, 8, 16):
            ops.set_gemm_tile(60 + band)
            for _ in range(3):
                ops.gemm(a, b, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm(a, b, out=out)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            line += f"  b{band}: {ms*1e3:5.0f}us {2.0*m*n*k/ms/1e9:6.0f}TF"
        line += " |"
    print(line, flush=True)
ops.set_gemm_tile(60); ops.set_gemm_tile(-1)
