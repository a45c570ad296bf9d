"""Ring-kernel ablation: full / no-MFMA (DMA + barriers only) / no-DMA (LDS reads + MFMA only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zero_amd import ops

SHAPES = ((6144, 16384, 3072), (6144, 5120, 3072), (6144, 3072, 8192), (4096, 4096, 4096), (36928, 4096, 1024), (4616, 4096, 1024))
for (m, n, k) in SHAPES:
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    b = torch.randn(n, k, device="cuda").to(torch.bfloat16)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for tile, pf in ((2, 0), (4, 0), (2, 1), (4, 1)):   # pf = 1: persistent ring (only modes 0 and 3 exist there)
        ops.set_gemm_tile(101); ops.set_gemm_tile(21); ops.set_gemm_tile(tile); ops.set_gemm_tile(50 + pf)
        line = f"M={m} N={n} K={k} tile{tile} persistent{pf}:"
        for mode in ((0, 3) if pf else (0, 1, 2, 3)):
            ops.set_gemm_tile(30 + mode)
            for _ in range(3):
                ops.gemm(a, b, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm(a, b, out=out)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            line += f"  mode{mode}: {ms*1e3:6.0f} us ({2.0*m*n*k/ms/1e9:6.0f} TF-equiv)"
        print(line, flush=True)
ops.set_gemm_tile(30)
ops.set_gemm_tile(51)
