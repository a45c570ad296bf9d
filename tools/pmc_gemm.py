"""One GEMM shape, one tile config, a few launches: the target of a `rocprofv3 --pmc` pass.
    python tools/pmc_gemm.py M N K tile [ring]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zero_amd import ops

m, n, k, tile = (int(x) for x in sys.argv[1:5])
ring = int(sys.argv[5]) if len(sys.argv) > 5 else 1
a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
b = torch.randn(n, k, device="cuda").to(torch.bfloat16)
out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
ops.set_gemm_tile(101)
ops.set_gemm_tile(20 + ring)
ops.set_gemm_tile(tile)
for _ in range(5):
    ops.gemm(a, b, out=out)
torch.cuda.synchronize()
