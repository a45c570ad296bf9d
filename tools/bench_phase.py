"""Experiment: do in-phase epilogue bursts cost time?  Stagger the persistent workgroups at start by
d, 2d, 3d (4 phases) and compare the launch time with the unstaggered one + 3d."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zero_amd import ops

M, K, F = 6144, 3072, 8192
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
gw = (torch.randn(F, K, device="cuda") * 0.05).to(torch.bfloat16)
uw = (torch.randn(F, K, device="cuda") * 0.05).to(torch.bfloat16)
wi = ops.interleave_gate_up(gw, uw)
dy = torch.randn(M, K, device="cuda").to(torch.bfloat16)
wdt = (torch.randn(F, K, device="cuda") * 0.05).to(torch.bfloat16)
gu, h = ops.gemm_swiglu(x, wi)
wplain = torch.cat((gw, uw), 0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


xc = torch.randn(36928, 1024, device="cuda").to(torch.bfloat16)
w1 = (torch.randn(4096, 1024, device="cuda") * 0.05).to(torch.bfloat16)
b1 = torch.randn(4096, device="cuda").to(torch.bfloat16)
pre = torch.empty(36928, 4096, device="cuda", dtype=torch.bfloat16)
act = torch.empty(36928, 4096, device="cuda", dtype=torch.bfloat16)
hn = torch.randn(1024, 3072, device="cuda").to(torch.bfloat16)
lm = (torch.randn(128256, 3072, device="cuda") * 0.05).to(torch.bfloat16)
wq = (torch.randn(5120, 3072, device="cuda") * 0.05).to(torch.bfloat16)
CASES = (("gate|up fused swiglu", lambda: ops.gemm_swiglu(x, wi)),
         ("down-dgrad fused swiglu_bwd", lambda: ops.gemm_swiglu_bwd(dy, wdt, gu)),
         ("gate|up plain", lambda: ops.gemm(x, wplain)),
         ("qkv 6144x5120x3072", lambda: ops.gemm(x, wq)),
         ("lm_head 1024x128256x3072", lambda: ops.gemm(hn, lm)),
         ("c2 fc1 quick_gelu+aux", lambda: ops.gemm(xc, w1, bias=b1, epilogue=ops.EPI_QUICK_GELU, aux_out=pre, out=act)))
for name, fn in CASES:
    line = f"{name:28s}:"
    for nph in (4, 8):
        for us_ in (0, 1, 2, 3, 4, 5, 6, 8):
            if nph == 8 and us_ == 0:
                continue
            ops.set_gemm_tile(100000 + (nph << 20) + us_ * 100)
            line += f"  p{nph} d{us_}: {timeit(fn):5.0f}"
    print(line, flush=True)
ops.set_gemm_tile(99999)
