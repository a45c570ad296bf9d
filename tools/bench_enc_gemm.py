"""Encoder / adapter GEMMs at B = 8 (M = 4616) WITH their epilogues, per forced tile config."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zero_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4616
dev = "cuda"
r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
x1, x4 = r(M, 1024), r(M, 4096)
res1 = r(M, 1024)
w_fc1, b_fc1 = r(4096, 1024), r(4096)
w_fc2, b_fc2 = r(1024, 4096), r(1024)
w_qkv, b_qkv = r(3072, 1024), r(3072)
w_o, b_o = r(1024, 1024), r(1024)
pre = torch.empty(M, 4096, device=dev, dtype=torch.bfloat16)
out4 = torch.empty(M, 4096, device=dev, dtype=torch.bfloat16)
out1 = torch.empty(M, 1024, device=dev, dtype=torch.bfloat16)
out3 = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16)
dy4, dy3 = r(M, 4096), r(M, 3072)
gw = torch.empty(4096, 1024, device=dev, dtype=torch.bfloat16)
CASES = [
    ("fc1 fwd gelu+aux", 2.0 * M * 4096 * 1024, lambda: ops.gemm(x1, w_fc1, bias=b_fc1, epilogue=ops.EPI_QUICK_GELU, aux_out=pre, out=out4)),
    ("fc1 fwd gelu", 2.0 * M * 4096 * 1024, lambda: ops.gemm(x1, w_fc1, bias=b_fc1, epilogue=ops.EPI_QUICK_GELU, out=out4)),
    ("fc2 fwd bias+res", 2.0 * M * 4096 * 1024, lambda: ops.gemm(x4, w_fc2, bias=b_fc2, residual=res1, out=out1)),
    ("qkv fwd bias", 2.0 * M * 3072 * 1024, lambda: ops.gemm(x1, w_qkv, bias=b_qkv, out=out3)),
    ("o fwd bias+res", 2.0 * M * 1024 * 1024, lambda: ops.gemm(x1, w_o, bias=b_o, residual=res1, out=out1)),
    ("fc2 dgrad *dgelu (tB)", 2.0 * M * 4096 * 1024, lambda: ops.gemm(res1, w_fc2, trans_b=True, epilogue=ops.EPI_MUL_DQUICK_GELU, aux_in=pre, out=out4)),
    ("fc1 dgrad (tB)", 2.0 * M * 4096 * 1024, lambda: ops.gemm(dy4, w_fc1, trans_b=True, out=out1)),
    ("qkv dgrad (tB)", 2.0 * M * 3072 * 1024, lambda: ops.gemm(dy3, w_qkv, trans_b=True, out=out1)),
    ("fc1 wgrad (tA tB)", 2.0 * M * 4096 * 1024, lambda: ops.gemm(dy4, x1, trans_a=True, trans_b=True, out=gw)),
    ("fc2 wgrad (tA tB)", 2.0 * M * 4096 * 1024, lambda: ops.gemm(res1, x4, trans_a=True, trans_b=True, out=w_fc2.clone())),
]


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, fl, fn in CASES:
    line = f"{name:24s}:"
    for cfg in (-1, 0, 1, 2, 3, 4):
        ops.set_gemm_tile(cfg)
        try:
            us = timeit(fn)
            line += f"  t{cfg:2d}: {us:5.0f}us {fl / us / 1e6:5.0f}TF"
        except Exception as e:
            line += f"  t{cfg:2d}: n/a"
    ops.set_gemm_tile(101)
    ops.set_gemm_tile(-1)
    line += f"  nosplit: {timeit(fn):5.0f}us"
    ops.set_gemm_tile(100)
    print(line, flush=True)
