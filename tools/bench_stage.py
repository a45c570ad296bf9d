"""Transposed-operand staging in the two-stage kernel: registers (stage mode 1, the default) against LDS-DMA (mode 0)
on the step's wgrad / trainable-dgrad shapes.   python tools/bench_stage.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zero_amd import ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# (M, N, K, trans_a, trans_b): wgrad = dy^T x (both operands mn-contiguous), dgrad through a trainable weight (B only)
SHAPES = [(3072, 1024, 4616, 1, 1), (1024, 4096, 4616, 1, 1), (4096, 1024, 4616, 1, 1), (1024, 1024, 4616, 1, 1),
          (3072, 3072, 4616, 1, 1), (4096, 1024, 36928, 1, 1), (1024, 4096, 36928, 1, 1), (4616, 1024, 4096, 0, 1),
          (4616, 4096, 1024, 0, 1), (36928, 1024, 4096, 0, 1)]
for M, N, K, ta, tb in SHAPES:
    a = (torch.randn((K, M) if ta else (M, K), device="cuda") * 0.5).to(torch.bfloat16)
    b = (torch.randn((K, N) if tb else (N, K), device="cuda") * 0.5).to(torch.bfloat16)
    line, outs = f"M,N,K=({M},{N},{K}) tA{ta}tB{tb}:", []
    for mode in (1, 0):
        ops.set_gemm_tile(10 + mode)
        c = ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb))
        outs.append(c)
        t = timeit(lambda: ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb)))
        line += f"  {'regs' if mode else 'dma '} {t:7.1f} us ({2.0 * M * N * K / t / 1e6:6.0f} TF) plan {ops.lib.zk_gemm_last_plan()}"
    ops.set_gemm_tile(11)
    print(line + ("  bit-equal" if torch.equal(outs[0], outs[1]) else "  DIFFERENT"), flush=True)
