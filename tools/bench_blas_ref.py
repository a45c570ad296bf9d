"""What the vendor library reaches on the step's GEMM shapes (torch.matmul -> hipBLASLt / rocBLAS), next to
zk_gemm_bf16: a measured practical ceiling for the kernels, not part of the product path.
    python tools/bench_blas_ref.py            (on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zero_amd import ops

SHAPES = [("dec qkv", 6144, 5120, 3072), ("dec o", 6144, 3072, 3072), ("dec gateup", 6144, 16384, 3072),
          ("dec down", 6144, 3072, 8192), ("dec d_down", 6144, 8192, 3072), ("dec d_gu", 6144, 3072, 16384),
          ("enc fc1", 4616, 4096, 1024), ("enc fc2", 4616, 1024, 4096), ("enc qkv", 4616, 3072, 1024),
          ("c4 o", 6144, 4096, 4096), ("c4 down", 6144, 4096, 14336), ("c4 d_gu", 6144, 4096, 28672),
          ("c2 fc1", 36928, 4096, 1024), ("square 8k", 8192, 8192, 8192)]


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
        b = torch.randn(n, k, device="cuda").to(torch.bfloat16)
        out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        bt = b.t()
        t_blas = timed(lambda: torch.matmul(a, bt, out=out))
        t_zk = timed(lambda: ops.gemm(a, b, out=out))
        f = 2.0 * m * n * k / 1e9
        print(f"{name:11s} M={m:6d} N={n:6d} K={k:6d}: vendor {f / t_blas:7.1f} TF ({t_blas * 1e3:6.0f} us)   "
              f"zk {f / t_zk:7.1f} TF ({t_zk * 1e3:6.0f} us)   zk/vendor time {t_zk / t_blas:.2f}", flush=True)


if __name__ == "__main__":
    main()
