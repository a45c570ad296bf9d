"""Where a persistent-ring workgroup's time goes (diagnostic build: gemm.hip with -DZK_RING_STAMPS linked into
zero_amd/libzero_hip_stamps.so; run with ZERO_AMD_LIB pointing at it).  Per product: per tile round the mean K-loop and
epilogue time of the waves, the spread of the epilogue start over the workgroups, and the launch span."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zero_amd import ops

M, K, F = 6144, 3072, 8192
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
wi = (torch.randn(2 * F, K, device="cuda") * 0.05).to(torch.bfloat16)
dy = torch.randn(M, K, device="cuda").to(torch.bfloat16)
wdt = (torch.randn(F, K, device="cuda") * 0.05).to(torch.bfloat16)
wq = (torch.randn(5120, K, device="cuda") * 0.05).to(torch.bfloat16)
h = torch.randn(M, F, device="cuda").to(torch.bfloat16)
wd = (torch.randn(K, F, device="cuda") * 0.05).to(torch.bfloat16)
res = torch.randn(M, K, device="cuda").to(torch.bfloat16)
gu, _ = ops.gemm_swiglu(x, wi)
CASES = (("gate|up fused swiglu (256x256, 6 rounds)", lambda: ops.gemm_swiglu(x, wi)),
         ("down-dgrad fused swiglu_bwd (256x256, 3 rounds)", lambda: ops.gemm_swiglu_bwd(dy, wdt, gu)),
         ("gate|up plain (256x256, 6 rounds)", lambda: ops.gemm(x, wi)),
         ("qkv 6144x5120x3072 (256x256, 1.9 rounds)", lambda: ops.gemm(x, wq)),
         ("down + residual 6144x3072x8192 (384x192, 1 round)", lambda: ops.gemm(h, wd, residual=res)),
         ("d_gu 6144x3072x16384 (384x192, 1 round)", lambda: ops.gemm(gu, wi.t().contiguous().view(K, 2 * F))))
buf = np.zeros(256 * 8 * 8 * 4, dtype=np.uint64)
for name, fn in CASES:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    assert ops.lib.zk_debug_clear_ring_stamps() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    assert ops.lib.zk_debug_read_ring_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    st = buf.reshape(256, 8, 8, 4).astype(np.float64) / 100.0      # us
    live = st[:, :, 0, 0] > 0
    t0 = st[:, :, :, 0][st[:, :, :, 0] > 0].min()
    print(f"{name}: launch {e0.elapsed_time(e1) * 1e3:.0f} us (events), span by stamps {st[:, :, :, 2].max() - t0:.0f} us")
    for o in range(8):
        m = live[:, o]
        if not m.any():
            break
        s = st[m, o]                                               # [wg, wave, stamp]
        kl = (s[:, :, 1] - s[:, :, 0]).mean()
        ep = (s[:, :, 2] - s[:, :, 1])
        nxt = ""
        if o + 1 < 8 and live[:, o + 1].any():
            m2 = m & live[:, o + 1]
            gap = (st[m2, o + 1, :, 0] - st[m2, o, :, 2]).mean()
            nxt = f"  -> next tile start +{gap:.2f}"
        print(f"  tile {o}: {int(m.sum()):3d} workgroups  start {s[:, :, 0].mean() - t0:7.1f} (+-{s[:, 0, 0].std():.1f})  K loop {kl:6.2f}  "
              f"epilogue mean {ep.mean():5.2f} / slowest wave {ep.max(1).mean():5.2f} / fastest {ep.min(1).mean():5.2f}{nxt}")
