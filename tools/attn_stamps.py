"""Where a forward-attention workgroup's time goes (diagnostic build: make the library with -DZK_ATTN_STAMPS on
attention.hip and point ZERO_AMD_LIB at it).  Prints, per causal query block, the mean microseconds between the stamps
entry -> first K/V tile landed -> loop end -> stores issued -> stores drained, and the launch's span."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zero_amd import ops

B, S, Hq, Hkv, hd, causal = 8, 768, 24, 8, 128, True
if len(sys.argv) > 1 and sys.argv[1] == "enc":
    B, S, Hq, Hkv, hd, causal = 8, 577, 16, 16, 64, False
torch.manual_seed(0)
qkv = torch.randn(B * S, (Hq + 2 * Hkv) * hd, device="cuda").to(torch.bfloat16)
q, k, v = qkv[:, :Hq * hd], qkv[:, Hq * hd:(Hq + Hkv) * hd], qkv[:, (Hq + Hkv) * hd:]
for _ in range(5):
    o, lse = ops.attn_fwd(q, k, v, B, S, Hq, Hkv, hd, hd ** -0.5, causal)
torch.cuda.synchronize()
nblk = ((S + 127) // 128) * Hq * B
buf = np.zeros(nblk * 8, dtype=np.uint64)
assert ops.lib.zk_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), nblk) == 0
st = buf.reshape(nblk, 8).astype(np.float64)
t = st[:, :5] / 100.0          # 100 MHz -> microseconds
t0 = t[:, 0].min()
print(f"launch span {t[:, 4].max() - t0:.1f} us, {nblk} workgroups; last entry at {t[:, 0].max() - t0:.1f} us")
for qb in sorted(set(st[:, 5].astype(int))):
    m = st[:, 5].astype(int) == qb
    d = np.diff(t[m], axis=1).mean(0)
    print(f"qblk {qb} (tiles {int(st[m, 6][0])}): start {np.mean(t[m, 0] - t0):6.1f}  prologue {d[0]:5.2f}  loop {d[1]:6.2f} "
          f"({d[1] / st[m, 6][0]:.2f}/tile)  store issue {d[2]:5.2f}  drain {d[3]:5.2f}  total {np.mean(t[m, 4] - t[m, 0]):6.2f} us")

# ---- backward kernels (same build): dQ blocks are 128 queries (heaviest = last first), dK/dV blocks 128 keys of one kv head
d_o = torch.randn(B * S, Hq * hd, device="cuda").to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
dq_, dk_, dv_ = dqkv[:, :Hq * hd], dqkv[:, Hq * hd:(Hq + Hkv) * hd], dqkv[:, (Hq + Hkv) * hd:]
for _ in range(5):
    ops.attn_bwd(q, k, v, o, d_o, lse, dq_, dk_, dv_, B, S, Hq, Hkv, hd, hd ** -0.5, causal)
torch.cuda.synchronize()
for which, name, nb in ((0, "dQ", ((S + 127) // 128) * Hq * B), (1, "dK/dV", ((S + 127) // 128) * Hkv * B)):
    buf = np.zeros(nb * 8, dtype=np.uint64)
    assert ops.lib.zk_debug_read_stamps_bwd(buf.ctypes.data_as(ctypes.c_void_p), nb, which) == 0
    st = buf.reshape(nb, 8).astype(np.float64)
    t = st[:, :4] / 100.0
    t0 = t[:, 0].min()
    print(f"{name}: launch span {t[:, 3].max() - t0:.1f} us, {nb} workgroups; last entry at {t[:, 0].max() - t0:.1f} us; "
          f"sum of workgroup times {np.sum(t[:, 3] - t[:, 0]) / 1e3:.2f} ms-slots")
    for blk in sorted(set(st[:, 5].astype(int))):
        m = st[:, 5].astype(int) == blk
        d = np.diff(t[m], axis=1).mean(0)
        it = st[m, 6][0]
        print(f"  block {blk} ({int(it)} iterations): start {np.mean(t[m, 0] - t0):6.1f}  prologue {d[0]:5.2f}  loop {d[1]:6.2f} "
              f"({d[1] / max(it, 1):.2f}/iteration)  epilogue {d[2]:5.2f}  total {np.mean(t[m, 3] - t[m, 0]):6.2f} us")
