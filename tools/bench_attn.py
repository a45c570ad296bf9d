"""Attention micro-benchmark on the two shapes of the C3 step.   python tools/bench_attn.py [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zero_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
VARIANT = os.environ.get("ZERO_AMD_ATTN_VARIANT", "default")     # "w64": time zk_attn_fwd_w64 as the forward
SHAPES = [("decoder causal", 8, 768, 24, 8, 128, True), ("encoder bidir ", 8, 577, 16, 16, 64, False)]
if os.environ.get("ZERO_AMD_ATTN_SWEEP"):     # where the time goes: fixed cost per launch / block vs per tile
    SHAPES += [("causal S=1536", 8, 1536, 24, 8, 128, True), ("causal S=3072", 4, 3072, 24, 8, 128, True),
               ("causal S=6144", 2, 6144, 24, 8, 128, True), ("bidir  S=768 ", 8, 768, 24, 8, 128, False),
               ("bidir  S=3072", 4, 3072, 24, 8, 128, False), ("causal B=32  ", 32, 768, 24, 8, 128, True),
               ("enc    B=64  ", 64, 577, 16, 16, 64, False), ("enc  S=2308  ", 8, 2308, 16, 16, 64, False)]
for name, B, S, Hq, Hkv, hd, causal in SHAPES:
    torch.manual_seed(0)
    qkv = torch.randn(B * S, (Hq + 2 * Hkv) * hd, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[:, :Hq * hd], qkv[:, Hq * hd:(Hq + Hkv) * hd], qkv[:, (Hq + Hkv) * hd:]
    d_o = torch.randn(B * S, Hq * hd, device="cuda").to(torch.bfloat16)
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[:, :Hq * hd], dqkv[:, Hq * hd:(Hq + Hkv) * hd], dqkv[:, (Hq + Hkv) * hd:]
    scale = hd ** -0.5
    o, lse = ops.attn_fwd(q, k, v, B, S, Hq, Hkv, hd, scale, causal)
    fl = 2.0 * S * S * hd * Hq * B * (0.5 if causal else 1.0)   # one S x S x hd product per (b, head)

    def timeit(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    tf = timeit(lambda: ops.attn_fwd(q, k, v, B, S, Hq, Hkv, hd, scale, causal, out=o, variant=VARIANT))
    tb = timeit(lambda: ops.attn_bwd(q, k, v, o, d_o, lse, dq, dk, dv, B, S, Hq, Hkv, hd, scale, causal))
    print(f"{name} B={B} S={S} Hq={Hq} Hkv={Hkv} hd={hd}: fwd {tf*1e3:6.0f} us ({2*fl/tf/1e9:6.0f} TF)   "
          f"bwd {tb*1e3:6.0f} us ({7*fl/tb/1e9:6.0f} TF executed-7-products, {5*fl/tb/1e9:6.0f} TF algorithmic)", flush=True)
