"""Achieved HBM bandwidth of the bandwidth-bound kernels at the C3 step's shapes (SURVEY.md §8d:
"achieved HBM GB/s for norm/elementwise kernels").   python tools/bench_bw.py [iters] [--json out]

Every launch gets operand buffers of its own, rotated through a pool larger than the 256 MB
Infinity Cache plus the L2s, so a launch finds none of its operands cached: this is the HBM rate,
not the cache rate.  Algorithmic bytes are DESIGN.md §4.3's per-element figures times the elements
of one launch; peak = 8 TB/s (MI355X_MICROARCH.md).
"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zero_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 40
out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
PEAK = 8000.0   # GB/s
POOL_BYTES = 768 << 20
dev = "cuda"
BF = torch.bfloat16


def pool(make, nbytes_one):
    n = max(2, min(64, POOL_BYTES // max(1, nbytes_one) + 1))
    return [make() for _ in range(n)]


def timeit(fn, n_sets):
    """One launch per operand set, captured in a HIP graph (the small kernels take less time than one
    Python call: timed from eager code they would show the host's rate, not the device's)."""
    for i in range(min(3, n_sets)):
        fn(i % n_sets)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n_sets):
            fn(i)
    reps = max(1, (iters + n_sets - 1) // n_sets)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * n_sets) * 1e-3   # seconds per launch


results = []


def report(name, shape, nbytes, sec):
    gbs = nbytes / sec / 1e9
    results.append({"kernel": name, "shape": shape, "algorithmic_MB": round(nbytes / 1e6, 2),
                    "us": round(sec * 1e6, 1), "GBps": round(gbs, 0), "frac_of_8TBps": round(gbs / PEAK, 3)})
    print(f"{name:28s} {shape:28s} {nbytes/1e6:9.1f} MB {sec*1e6:8.1f} us {gbs:7.0f} GB/s  {gbs/PEAK:5.2f} of peak",
          flush=True)


def rnd(*shape):
    return torch.randn(*shape, device=dev).to(BF)


# ---- decoder RMSNorm: rows = B*S = 6144, d = 3072 --------------------------------------------
R, D = 6144, 3072
w = torch.ones(D, device=dev, dtype=BF)
xs = pool(lambda: rnd(R, D), R * D * 4)
ys_rs = [ops.rmsnorm_fwd(x, w, 1e-5) for x in xs]
report("rmsnorm_fwd", f"[{R},{D}]", R * D * 4, timeit(lambda i: ops.rmsnorm_fwd(xs[i], w, 1e-5), len(xs)))
dys = [rnd(R, D) for _ in xs]
report("rmsnorm_bwd", f"[{R},{D}]", R * D * 6,
       timeit(lambda i: ops.rmsnorm_bwd(dys[i], xs[i], w, ys_rs[i][1]), len(xs)))
report("rmsnorm_bwd + residual grad", f"[{R},{D}]", R * D * 8,
       timeit(lambda i: ops.rmsnorm_bwd(dys[i], xs[i], w, ys_rs[i][1], dres=ys_rs[i][0]), len(xs)))
dw = torch.zeros(ops.norm_parts(R), D, device=dev, dtype=torch.float32)
report("rmsnorm_dweight", f"[{R},{D}]", R * D * 4,
       timeit(lambda i: ops.rmsnorm_dweight_(dys[i], xs[i], ys_rs[i][1], dw), len(xs)))
del xs, ys_rs, dys

# ---- encoder LayerNorm: rows = 8*577 = 4616, D = 1024 ---------------------------------------
R, D = 4616, 1024
g, b = torch.ones(D, device=dev, dtype=BF), torch.zeros(D, device=dev, dtype=BF)
xs = pool(lambda: rnd(R, D), R * D * 4)
st = [ops.layernorm_fwd(x, g, b, 1e-5) for x in xs]
report("layernorm_fwd", f"[{R},{D}]", R * D * 4, timeit(lambda i: ops.layernorm_fwd(xs[i], g, b, 1e-5), len(xs)))
dys = [rnd(R, D) for _ in xs]
dg, db = (torch.zeros(ops.norm_parts(R), D, device=dev, dtype=torch.float32) for _ in range(2))
report("layernorm_bwd (+dgamma,dbeta)", f"[{R},{D}]", R * D * 8,
       timeit(lambda i: ops.layernorm_bwd(dys[i], xs[i], g, st[i][1], st[i][2], dres=st[i][0], dgamma=dg, dbeta=db),
              len(xs)))
del xs, st, dys

# ---- RoPE in place on packed q|k|v: rows 6144, 24 + 8 heads of 128 (row stride 5120) -------
R, S, Hq, Hkv, hd = 6144, 768, 24, 8, 128
inv = 1.0 / (5e5 ** (torch.arange(0, hd, 2, device=dev, dtype=torch.float32) / hd))
ang = torch.arange(S, device=dev, dtype=torch.float32)[:, None] * inv[None]
cos_t, sin_t = ang.cos().contiguous(), ang.sin().contiguous()
qkvs = pool(lambda: rnd(R, (Hq + 2 * Hkv) * hd), R * (Hq + Hkv) * hd * 4)
report("rope (q|k in place)", f"[{R},{(Hq+Hkv)*hd}] of 5120", R * (Hq + Hkv) * hd * 4,
       timeit(lambda i: ops.rope_(qkvs[i], cos_t, sin_t, S, Hq + Hkv, hd), len(qkvs)))
del qkvs

# ---- SwiGLU (trainable translator layer; the frozen layers use the GEMM epilogues) ----------
R, F = 6144, 8192
gus = pool(lambda: rnd(R, 2 * F), R * F * 6)
report("swiglu_fwd", f"[{R},2x{F}]", R * F * 6, timeit(lambda i: ops.swiglu_fwd(gus[i], F), len(gus)))
dhs = [rnd(R, F) for _ in gus]
report("swiglu_bwd", f"[{R},2x{F}]", R * F * 10, timeit(lambda i: ops.swiglu_bwd(dhs[i], gus[i], F), len(gus)))
del gus, dhs

# ---- quick-GELU backward multiply (encoder layers 16-23 when not folded in an epilogue) -----
R, F = 4616, 4096
pre = pool(lambda: rnd(R, F), R * F * 6)
dys = [rnd(R, F) for _ in pre]
report("act_bwd quick_gelu", f"[{R},{F}]", R * F * 6, timeit(lambda i: ops.act_bwd(dys[i], pre[i], 1), len(pre)))
ob = torch.zeros(ops.colsum_parts(R), F, device=dev, dtype=torch.float32)
report("colsum (bias gradient)", f"[{R},{F}]", R * F * 2, timeit(lambda i: ops.colsum_(dys[i], ob), len(pre)))
del pre, dys

# ---- cross entropy on bf16 logits: 8 x 128 labelled rows, V = 128256 (read + write dlogits in place)
R, V = 1024, 128256
lgs = pool(lambda: rnd(R, V), R * V * 4)
tg = torch.randint(0, V, (R,), device=dev, dtype=torch.int32)
acc = torch.zeros(2, device=dev, dtype=torch.float32)
report("cross_entropy (+dlogits)", f"[{R},{V}]", R * V * 4,
       timeit(lambda i: ops.cross_entropy(lgs[i], tg, acc, dlogits=lgs[i]), len(lgs)))
del lgs

# ---- optimizer tail on the flat buffers: 113.4 M trainable parameters -----------------------
N = 113_400_000 // 8 * 8
p, gr = rnd(N), rnd(N)
m, v = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
ss = torch.zeros(1, device=dev, dtype=torch.float32)
report("sumsq (clip norm)", f"[{N}]", N * 2, timeit(lambda i: ops.sumsq_(gr, ss), 1))
report("adamw (clip folded in)", f"[{N}]", N * 22,
       timeit(lambda i: ops.adamw_step_(p, gr, m, v, lr=1e-5, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
                                        step=1, max_norm=1.0, sumsq=ss), 1))

if out_json:
    with open(out_json, "w") as f:
        json.dump({"peak_GBps": PEAK, "iters": iters, "pool_bytes": POOL_BYTES, "kernels": results}, f, indent=1)
