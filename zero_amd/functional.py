"""Autograd nodes of the hot path.  One node per LAYER (not per op): forward and backward
are explicit sequences of C-ABI kernel launches, activations are saved by hand, frozen
weights get no wgrad and frozen layers save nothing.

Parameter gradients: if a parameter carries a pre-assigned slot (``p._zk_grad``, a bf16 view
into the trainer's flat gradient buffer, see distributed/ddp.py) the wgrad kernels write
(or accumulate) straight into it and the node returns None for that input; otherwise the
gradient tensor is returned to autograd as usual (plain drop-in behaviour).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import os
import threading

import torch

from . import ops

BF16 = torch.bfloat16


# ------------------------------------------------------------------------------------------
# gradient sinks
# ------------------------------------------------------------------------------------------
class ZeroArena:
    """fp32 scratch for the partial sums of the vector gradients (bias / norm-affine): one buffer handed out in
    64-element-aligned slices and recycled after every layer's conversion launch (GradSink.flush) — the kernels
    OVERWRITE their partial-sum slabs, so nothing is ever zero-filled (the name is historical)."""

    def __init__(self, device, numel: int = 8 << 20):
        self.buf = torch.empty(numel, dtype=torch.float32, device=device)
        self.ptr = 0

    def reset(self):
        self.ptr = 0

    def take(self, n: int) -> Optional[torch.Tensor]:
        start = self.ptr
        end = start + (n + 63) // 64 * 64
        if end > self.buf.numel():
            return None
        self.ptr = end
        return self.buf[start:start + n]


class SinkState:
    """What one gradient owner (a FlatGradReducer, i.e. one Trainer replica) keeps for its slots: the fp32 arena of
    the vector gradients' partial sums, the conversions queued by the running backward node, and the callback
    fired when a slot has received its gradient.  Reached through the PARAMETER (``p._zk_sink``), never through
    process-wide state: two Trainers in one process (the rank-thread DDP harness) keep separate queues."""

    def __init__(self, on_grad=None, arena: Optional[ZeroArena] = None):
        self.on_grad = on_grad
        self.arena = arena
        self.pending = []       # (fp32 accumulator, bf16 slot, accumulate, param): converted by GradSink.flush()


class GradSink:
    """Routes a parameter gradient either into its flat-buffer slot or back to autograd.  A layer node's backward
    calls ``bind(params...)`` first: the SinkState of its parameters becomes the node's current one (thread-local:
    a node runs start to end on one autograd thread) until ``flush()`` ends the node."""

    _tls = threading.local()

    @staticmethod
    def slot(p: torch.Tensor) -> Optional[torch.Tensor]:
        return getattr(p, "_zk_grad", None)

    @staticmethod
    def is_dirty(p) -> bool:
        return getattr(p, "_zk_dirty", False)

    @staticmethod
    def bind(*params):
        cur = None
        for p in params:
            st = getattr(p, "_zk_sink", None) if p is not None else None
            if st is not None:
                if cur is not None and st is not cur:
                    raise RuntimeError("one layer's parameters belong to two gradient owners")
                cur = st
        GradSink._tls.cur = cur

    @staticmethod
    def _cur() -> Optional[SinkState]:
        return getattr(GradSink._tls, "cur", None)

    @staticmethod
    def done(p):
        p._zk_dirty = True
        st = getattr(p, "_zk_sink", None)
        if st is not None and st.on_grad is not None:
            st.on_grad(p)

    @staticmethod
    def queue(parts_f32, slot, p):
        st = p._zk_sink
        cur = GradSink._cur()
        if cur is None:
            GradSink._tls.cur = st
        elif cur is not st:
            raise RuntimeError("a vector gradient was queued for another owner than the node is bound to")
        st.pending.append((parts_f32, slot, GradSink.is_dirty(p), p))

    @staticmethod
    def parts(nparts: int, n: int, device) -> torch.Tensor:
        """Uninitialised fp32 [nparts, n] slab for one vector gradient's partial sums."""
        cur = GradSink._cur()
        a = cur.arena if cur is not None else None
        if a is not None and a.buf.device == torch.device(device):
            t = a.take(nparts * n)
            if t is not None:
                return t.view(nparts, n)
        return torch.empty((nparts, n), dtype=torch.float32, device=device)

    @staticmethod
    def flush():
        """One conversion launch for every vector gradient queued by this node (called at the end of each
        layer node's backward), then the slot notifications; ends the node's binding."""
        cur = GradSink._cur()
        GradSink._tls.cur = None
        if cur is None:
            return
        if cur.pending:
            batch, cur.pending = cur.pending, []
            ops.f32_to_bf16_multi_([(a, s_, acc) for a, s_, acc, _ in batch])
            for _, _, _, p in batch:
                GradSink.done(p)
        if cur.arena is not None:
            cur.arena.reset()            # stream-ordered: the next layer's kernels run after this conversion


# ------------------------------------------------------------------------------------------
# dgrad through FROZEN weights: keep a transposed copy resident (288 GB HBM: the 3B surrogate's
# second layout costs 4.4 GB) so dx = dy W runs as the k-contiguous (NT) GEMM variant, which is
# ~25 % faster on MI355X than reading W through the transposed-LDS path (tools/bench_gemm.py).
# Trainable weights change every step and keep the copy-free transposed-read variant.
# ------------------------------------------------------------------------------------------
USE_TRANSPOSED_FROZEN = os.environ.get("ZERO_AMD_DGRAD_T", "1") != "0"
FROZEN_T_BUILDS = 0      # resident copies built so far (tests: a steady-state step builds none)


def _transposed(w: torch.Tensor) -> torch.Tensor:
    """[in, out] copy of an [out, in] weight: zk_transpose_bf16 (HBM rate) where it applies, ATen otherwise."""
    if w.is_cuda and w.dtype == BF16 and w.dim() == 2 and w.stride(1) == 1 and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0:
        return ops.transpose(w)
    return w.t().contiguous()


def _frozen_t(w: torch.Tensor, owner: torch.Tensor) -> torch.Tensor:
    """The copy lives on the owning Parameter object (not in a global table keyed by address:
    a freed model's addresses get reused), and is rebuilt if the weights were written to."""
    cache = owner.__dict__.setdefault("_zk_wt", {})
    key = tuple(w.shape)
    hit = cache.get(key)
    if hit is not None and hit[0] == owner._version and hit[1] == w.data_ptr():
        return hit[2]
    global FROZEN_T_BUILDS
    FROZEN_T_BUILDS += 1
    wt = _transposed(w.detach())              # one-time layout preparation (setup, not the hot path)
    cache[key] = (owner._version, w.data_ptr(), wt)
    return wt


def _frozen_gate_up(gw: torch.Tensor, uw: torch.Tensor):
    """Resident copies for the fused SwiGLU of a FROZEN MLP: (interleaved [2F, d] weight, its transpose
    [d, 2F] for the dgrad, down-projection helper not included).  Cached on the gate Parameter."""
    cache = gw.__dict__.setdefault("_zk_gu", {})
    hit = cache.get("gu")
    if (hit is not None and hit[0] == (gw._version, uw._version) and hit[1] == (gw.data_ptr(), uw.data_ptr())):
        return hit[2], hit[3]
    global FROZEN_T_BUILDS
    FROZEN_T_BUILDS += 1
    wi = ops.interleave_gate_up(gw.detach(), uw.detach())
    wit = wi.t().contiguous()
    cache["gu"] = ((gw._version, uw._version), (gw.data_ptr(), uw.data_ptr()), wi, wit)
    return wi, wit


FUSE_SWIGLU = os.environ.get("ZERO_AMD_FUSE_SWIGLU", "1") != "0"


def dgrad(dy2d: torch.Tensor, w: torch.Tensor, owner: Optional[torch.Tensor] = None, **kw) -> torch.Tensor:
    """dx[M, in] = dy[M, out] @ W[out, in].  ``owner``: the Parameter whose storage ``w`` views."""
    owner = w if owner is None else owner
    if (USE_TRANSPOSED_FROZEN and not owner.requires_grad and GradSink.slot(owner) is None
            and w.shape[0] >= 256):
        return ops.gemm(dy2d, _frozen_t(w, owner), **kw)
    return ops.gemm(dy2d, w, trans_b=True, **kw)


# ------------------------------------------------------------------------------------------
# dgrad through TRAINABLE weights with many rows (C2: 64 x 577 = 36928 tokens): transposing the weight once
# per optimizer step (8-34 MB of byte traffic per encoder layer) lets dx = dy W run as the NT product on the
# persistent-ring kernel (~1.0 PF) instead of the transposed-LDS-read kernel (~0.5 PF, tools/bench_enc_gemm.py).
# With ATen's strided copy (~0.4 TB/s) the copies only paid from ~8k rows; zk_transpose_bf16 makes them at HBM rate
# (25 MB per encoder layer: ~10 us), so C3's encoder (M = 4616: 230 -> 156 us of dgrad per layer) takes this path
# too (threshold 2048 rows; C3 139.5 -> 140.3 img/s, C2 ~1045 -> 1076).  The copy is keyed on (weights epoch, tensor version, address): the Trainer's fused AdamW writes the
# parameters through raw pointers, which no autograd version counter sees, so it bumps WEIGHTS_EPOCH.
# ------------------------------------------------------------------------------------------
WEIGHTS_EPOCH = 0
TRANSPOSED_TRAINABLE_MIN_ROWS = int(os.environ.get("ZERO_AMD_DGRAD_T_MIN_ROWS", "2048"))


def bump_weights_epoch():
    global WEIGHTS_EPOCH
    WEIGHTS_EPOCH += 1


def _step_t(w: torch.Tensor, owner: torch.Tensor) -> torch.Tensor:
    cache = owner.__dict__.setdefault("_zk_wt_step", {})
    key = tuple(w.shape)
    stamp = (WEIGHTS_EPOCH, owner._version, w.data_ptr())
    hit = cache.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    wt = _transposed(w.detach())
    cache[key] = (stamp, wt)
    return wt


def dgrad_trainable(dy2d: torch.Tensor, w: torch.Tensor, owner: Optional[torch.Tensor] = None, **kw) -> torch.Tensor:
    """dx = dy @ W for a weight that changes every step (``owner``: the Parameter whose storage ``w`` views)."""
    if dy2d.shape[0] >= TRANSPOSED_TRAINABLE_MIN_ROWS and w.shape[0] >= 256 and w.shape[1] >= 256:
        return ops.gemm(dy2d, _step_t(w, w if owner is None else owner), **kw)
    return ops.gemm(dy2d, w, trans_b=True, **kw)


def _wgrad(dy2d, x2d, w: torch.Tensor, need: bool):
    """dW[out,in] = dY^T X, into the slot when there is one.  Returns what autograd should get."""
    if not need:
        return None
    slot = GradSink.slot(w)
    if slot is not None:
        ops.gemm(dy2d, x2d, trans_a=True, trans_b=True, out=slot.view(w.shape), accumulate=GradSink.is_dirty(w))
        GradSink.done(w)
        return None
    return ops.gemm(dy2d, x2d, trans_a=True, trans_b=True)


def _vec_grad(parts_f32: torch.Tensor, p: torch.Tensor, need: bool):
    """fp32 partial sums [nparts, n] of a vector gradient (bias / norm affine) -> bf16 slot or autograd."""
    if not need:
        return None
    slot = GradSink.slot(p)
    if slot is not None:
        GradSink.queue(parts_f32, slot, p)       # summed + converted by GradSink.flush()
        return None
    out = torch.empty(p.shape, dtype=BF16, device=p.device)
    ops.f32_to_bf16_multi_([(parts_f32, out.view(-1), False)])
    return out


def _colsum(dy2d) -> torch.Tensor:
    parts = GradSink.parts(ops.colsum_parts(dy2d.shape[0]), dy2d.shape[1], dy2d.device)
    return ops.colsum_(dy2d, parts)


def _bias_grad(dy2d, b: torch.Tensor, need: bool):
    if not need:
        return None
    return _vec_grad(_colsum(dy2d), b, True)


def adjacent(*ts: torch.Tensor) -> bool:
    """True when the tensors are back-to-back slices of one storage (packed weights)."""
    for a, b in zip(ts, ts[1:]):
        if not (a.is_contiguous() and b.is_contiguous()):
            return False
        if a.untyped_storage().data_ptr() != b.untyped_storage().data_ptr():
            return False
        if b.storage_offset() != a.storage_offset() + a.numel():
            return False
    return True


def packed_rows(*ts: torch.Tensor) -> Optional[torch.Tensor]:
    """[sum(out_i), in] view over adjacent [out_i, in] weights, or None."""
    if not adjacent(*ts):
        return None
    rows = sum(t.shape[0] for t in ts)
    return ts[0].as_strided((rows, ts[0].shape[1]), (ts[0].shape[1], 1))


def packed_vec(*ts: torch.Tensor) -> Optional[torch.Tensor]:
    if not adjacent(*ts):
        return None
    return ts[0].as_strided((sum(t.numel() for t in ts),), (1,))


# ------------------------------------------------------------------------------------------
# single ops as autograd nodes (drop-in module building blocks)
# ------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b); epilogue in {NONE, QUICK_GELU, GELU_ERF}."""

    @staticmethod
    def forward(ctx, x, w, b, epilogue):
        x2 = x.reshape(-1, x.shape[-1])
        need = any(ctx.needs_input_grad)
        pre = None
        if epilogue != ops.EPI_NONE and need:
            pre = torch.empty((x2.shape[0], w.shape[0]), dtype=BF16, device=x.device)
        y = ops.gemm(x2, w, bias=b, epilogue=epilogue, aux_out=pre)
        ctx.epilogue = epilogue
        ctx.save_for_backward(x2, w, b, pre)
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, b, pre = ctx.saved_tensors
        GradSink.bind(w, b)
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        if ctx.epilogue != ops.EPI_NONE:
            dy2 = ops.act_bwd(dy2, pre, ctx.epilogue)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = dgrad(dy2, w).view(*dy.shape[:-1], w.shape[1])
        dw = _wgrad(dy2, x2, w, ctx.needs_input_grad[1])
        db = _bias_grad(dy2, b, b is not None and ctx.needs_input_grad[2])
        GradSink.flush()
        return dx, dw, db, None


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        need = any(ctx.needs_input_grad)
        y, mean, rstd = ops.layernorm_fwd(x2, gamma, beta, eps, save_stats=need)
        ctx.save_for_backward(x2, gamma, beta, mean, rstd)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, gamma, beta, mean, rstd = ctx.saved_tensors
        GradSink.bind(gamma, beta)
        dy2 = dy.reshape(x2.shape)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        need_p = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        D = x2.shape[1]
        npn = ops.norm_parts(x2.shape[0])
        dg = GradSink.parts(npn, D, x2.device) if need_p else None
        db = GradSink.parts(npn, D, x2.device) if need_p else None
        dx = ops.layernorm_bwd(dy2, x2, gamma, mean, rstd, dgamma=dg, dbeta=db, need_dx=ctx.needs_input_grad[0])
        out = (dx.view(dy.shape) if dx is not None else None, _vec_grad(dg, gamma, ctx.needs_input_grad[1]),
               _vec_grad(db, beta, ctx.needs_input_grad[2]), None)
        GradSink.flush()
        return out


class RMSNormFn(torch.autograd.Function):
    """Weight is treated as frozen (the surrogate decoder is frozen on this path)."""

    @staticmethod
    def forward(ctx, x, w, eps):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        y, rstd = ops.rmsnorm_fwd(x2, w, eps, save_stats=ctx.needs_input_grad[0])
        ctx.save_for_backward(x2, w, rstd)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w, rstd = ctx.saved_tensors
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("RMSNorm weight gradient: the decoder is frozen on this path")
        dy2 = dy.reshape(x2.shape)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        return ops.rmsnorm_bwd(dy2, x2, w, rstd).view(dy.shape), None, None


# ------------------------------------------------------------------------------------------
# CLIP encoder layer (tf:clip:362-383) as one node
# ------------------------------------------------------------------------------------------
class EncoderLayerFn(torch.autograd.Function):
    """inputs: x [B,T,D], then the 16 parameter tensors in state-dict order:
    ln1.w ln1.b  q.w q.b k.w k.b v.w v.b o.w o.b  ln2.w ln2.b  fc1.w fc1.b fc2.w fc2.b ; heads, eps"""

    @staticmethod
    def forward(ctx, x, l1w, l1b, qw, qb, kw, kb, vw, vb, ow, ob, l2w, l2b, f1w, f1b, f2w, f2b, heads, eps):
        B, T, D = x.shape
        M = B * T
        hd = D // heads
        need = any(ctx.needs_input_grad)
        x2 = x.reshape(M, D)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        h1, mean1, rstd1 = ops.layernorm_fwd(x2, l1w, l1b, eps, save_stats=need)
        qkv = torch.empty((M, 3 * D), dtype=BF16, device=x.device)
        wqkv, bqkv = packed_rows(qw, kw, vw), packed_vec(qb, kb, vb)
        if wqkv is not None and bqkv is not None:
            ops.gemm(h1, wqkv, bias=bqkv, out=qkv)
        else:
            for i, (w_, b_) in enumerate(((qw, qb), (kw, kb), (vw, vb))):
                ops.gemm(h1, w_, bias=b_, out=qkv[:, i * D:(i + 1) * D])
        ao, lse = ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, T, heads, heads, hd, hd ** -0.5,
                               False, need_lse=need)
        xa = ops.gemm(ao, ow, bias=ob, residual=x2)
        h2, mean2, rstd2 = ops.layernorm_fwd(xa, l2w, l2b, eps, save_stats=need)
        pre = torch.empty((M, f1w.shape[0]), dtype=BF16, device=x.device) if need else None
        act = ops.gemm(h2, f1w, bias=f1b, epilogue=ops.EPI_QUICK_GELU, aux_out=pre)
        y = ops.gemm(act, f2w, bias=f2b, residual=xa)
        if need:
            ctx.save_for_backward(x2, mean1, rstd1, h1, qkv, ao, lse, xa, mean2, rstd2, h2, pre, act,
                                  l1w, l1b, qw, qb, kw, kb, vw, vb, ow, ob, l2w, l2b, f1w, f1b, f2w, f2b)
            ctx.dims = (B, T, D, heads, hd)
        return y.view(B, T, D)

    @staticmethod
    def backward(ctx, dy):
        (x2, mean1, rstd1, h1, qkv, ao, lse, xa, mean2, rstd2, h2, pre, act,
         l1w, l1b, qw, qb, kw, kb, vw, vb, ow, ob, l2w, l2b, f1w, f1b, f2w, f2b) = ctx.saved_tensors
        GradSink.bind(l1w, l1b, qw, qb, kw, kb, vw, vb, ow, ob, l2w, l2b, f1w, f1b, f2w, f2b)
        B, T, D, heads, hd = ctx.dims
        M = B * T
        nig = ctx.needs_input_grad
        dev = dy.device
        dy2 = dy.reshape(M, D)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        f32 = lambda n: GradSink.parts(ops.norm_parts(M), n, dev)      # LayerNorm affine partial sums
        # ---- MLP block:  y = xa + fc2(quick_gelu(fc1(LN2(xa))))
        d_pre = dgrad_trainable(dy2, f2w, epilogue=ops.EPI_MUL_DQUICK_GELU, aux_in=pre)
        g_f2w = _wgrad(dy2, act, f2w, nig[15])
        g_f2b = _bias_grad(dy2, f2b, nig[16])
        g_f1w = _wgrad(d_pre, h2, f1w, nig[13])
        g_f1b = _bias_grad(d_pre, f1b, nig[14])
        d_h2 = dgrad_trainable(d_pre, f1w)
        need_ln2 = nig[11] or nig[12]
        dg2, db2 = (f32(D), f32(D)) if need_ln2 else (None, None)
        dxa = ops.layernorm_bwd(d_h2, xa, l2w, mean2, rstd2, dres=dy2, dgamma=dg2, dbeta=db2)
        g_l2w, g_l2b = _vec_grad(dg2, l2w, nig[11]), _vec_grad(db2, l2b, nig[12])
        # ---- attention block:  xa = x + out(attn(qkv(LN1(x))))
        d_ao = dgrad_trainable(dxa, ow)
        g_ow = _wgrad(dxa, ao, ow, nig[9])
        g_ob = _bias_grad(dxa, ob, nig[10])
        d_qkv = torch.empty_like(qkv)
        ops.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], ao, d_ao, lse, d_qkv[:, :D], d_qkv[:, D:2 * D],
                     d_qkv[:, 2 * D:], B, T, heads, heads, hd, hd ** -0.5, False)
        wqkv = packed_rows(qw, kw, vw)
        slots_w = [GradSink.slot(p) for p in (qw, kw, vw)]
        slots_b = [GradSink.slot(p) for p in (qb, kb, vb)]
        g_qw = g_kw = g_vw = g_qb = g_kb = g_vb = None
        packed_slots = (all(s is not None for s in slots_w) and adjacent(*slots_w) and nig[3] and nig[5] and nig[7]
                        and len({GradSink.is_dirty(p) for p in (qw, kw, vw)}) == 1)
        if packed_slots:
            ops.gemm(d_qkv, h1, trans_a=True, trans_b=True, out=packed_rows(*[s.view(D, D) for s in slots_w]),
                     accumulate=GradSink.is_dirty(qw))
            for p in (qw, kw, vw):
                GradSink.done(p)
        else:
            g_qw = _wgrad(d_qkv[:, :D], h1, qw, nig[3])
            g_kw = _wgrad(d_qkv[:, D:2 * D], h1, kw, nig[5])
            g_vw = _wgrad(d_qkv[:, 2 * D:], h1, vw, nig[7])
        if nig[4] or nig[6] or nig[8]:
            acc = _colsum(d_qkv)                       # [nparts, 3D]: column slices keep the part stride
            g_qb = _vec_grad(acc[:, :D], qb, nig[4])
            g_kb = _vec_grad(acc[:, D:2 * D], kb, nig[6])
            g_vb = _vec_grad(acc[:, 2 * D:], vb, nig[8])
        need_dx = nig[0]
        dx = None
        g_l1w = g_l1b = None
        if need_dx or nig[1] or nig[2]:
            if wqkv is not None:
                d_h1 = dgrad_trainable(d_qkv, wqkv, owner=qw)
            else:
                d_h1 = ops.gemm(d_qkv[:, :D], qw, trans_b=True)
                ops.gemm(d_qkv[:, D:2 * D], kw, trans_b=True, out=d_h1, accumulate=True)
                ops.gemm(d_qkv[:, 2 * D:], vw, trans_b=True, out=d_h1, accumulate=True)
            dg1, db1 = (f32(D), f32(D)) if (nig[1] or nig[2]) else (None, None)
            dx = ops.layernorm_bwd(d_h1, x2, l1w, mean1, rstd1, dres=dxa, dgamma=dg1, dbeta=db1, need_dx=need_dx)
            g_l1w, g_l1b = _vec_grad(dg1, l1w, nig[1]), _vec_grad(db1, l1b, nig[2])
            if dx is not None:
                dx = dx.view(B, T, D)
        GradSink.flush()
        return (dx, g_l1w, g_l1b, g_qw, g_qb, g_kw, g_kb, g_vw, g_vb, g_ow, g_ob, g_l2w, g_l2b,
                g_f1w, g_f1b, g_f2w, g_f2b, None, None)


# ------------------------------------------------------------------------------------------
# Llama decoder layer (tf:llama:295-324).  Frozen layers (the surrogate on the finetune-encoder
# path): dgrad only, resident transposed / interleaved weight copies, fused SwiGLU epilogues.
# Trainable layers (stage "pretrain": the translator, ref:configs/zero/train_llama.py:281-307):
# the same forward through the copy-free kernels (the weights change every step), h1 / h2 / hm kept
# for the seven weight gradients, RMSNorm weight gradients by zk_rmsnorm_dweight.
# ------------------------------------------------------------------------------------------
class DecoderLayerFn(torch.autograd.Function):
    """inputs: x [B,S,d]; ln1.w; q.w k.w v.w o.w; ln2.w; gate.w up.w down.w; cos, sin; cfg tuple.

    ``rows`` / ``inv_rows`` (int32, both or neither; LAST layer of a frozen decoder only): the flat positions whose
    output anything reads (the rows the loss is formed on) and the inverse map position -> index or -1.  Everything
    behind the attention of a decoder layer is row-wise (o-projection, residual adds, RMSNorm, the MLP), and no later
    layer attends to this layer's output, so the output rows that are not listed feed nothing: the node then returns
    the listed rows only, as [1, R, d], and its backward takes their gradient.  Keys, values and the attention itself
    still cover every position; loss and gradients are those of the full computation."""

    @staticmethod
    def forward(ctx, x, l1w, qw, kw, vw, ow, l2w, gw, uw, dw, cos_t, sin_t, dims, rows=None, inv_rows=None):
        nh, nkv, hd, eps = dims
        B, S, d = x.shape
        M = B * S
        trainable = any(ctx.needs_input_grad[1:10])
        need = ctx.needs_input_grad[0] or trainable
        x2 = x.reshape(M, d)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        h1, rstd1 = ops.rmsnorm_fwd(x2, l1w, eps, save_stats=need)
        W = (nh + 2 * nkv) * hd
        qkv = torch.empty((M, W), dtype=BF16, device=x.device)
        wqkv = packed_rows(qw, kw, vw)
        if wqkv is not None:
            ops.gemm(h1, wqkv, out=qkv)
        else:
            ops.gemm(h1, qw, out=qkv[:, :nh * hd])
            ops.gemm(h1, kw, out=qkv[:, nh * hd:(nh + nkv) * hd])
            ops.gemm(h1, vw, out=qkv[:, (nh + nkv) * hd:])
        ops.rope_(qkv, cos_t, sin_t, S, nh + nkv, hd)
        q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
        ao, lse = ops.attn_fwd(q, k, v, B, S, nh, nkv, hd, hd ** -0.5, S > 1, need_lse=need)
        tail = rows is not None
        if tail and trainable:
            raise ValueError("DecoderLayerFn: the row-restricted tail is for frozen layers")
        if tail:
            xa = ops.gemm(ops.gather_rows(ao, None, rows, ao.shape[1]), ow, residual=ops.gather_rows(x2, None, rows, d))
        else:
            xa = ops.gemm(ao, ow, residual=x2)
        h2, rstd2 = ops.rmsnorm_fwd(xa, l2w, eps, save_stats=need)
        F = gw.shape[0]
        fused = (FUSE_SWIGLU and USE_TRANSPOSED_FROZEN and F % 32 == 0 and not trainable
                 and not (gw.requires_grad or uw.requires_grad or dw.requires_grad)
                 and GradSink.slot(gw) is None and GradSink.slot(uw) is None and GradSink.slot(dw) is None)
        wgu = packed_rows(gw, uw)
        if fused:
            # gate|up GEMM with the SwiGLU in its epilogue; gu is kept in the interleaved column order
            gu, hm = ops.gemm_swiglu(h2, _frozen_gate_up(gw, uw)[0], keep_gu=need)   # (no backward: gu is not written)
        elif wgu is not None:
            gu = ops.gemm(h2, wgu)
            hm = ops.swiglu_fwd(gu, F)
        else:
            gu = torch.empty((h2.shape[0], 2 * F), dtype=BF16, device=x.device)
            ops.gemm(h2, gw, out=gu[:, :F])
            ops.gemm(h2, uw, out=gu[:, F:])
            hm = ops.swiglu_fwd(gu, F)
        y = ops.gemm(hm, dw, residual=xa)
        if need:
            extra = (h1, h2, hm) if trainable else (None, None, None)
            ctx.save_for_backward(x2, rstd1, qkv, ao, lse, xa, rstd2, gu, l1w, qw, kw, vw, ow, l2w, gw, uw, dw,
                                  cos_t, sin_t, *extra, *((inv_rows,) if tail else ()))
            ctx.dims = (B, S, d, nh, nkv, hd, F)
            ctx.fused_swiglu = fused
            ctx.tail = tail
        return y.view(1, -1, d) if tail else y.view(B, S, d)

    @staticmethod
    def backward(ctx, dy):
        (x2, rstd1, qkv, ao, lse, xa, rstd2, gu, l1w, qw, kw, vw, ow, l2w, gw, uw, dw, cos_t, sin_t,
         h1, h2, hm) = ctx.saved_tensors[:22]
        inv_rows = ctx.saved_tensors[22] if ctx.tail else None
        GradSink.bind(l1w, qw, kw, vw, ow, l2w, gw, uw, dw)
        B, S, d, nh, nkv, hd, F = ctx.dims
        M = B * S
        nig = ctx.needs_input_grad
        dy2 = dy.reshape(-1, d)          # (row-restricted tail: R rows, otherwise M)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        f32 = lambda n: GradSink.parts(ops.norm_parts(M), n, dy.device)      # RMSNorm weight partial sums
        wgu = packed_rows(gw, uw)
        g_dw = _wgrad(dy2, hm, dw, nig[9])
        if ctx.fused_swiglu:
            # dh = dy W_down with the SwiGLU backward in the epilogue (interleaved d_gate|d_up), then the
            # dgrad through the interleaved gate|up weight (same K order on both operands)
            d_gu = ops.gemm_swiglu_bwd(dy2, _frozen_t(dw, dw), gu)
            d_h2 = ops.gemm(d_gu, _frozen_gate_up(gw, uw)[1])
        elif wgu is not None:
            d_gu = ops.swiglu_bwd(dgrad(dy2, dw), gu, F)
            d_h2 = dgrad(d_gu, wgu, owner=gw)
        else:
            d_gu = ops.swiglu_bwd(dgrad(dy2, dw), gu, F)
            d_h2 = ops.gemm(d_gu[:, :F], gw, trans_b=True)
            ops.gemm(d_gu[:, F:], uw, trans_b=True, out=d_h2, accumulate=True)
        g_gw = _wgrad(d_gu[:, :F], h2, gw, nig[7])
        g_uw = _wgrad(d_gu[:, F:], h2, uw, nig[8])
        g_l2w = None
        if nig[6]:
            g_l2w = _vec_grad(ops.rmsnorm_dweight_(d_h2, xa, rstd2, f32(d)), l2w, True)
        dxa = ops.rmsnorm_bwd(d_h2, xa, l2w, rstd2, dres=dy2)
        d_ao = dgrad(dxa, ow)
        if inv_rows is not None:
            # back to every position: the rows nothing read have no gradient (zero rows through the inverse map)
            d_ao = ops.gather_rows(d_ao, None, inv_rows, d_ao.shape[1])
            dxa = ops.gather_rows(dxa, None, inv_rows, d)
        g_ow = _wgrad(dxa, ao, ow, nig[5])
        d_qkv = torch.empty_like(qkv)
        q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
        ops.attn_bwd(q, k, v, ao, d_ao, lse, d_qkv[:, :nh * hd], d_qkv[:, nh * hd:(nh + nkv) * hd],
                     d_qkv[:, (nh + nkv) * hd:], B, S, nh, nkv, hd, hd ** -0.5, S > 1,
                     rope=(cos_t, sin_t))      # the rotary backward rides on the dq / dk stores
        g_qw = _wgrad(d_qkv[:, :nh * hd], h1, qw, nig[2])
        g_kw = _wgrad(d_qkv[:, nh * hd:(nh + nkv) * hd], h1, kw, nig[3])
        g_vw = _wgrad(d_qkv[:, (nh + nkv) * hd:], h1, vw, nig[4])
        wqkv = packed_rows(qw, kw, vw)
        dx = g_l1w = None
        if nig[0] or nig[1]:
            if wqkv is not None:
                d_h1 = dgrad(d_qkv, wqkv, owner=qw)
            else:
                d_h1 = ops.gemm(d_qkv[:, :nh * hd], qw, trans_b=True)
                ops.gemm(d_qkv[:, nh * hd:(nh + nkv) * hd], kw, trans_b=True, out=d_h1, accumulate=True)
                ops.gemm(d_qkv[:, (nh + nkv) * hd:], vw, trans_b=True, out=d_h1, accumulate=True)
            if nig[1]:
                g_l1w = _vec_grad(ops.rmsnorm_dweight_(d_h1, x2, rstd1, f32(d)), l1w, True)
            if nig[0]:
                dx = ops.rmsnorm_bwd(d_h1, x2, l1w, rstd1, dres=dxa).view(B, S, d)
        GradSink.flush()
        return (dx, g_l1w, g_qw, g_kw, g_vw, g_ow, g_l2w, g_gw, g_uw, g_dw, None, None, None, None, None)


# ------------------------------------------------------------------------------------------
# adapter MLP (ref:models/adapter.py:34-38,95): Linear -> exact GELU -> Linear
# ------------------------------------------------------------------------------------------
class AdapterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w0, b0, w2, b2):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        need = any(ctx.needs_input_grad)
        pre = torch.empty((x2.shape[0], w0.shape[0]), dtype=BF16, device=x.device) if need else None
        act = ops.gemm(x2, w0, bias=b0, epilogue=ops.EPI_GELU_ERF, aux_out=pre)
        y = ops.gemm(act, w2, bias=b2)
        if need:
            ctx.save_for_backward(x2, pre, act, w0, b0, w2, b2)
        return y.view(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, pre, act, w0, b0, w2, b2 = ctx.saved_tensors
        GradSink.bind(w0, b0, w2, b2)
        nig = ctx.needs_input_grad
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        d_pre = dgrad_trainable(dy2, w2, epilogue=ops.EPI_MUL_DGELU_ERF, aux_in=pre)
        g_w2 = _wgrad(dy2, act, w2, nig[3])
        g_b2 = _bias_grad(dy2, b2, nig[4])
        g_w0 = _wgrad(d_pre, x2, w0, nig[1])
        g_b0 = _bias_grad(d_pre, b0, nig[2])
        dx = dgrad_trainable(d_pre, w0) if nig[0] else None
        if dx is not None:
            dx = dx.view(*dy.shape[:-1], w0.shape[1])
        GradSink.flush()
        return dx, g_w0, g_b0, g_w2, g_b2


# ------------------------------------------------------------------------------------------
# input assembly (ref:engine:186-236) and its backward
# ------------------------------------------------------------------------------------------
class SpliceFn(torch.autograd.Function):
    """inputs_embeds[r] = wte[code] | 0 | imgtab[-2-code]  for the flat [B*S] row codes.

    imgtab = [adapter rows ; soi ; eoi].  Backward gathers d_inputs through ``inv_codes``
    (row of d_inputs that consumed each adapter row, -1 if truncated away) and sums the
    soi / eoi rows over the batch; wte is frozen."""

    @staticmethod
    def forward(ctx, img_rows, soi, eoi, wte, codes, inv_codes, soi_pos, eoi_pos, B, S):
        d = wte.shape[1]
        n = img_rows.shape[0]
        imgtab = torch.empty((n + 2, d), dtype=BF16, device=wte.device)
        imgtab[:n].copy_(img_rows)
        imgtab[n].copy_(soi.reshape(-1))
        imgtab[n + 1].copy_(eoi.reshape(-1))
        out = ops.gather_rows(wte, imgtab, codes, d)
        ctx.save_for_backward(inv_codes, soi_pos, eoi_pos, soi, eoi)
        ctx.d = d
        return out.view(B, S, d)

    @staticmethod
    def backward(ctx, dy):
        inv_codes, soi_pos, eoi_pos, soi, eoi = ctx.saved_tensors
        GradSink.bind(soi, eoi)
        d = ctx.d
        dy2 = dy.reshape(-1, d)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        d_img = ops.gather_rows(dy2, None, inv_codes, d) if ctx.needs_input_grad[0] else None
        outs = []
        for pos, p, need in ((soi_pos, soi, ctx.needs_input_grad[1]), (eoi_pos, eoi, ctx.needs_input_grad[2])):
            if not need:
                outs.append(None)
                continue
            rows = ops.gather_rows(dy2, None, pos, d)
            outs.append(_vec_grad(_colsum(rows), p, True))
        GradSink.flush()
        return (d_img, outs[0], outs[1]) + (None,) * 7


# ------------------------------------------------------------------------------------------
# final norm + lm_head + shifted token-sum cross entropy (tf:llama:413,480; ref:engine:280-305)
# ------------------------------------------------------------------------------------------
class LMHeadLossFn(torch.autograd.Function):
    """loss_sum over the rows listed in ``rows`` (flat positions whose shifted label is not
    ignored) — rows with ignore_index contribute neither loss nor gradient, so the lm_head
    product is only formed for the listed rows.  Returns the fp32 loss sum (0-dim)."""

    @staticmethod
    def forward(ctx, hidden, norm_w, lm_w, rows, labels, inv_rows, eps, lm_owner=None):
        # lm_owner: 1-tuple holding the Parameter whose storage ``lm_w`` views (a tuple so autograd does not treat it
        # as an input).  The resident transposed copy for the dgrad is cached ON that object; a fresh
        # ``weight.detach()`` view per step would miss the cache and re-transpose 788 MB every step.
        ctx.lm_owner = lm_owner[0] if lm_owner else None
        B, S, d = hidden.shape
        x2 = hidden.reshape(B * S, d)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        need = ctx.needs_input_grad[0]
        acc = torch.zeros(2, dtype=torch.float32, device=hidden.device)
        n = rows.numel() if rows is not None else x2.shape[0]
        if n == 0:
            ctx.empty = True
            ctx.shape = hidden.shape
            return acc[0]
        # rows None: ``hidden`` already holds exactly the listed rows (row-restricted last decoder layer)
        xs = ops.gather_rows(x2, None, rows, d) if rows is not None else x2
        hn, rstd = ops.rmsnorm_fwd(xs, norm_w, eps, save_stats=need)
        logits = ops.gemm(hn, lm_w)
        ops.cross_entropy(logits, labels, acc, logits if need else None)
        ctx.empty = False
        ctx.shape = hidden.shape
        ctx.compact = rows is None
        if need:
            ctx.save_for_backward(xs, rstd, logits, norm_w, lm_w, *(() if ctx.compact else (inv_rows,)))
        return acc[0]

    @staticmethod
    def backward(ctx, g):
        B, S, d = ctx.shape
        if ctx.empty:
            return (torch.zeros(ctx.shape, dtype=BF16, device=g.device),) + (None,) * 7
        xs, rstd, dlogits, norm_w, lm_w = ctx.saved_tensors[:5]
        g32 = g.reshape(1).to(torch.float32).contiguous()
        d_hn = dgrad(dlogits, lm_w, owner=ctx.lm_owner, alpha_dev=g32)
        d_xs = ops.rmsnorm_bwd(d_hn, xs, norm_w, rstd)
        if ctx.compact:
            return (d_xs.view(B, S, d),) + (None,) * 7
        # scatter the listed rows back through the inverse map (-1 = no gradient: zero row)
        dx = ops.gather_rows(d_xs, None, ctx.saved_tensors[5], d)
        return (dx.view(B, S, d),) + (None,) * 7


class CrossEntropySumFn(torch.autograd.Function):
    """Token-sum cross entropy over FULL bf16 logits [rows, V] with int32 labels (-100 ignored):
    the drop-in tail for a decoder that only exposes ``["logits"]`` (ref:engine:280-305)."""

    @staticmethod
    def forward(ctx, logits2d, labels):
        acc = torch.zeros(2, dtype=torch.float32, device=logits2d.device)
        need = ctx.needs_input_grad[0]
        dl = torch.empty_like(logits2d) if need else None
        ops.cross_entropy(logits2d, labels, acc, dl)
        if need:
            ctx.save_for_backward(dl)
        return acc[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return (dl.float() * g).to(BF16), None
