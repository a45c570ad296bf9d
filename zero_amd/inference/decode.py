"""Inference through the decoder with a KV cache: prefill (the training path's causal attention
kernel) + greedy decoding (single-query attention over the cache), optionally tensor parallel.

Mirrors what the reference gets from ``transformers`` in ref:inference/vlm_infer.py:259-270:
``decoder.generate(inputs_embeds=…, attention_mask=ones, do_sample=False, max_new_tokens=…,
use_cache=True)`` returns the NEW token ids only (inputs_embeds prompt), finished sequences are
filled with ``pad_token_id``, generation stops when every sequence has produced an EOS token.
No autograd here; the layers' (possibly sharded) parameters are read as they are.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .. import functional as ZF
from .. import ops
from ..distributed.tensor_parallel import TPContext

BF16 = torch.bfloat16
USE_SPLIT_ATTENTION = True   # native steps: RoPE + append + split-cache attention (False: zk_rope, append, zk_attn_decode)
USE_NATIVE_DECODE = True     # single-token steps through the C layer drivers (False: kernel by kernel from Python)


class KVCache:
    """Per layer: K and V [B, max_len, local kv heads * hd] bf16, post-RoPE keys (what
    tf DynamicCache holds, tf:llama:261-262).  All sequences of a batch share one length (the
    reference pads the prompt and passes an all-ones mask, ref:vlm_infer.py:257-263)."""

    def __init__(self, n_layers: int, B: int, max_len: int, kv_width: int, device):
        self.k = [torch.empty((B, max_len, kv_width), dtype=BF16, device=device) for _ in range(n_layers)]
        self.v = [torch.empty((B, max_len, kv_width), dtype=BF16, device=device) for _ in range(n_layers)]
        self.max_len, self.length = max_len, 0
        self.lens = torch.zeros(B, dtype=torch.int32, device=device)


def _gate_up_interleaved(gw, uw):
    """Resident interleaved [2F, d] copy for the fused SwiGLU epilogue (inference: no transpose)."""
    cache = gw.__dict__.setdefault("_zk_gu_inf", {})
    key = (gw._version, uw._version, gw.data_ptr(), uw.data_ptr())
    if cache.get("key") != key:
        cache["key"], cache["wi"] = key, ops.interleave_gate_up(gw.detach(), uw.detach())
    return cache["wi"]


@torch.no_grad()
def layer_infer(layer, x2: torch.Tensor, B: int, S: int, pos0: int, cos_t, sin_t, cache: KVCache, li: int,
                tp: TPContext) -> torch.Tensor:
    """One decoder layer (tf:llama:295-324) on rows x2 [B*S, d]; positions pos0 .. pos0+S-1."""
    a, m = layer.self_attn, layer.mlp
    _, _, hd, eps = layer.cfg_dims
    nh, nkv = a.q_proj.weight.shape[0] // hd, a.k_proj.weight.shape[0] // hd      # local (sharded) head counts
    M, d = x2.shape
    h1, _ = ops.rmsnorm_fwd(x2, layer.input_layernorm.weight, eps, save_stats=False)
    qkv = torch.empty((M, (nh + 2 * nkv) * hd), dtype=BF16, device=x2.device)
    wqkv = ZF.packed_rows(a.q_proj.weight, a.k_proj.weight, a.v_proj.weight)
    if wqkv is not None:
        ops.gemm(h1, wqkv.detach(), out=qkv)
    else:
        ops.gemm(h1, a.q_proj.weight.detach(), out=qkv[:, :nh * hd])
        ops.gemm(h1, a.k_proj.weight.detach(), out=qkv[:, nh * hd:(nh + nkv) * hd])
        ops.gemm(h1, a.v_proj.weight.detach(), out=qkv[:, (nh + nkv) * hd:])
    # row r of qkv is position pos0 + r % S: hand the kernel the table rows from pos0 on
    ops.rope_(qkv, cos_t[pos0:pos0 + S], sin_t[pos0:pos0 + S], S, nh + nkv, hd)
    q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
    cache.k[li][:, pos0:pos0 + S].copy_(k.view(B, S, nkv * hd))
    cache.v[li][:, pos0:pos0 + S].copy_(v.view(B, S, nkv * hd))
    if S > 1:
        if pos0 != 0:
            raise NotImplementedError("chunked prefill (S > 1 on a non-empty cache)")
        ao, _ = ops.attn_fwd(q, k, v, B, S, nh, nkv, hd, hd ** -0.5, True, need_lse=False)
    else:
        ao = ops.attn_decode(q, cache.k[li], cache.v[li], cache.lens, pos0 + 1, nh, nkv, hd, hd ** -0.5)
    # row-parallel projections: every rank holds a slice of the contraction; ONE rank adds the residual
    xa = tp.reduce(ops.gemm(ao, a.o_proj.weight.detach(), residual=x2 if tp.lead else None))
    h2, _ = ops.rmsnorm_fwd(xa, layer.post_attention_layernorm.weight, eps, save_stats=False)
    gw, uw = m.gate_proj.weight, m.up_proj.weight
    wgu = ZF.packed_rows(gw, uw)
    if M <= ops.GEMV_MAX_ROWS and wgu is not None:
        # decoding: stream the packed gate|up weight once (GEMV), then the small SwiGLU kernel
        hm = ops.swiglu_fwd(ops.gemm(h2, wgu.detach()), gw.shape[0])
    elif gw.shape[0] % 32 == 0:
        _, hm = ops.gemm_swiglu(h2, _gate_up_interleaved(gw, uw), keep_gu=False)   # prefill: h only
    else:
        F = gw.shape[0]
        gu = torch.empty((M, 2 * F), dtype=BF16, device=x2.device)
        ops.gemm(h2, gw.detach(), out=gu[:, :F])
        ops.gemm(h2, uw.detach(), out=gu[:, F:])
        hm = ops.swiglu_fwd(gu, F)
    return tp.reduce(ops.gemm(hm, m.down_proj.weight.detach(), residual=xa if tp.lead else None))


class DecodePlan:
    """Single-token steps through the native layer drivers (zk_llama_decode_attn / _mlp): one
    zk_decode_layer_t per layer (weights, cache, scratch rows: pointers that stay put for the whole
    generation), built once per cache.  Needs packed q|k|v and gate|up weights and B <= 4."""

    def __init__(self, decoder, cache: KVCache, cos_t, sin_t):
        import ctypes
        from .._lib import DecodeLayer
        model = decoder.model
        layers = model.layers[: decoder.config.num_hidden_layers]
        l0 = layers[0]
        _, _, hd, eps = l0.cfg_dims
        a0, m0 = l0.self_attn, l0.mlp
        B = cache.k[0].shape[0]
        d = a0.q_proj.weight.shape[1]
        nh, nkv, F = a0.q_proj.weight.shape[0] // hd, a0.k_proj.weight.shape[0] // hd, m0.gate_proj.weight.shape[0]
        dev = cache.k[0].device
        mk = lambda n: torch.empty((B, n), dtype=BF16, device=dev)
        self.scratch = [mk(d), mk((nh + 2 * nkv) * hd), mk(nh * hd), mk(2 * F), mk(F)]
        self.bufs = [mk(d), mk(d), mk(d)]                 # layer input / attention-half output / layer output
        # fp32 partials of the split-cache attention (shared by all layers: the stream orders them)
        ws_floats = B * nh * ((cache.max_len + 63) // 64) * (hd + 2)
        self.attn_ws = torch.empty(ws_floats, dtype=torch.float32, device=dev) if USE_SPLIT_ATTENTION else None
        self.keep = []                                    # views the structs point into
        self.structs = []
        for li, layer in enumerate(layers):
            a, m = layer.self_attn, layer.mlp
            wqkv = ZF.packed_rows(a.q_proj.weight, a.k_proj.weight, a.v_proj.weight)
            wgu = ZF.packed_rows(m.gate_proj.weight, m.up_proj.weight)
            if wqkv is None or wgu is None:
                raise ValueError("DecodePlan needs packed q|k|v and gate|up weights")
            self.keep += [wqkv, wgu]
            st = DecodeLayer(B=B, d=d, nh=nh, nkv=nkv, hd=hd, F=F, cache_rows=cache.max_len, pos=0, eps=eps,
                             ln1_w=layer.input_layernorm.weight.data_ptr(), wqkv=wqkv.data_ptr(),
                             wo=a.o_proj.weight.data_ptr(), ln2_w=layer.post_attention_layernorm.weight.data_ptr(),
                             wgu=wgu.data_ptr(), wdown=m.down_proj.weight.data_ptr(), cos_t=cos_t.data_ptr(),
                             sin_t=sin_t.data_ptr(), k_cache=cache.k[li].data_ptr(), v_cache=cache.v[li].data_ptr(),
                             lens=cache.lens.data_ptr(), h=self.scratch[0].data_ptr(), qkv=self.scratch[1].data_ptr(),
                             ao=self.scratch[2].data_ptr(), gu=self.scratch[3].data_ptr(), hm=self.scratch[4].data_ptr(),
                             attn_ws=self.attn_ws.data_ptr() if self.attn_ws is not None else None,
                             attn_ws_floats=ws_floats if self.attn_ws is not None else 0)
            self.structs.append(st)
        self._byref = [ctypes.byref(s_) for s_ in self.structs]

    @staticmethod
    def usable(decoder, B: int) -> bool:
        l0 = decoder.model.layers[0]
        a, m = l0.self_attn, l0.mlp
        return (B <= 4 and ZF.packed_rows(a.q_proj.weight, a.k_proj.weight, a.v_proj.weight) is not None
                and ZF.packed_rows(m.gate_proj.weight, m.up_proj.weight) is not None
                and a.o_proj.weight.is_contiguous() and m.down_proj.weight.is_contiguous())

    def step(self, x2: torch.Tensor, pos: int, tp: TPContext) -> torch.Tensor:
        from .._lib import check, lib
        stream = torch.cuda.current_stream().cuda_stream
        cur, mid, nxt = self.bufs
        cur.copy_(x2)
        lead = int(tp.lead)
        for st, ref in zip(self.structs, self._byref):
            st.pos = pos
            check(lib.zk_llama_decode_attn(ref, cur.data_ptr(), mid.data_ptr(), lead, stream), "zk_llama_decode_attn")
            if tp.world > 1:
                tp.reduce(mid)
            check(lib.zk_llama_decode_mlp(ref, mid.data_ptr(), nxt.data_ptr(), lead, stream), "zk_llama_decode_mlp")
            if tp.world > 1:
                tp.reduce(nxt)
            cur, nxt = nxt, cur
        self.bufs = [cur, mid, nxt]
        return cur


@torch.no_grad()
def forward_cached(decoder, x: torch.Tensor, cache: KVCache, tp: Optional[TPContext] = None,
                   last_only: bool = True) -> torch.Tensor:
    """x: [B, S, d] embeddings of the NEXT S positions.  Returns logits [B, 1 or S, V] (bf16)."""
    tp = tp or TPContext()
    model = decoder.model
    B, S, d = x.shape
    pos0 = cache.length
    if pos0 + S > cache.max_len:
        raise ValueError(f"KV cache too small: {pos0 + S} > {cache.max_len}")
    cos_t, sin_t = model.rotary_emb.tables(cache.max_len, x.device)
    cache.lens.fill_(pos0 + S)
    h = x.reshape(B * S, d).to(BF16).contiguous()
    if S == 1 and USE_NATIVE_DECODE and DecodePlan.usable(decoder, B):
        plan = getattr(cache, "_plan", None)
        if plan is None:
            plan = cache._plan = DecodePlan(decoder, cache, cos_t, sin_t)
        h = plan.step(h, pos0, tp)
    else:
        for li, layer in enumerate(model.layers[: decoder.config.num_hidden_layers]):
            h = layer_infer(layer, h, B, S, pos0, cos_t, sin_t, cache, li, tp)
    cache.length = pos0 + S
    if last_only:
        h = h.view(B, S, d)[:, -1].clone(memory_format=torch.contiguous_format)
    hn, _ = ops.rmsnorm_fwd(h, model.norm.weight, model.norm.variance_epsilon, save_stats=False)
    logits = ops.gemm(hn, decoder.lm_head.weight.detach())
    return logits.view(B, 1 if last_only else S, -1)


@torch.no_grad()
def generate(decoder, inputs_embeds: torch.Tensor, attention_mask=None, do_sample: bool = False,
             max_new_tokens: int = 20, use_cache: bool = True, eos_token_id=None, pad_token_id=None,
             tp: Optional[TPContext] = None) -> torch.Tensor:
    """Greedy decoding from prompt embeddings; returns the new token ids [B, <= max_new_tokens] (int64)."""
    if do_sample:
        raise NotImplementedError("sampling: the reference decodes greedily (ref:vlm_infer.py:267)")
    if attention_mask is not None and not bool((attention_mask != 0).all()):
        raise NotImplementedError("attention masks with holes: the reference passes all ones (ref:vlm_infer.py:260-263)")
    gc = getattr(decoder, "generation_config", None)
    eos = eos_token_id if eos_token_id is not None else getattr(gc, "eos_token_id", None)
    pad = pad_token_id if pad_token_id is not None else getattr(gc, "pad_token_id", None)
    eos_set = set() if eos is None else set(eos if isinstance(eos, (list, tuple)) else [eos])
    if pad is None and eos_set:
        pad = sorted(eos_set)[0]
    B, S, d = inputs_embeds.shape
    a0 = decoder.model.layers[0].self_attn
    cache = KVCache(len(decoder.model.layers), B, S + max_new_tokens, a0.k_proj.weight.shape[0], inputs_embeds.device)
    wte = decoder.model.get_input_embeddings()
    out: List[torch.Tensor] = []
    done = torch.zeros(B, dtype=torch.bool, device=inputs_embeds.device)
    x = inputs_embeds
    for _ in range(max_new_tokens):
        logits = forward_cached(decoder, x, cache, tp)                    # [B, 1, V]
        nxt = logits[:, 0].float().argmax(-1)                             # greedy (tf: argmax of the fp32 scores)
        if eos_set:
            nxt = torch.where(done, torch.full_like(nxt, pad), nxt)
        out.append(nxt)
        if eos_set:
            hit = torch.zeros_like(done)
            for e in eos_set:
                hit |= nxt == e
            done |= hit
            if bool(done.all()):
                break
        x = wte(nxt.view(B, 1))
    return torch.stack(out, dim=1)
