"""CPU oracle for the encoder + adapter + surrogate-decoder training step.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (``zero_amd/``) may
import this module; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker / the timed
CPU baseline.

What it is: a restatement, in plain ``torch`` CPU ops (no ``transformers``
import, no reference import), of the arithmetic the reference's hot path
dispatches.  ``tf:`` citations point into the third-party dependency
``transformers`` (unpinned in /root/reference/requirements.txt:1; paper used
4.45.2, README.md:37; the copy used to pin this file is 5.15.0).

Pinning: ``tests/golden/make_golden.py`` runs the real ``transformers``
``CLIPVisionModel`` / ``LlamaForCausalLM`` (built from local configs, seeded
random weights) through the reference's own ``Zero._forward``
(/root/reference/src/models/adapter.py:66-116, imported) and
``reduce_sum`` (/root/reference/src/distributed/torchshard.py:36-41, imported)
and commits the results as ``tests/golden/*.safetensors``;
``tests/test_oracle_golden.py`` checks this file against them.  The reference
itself holds no tests, golden vectors or fixtures for this path,
so those generated vectors are the only pin.  The tokenizer / chat-template
stage (engine_ar_instruction_llama.py:59-183) is NOT pinned (no tokenizer
offline): parity unpinned for that stage; the oracle starts from token-id
lists.

Parameter names follow the transformers-4.45 state-dict layout the reference
targets (``encoder.vision_model.…``, ``decoder.model.layers.N.…``,
``adapter.{0,2}.…``, ``embd_soi``, ``embd_eoi``; SURVEY.md §5 checkpoint row).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

IGNORE_INDEX = -100  # engine_ar_instruction_llama.py:22

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------
# configs (public config.json constants carried locally)
# --------------------------------------------------------------------------
@dataclass
class EncoderCfg:
    hidden: int = 1024
    heads: int = 16
    intermediate: int = 4096
    layers: int = 24
    image: int = 336
    patch: int = 14
    eps: float = 1e-5

    @property
    def tokens(self) -> int:
        return (self.image // self.patch) ** 2 + 1


@dataclass
class DecoderCfg:
    hidden: int = 3072
    heads: int = 24
    kv_heads: int = 8
    head_dim: int = 128
    intermediate: int = 8192
    layers: int = 28           # layers of the ORIGINAL model, before surgery
    vocab: int = 128256
    rms_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_factor: float = 32.0
    rope_low: float = 1.0
    rope_high: float = 4.0
    rope_orig_ctx: int = 8192
    tie: bool = True
    translators_range: Tuple[Tuple[int, ...], ...] = (tuple(range(16, 27)),)
    num_translators: int = 1

    @property
    def surrogate_layers(self) -> int:
        return len(surrogate_layer_map(self.layers, self.translators_range, self.num_translators))


CLIP_L_336 = EncoderCfg()
LLAMA32_3B = DecoderCfg()
LLAMA31_8B = DecoderCfg(hidden=4096, heads=32, kv_heads=8, head_dim=128, intermediate=14336,
                        layers=32, rope_factor=8.0, tie=False,
                        translators_range=(tuple(range(17, 31)),))


# --------------------------------------------------------------------------
# surrogate surgery  (build_decoder.py:268-362)
# --------------------------------------------------------------------------
def surrogate_layer_map(num_layers: int, translators_range, num_translators: int) -> List[int]:
    """Return, for each surviving layer slot, the ORIGINAL layer index it holds.

    Restates build_decoder.py:283-330: for every range the first
    ``num_translators`` layers are re-classed in place (kept), the rest of the
    range is ``ModuleList.pop``-ed.  The reference pops ``layer_id - offset``
    with the offset growing by one per pop, i.e. always the slot that the
    original ``layer_id`` currently occupies.
    """
    slots = list(range(num_layers))
    ranges = [list(r) for r in translators_range]          # deepcopy, build_decoder.py:283
    for range_idx, trans_range in enumerate(ranges):
        trans_range.sort()
        assert all(b - a == 1 for a, b in zip(trans_range, trans_range[1:])), \
            "translators_range should be consecutive without skipping layers"
        assert all(v >= 0 for v in trans_range)
        assert 0 <= min(trans_range) < max(trans_range) < num_layers
        assert num_translators <= len(trans_range)
        index_offset = 0
        if range_idx > 0:
            # lengths of the ALREADY-MUTATED earlier lists = layers removed so far
            index_offset += sum(len(ranges[i]) for i in range(range_idx))
        # build_decoder.py:305-316: entry i of the shrinking list is re-classed
        # (kept) and popped from the to-remove list.
        for i in range(num_translators):
            trans_range.pop(i)
        for layer_id in trans_range:                       # build_decoder.py:320-330
            slots.pop(layer_id - index_offset)
            index_offset += 1
    return slots


def apply_surgery(params: Params, dec: DecoderCfg, prefix: str = "decoder.model.layers.") -> Params:
    """Renumber ``decoder.model.layers.N.*`` keys the way ModuleList.pop does."""
    keep = surrogate_layer_map(dec.layers, dec.translators_range, dec.num_translators)
    out: Params = {}
    for k, v in params.items():
        if not k.startswith(prefix):
            out[k] = v
            continue
        rest = k[len(prefix):]
        n_str, tail = rest.split(".", 1)
        n = int(n_str)
        if n in keep:
            out[f"{prefix}{keep.index(n)}.{tail}"] = v
    return out


# --------------------------------------------------------------------------
# synthetic weights (shared by the golden generator, the tests and bench.py)
# --------------------------------------------------------------------------
def encoder_param_shapes(enc: EncoderCfg, prefix="encoder.vision_model.") -> Dict[str, Tuple[int, ...]]:
    D, Fi = enc.hidden, enc.intermediate
    s: Dict[str, Tuple[int, ...]] = {
        prefix + "embeddings.class_embedding": (D,),
        prefix + "embeddings.patch_embedding.weight": (D, 3, enc.patch, enc.patch),
        prefix + "embeddings.position_embedding.weight": (enc.tokens, D),
        prefix + "pre_layrnorm.weight": (D,), prefix + "pre_layrnorm.bias": (D,),
        prefix + "post_layernorm.weight": (D,), prefix + "post_layernorm.bias": (D,),
    }
    for i in range(enc.layers):
        p = f"{prefix}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (D, D)
            s[p + f"self_attn.{n}.bias"] = (D,)
        s[p + "layer_norm1.weight"] = (D,); s[p + "layer_norm1.bias"] = (D,)
        s[p + "layer_norm2.weight"] = (D,); s[p + "layer_norm2.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (Fi, D); s[p + "mlp.fc1.bias"] = (Fi,)
        s[p + "mlp.fc2.weight"] = (D, Fi); s[p + "mlp.fc2.bias"] = (D,)
    return s


def decoder_param_shapes(dec: DecoderCfg, n_layers: Optional[int] = None,
                         prefix="decoder.") -> Dict[str, Tuple[int, ...]]:
    d, f = dec.hidden, dec.intermediate
    L = dec.layers if n_layers is None else n_layers
    s: Dict[str, Tuple[int, ...]] = {prefix + "model.embed_tokens.weight": (dec.vocab, d)}
    for i in range(L):
        p = f"{prefix}model.layers.{i}."
        s[p + "input_layernorm.weight"] = (d,)
        s[p + "self_attn.q_proj.weight"] = (dec.heads * dec.head_dim, d)
        s[p + "self_attn.k_proj.weight"] = (dec.kv_heads * dec.head_dim, d)
        s[p + "self_attn.v_proj.weight"] = (dec.kv_heads * dec.head_dim, d)
        s[p + "self_attn.o_proj.weight"] = (d, dec.heads * dec.head_dim)
        s[p + "post_attention_layernorm.weight"] = (d,)
        s[p + "mlp.gate_proj.weight"] = (f, d)
        s[p + "mlp.up_proj.weight"] = (f, d)
        s[p + "mlp.down_proj.weight"] = (d, f)
    s[prefix + "model.norm.weight"] = (d,)
    if not dec.tie:
        s[prefix + "lm_head.weight"] = (dec.vocab, d)
    return s


def adapter_param_shapes(num_embd: int, enc_hidden: int = 1024) -> Dict[str, Tuple[int, ...]]:
    # adapter.py:34-38 (depth 2) and :42-43
    return {
        "adapter.0.weight": (num_embd, enc_hidden), "adapter.0.bias": (num_embd,),
        "adapter.2.weight": (num_embd, num_embd), "adapter.2.bias": (num_embd,),
        "embd_soi": (1, num_embd), "embd_eoi": (1, num_embd),
    }


def make_params(enc: EncoderCfg, dec: DecoderCfg, seed: int, *, surrogate: bool = True,
                dtype=torch.float32, round_to_bf16: bool = True) -> Params:
    """Seeded synthetic weights, drawn name by name in sorted-name order.

    Matrices ~ N(0, 0.02) (plus a 1/sqrt(fan_in) floor so activations keep unit
    scale at tiny widths), norm weights ~ 1 + N(0, 0.1), biases ~ N(0, 0.02),
    soi/eoi ~ N(0, 1/d) (adapter.py:56-60).  With ``round_to_bf16`` every value
    is representable in bf16, so the fp32 oracle and the bf16 device path see
    the SAME weights.  When ``surrogate`` the decoder is built at its original
    depth and then cut by ``apply_surgery``.
    """
    shapes: Dict[str, Tuple[int, ...]] = {}
    shapes.update(encoder_param_shapes(enc))
    shapes.update(decoder_param_shapes(dec))
    shapes.update(adapter_param_shapes(dec.hidden, enc.hidden))
    g = torch.Generator(device="cpu").manual_seed(seed)
    out: Params = {}
    for name in sorted(shapes):
        shp = shapes[name]
        if name in ("embd_soi", "embd_eoi"):
            t = torch.randn(shp, generator=g) / math.sqrt(shp[1])
        elif "norm" in name and name.endswith(".weight"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif name.endswith(".bias") or name.endswith("class_embedding"):
            t = 0.02 * torch.randn(shp, generator=g)
        elif name.endswith("embed_tokens.weight") or name.endswith("position_embedding.weight"):
            t = 0.02 * torch.randn(shp, generator=g)
        else:
            fan_in = 1
            for s_ in shp[1:]:
                fan_in *= s_
            std = max(0.02, 0.5 / math.sqrt(fan_in))
            t = std * torch.randn(shp, generator=g)
        if round_to_bf16:
            t = t.to(torch.bfloat16).to(torch.float32)
        out[name] = t.to(dtype)
    if surrogate:
        out = apply_surgery(out, dec)
    return out


def translator_layer_ids(translators_range, num_translators: int) -> List[int]:
    """Translator indices after the surgery (configs/zero/train_llama.py:290-301)."""
    ids: List[int] = []
    for i, r in enumerate(translators_range):
        o = 0
        if i > 0:
            o += sum(len(translators_range[j]) for j in range(i)) - 1
        ids += [n - o for n in list(r)[:num_translators]]
    return ids


def trainable_names(params: Params, encoder_ft_layers_after: int = 16, stage: str = "finetune-encoder",
                    translator_ids: Sequence[int] = ()) -> List[str]:
    """Trainable set by stage (configs/zero/train_llama.py:259-323): the adapter and soi/eoi always;
    "finetune-encoder": post_layernorm + encoder layers >= encoder_ft_layers_after;
    "pretrain": post_layernorm + the translator decoder layers."""
    names = []
    for n in params:
        req = False
        if "adapter" in n:
            req = True
        if "embd_soi" in n or "embd_eoi" in n:
            req = True
        if stage == "pretrain":
            if "vision_model" in n and "post_layernorm" in n:
                req = True
            if "decoder" in n and "layers" in n:
                parts = n.split(".")
                if int(parts[parts.index("layers") + 1]) in translator_ids:
                    req = True
        elif "vision_model" in n:
            if "post_layernorm" in n:
                req = True
            if "layers" in n:
                parts = n.split(".")
                layer_id = int(parts[parts.index("layers") + 1])
                if layer_id >= encoder_ft_layers_after:
                    req = True
        if req:
            names.append(n)
    return names


# --------------------------------------------------------------------------
# encoder (tf:models/clip/modeling_clip.py)
# --------------------------------------------------------------------------
def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    # tf:activations.py:117-123
    return x * torch.sigmoid(1.702 * x)


def clip_embeddings(p: Params, pixel_values: torch.Tensor, enc: EncoderCfg,
                    pre="encoder.vision_model.") -> torch.Tensor:
    # tf:clip:202-218 — conv k=s=patch, no bias; CLS prepended; learned pos added.
    w = p[pre + "embeddings.patch_embedding.weight"]
    x = F.conv2d(pixel_values.to(w.dtype), w, bias=None, stride=enc.patch)
    x = x.flatten(2).transpose(1, 2)
    cls = p[pre + "embeddings.class_embedding"].expand(x.shape[0], 1, -1)
    x = torch.cat([cls, x], dim=1)
    return x + p[pre + "embeddings.position_embedding.weight"].unsqueeze(0)


def clip_layer(p: Params, x: torch.Tensor, i: int, enc: EncoderCfg,
               pre="encoder.vision_model.") -> torch.Tensor:
    # tf:clip:362-383 (pre-LN blocks), :298-335 (attention), :346-350 (MLP)
    lp = f"{pre}encoder.layers.{i}."
    B, T, D = x.shape
    H = enc.heads
    hd = D // H
    r = x
    h = F.layer_norm(x, (D,), p[lp + "layer_norm1.weight"], p[lp + "layer_norm1.bias"], enc.eps)
    q = F.linear(h, p[lp + "self_attn.q_proj.weight"], p[lp + "self_attn.q_proj.bias"])
    k = F.linear(h, p[lp + "self_attn.k_proj.weight"], p[lp + "self_attn.k_proj.bias"])
    v = F.linear(h, p[lp + "self_attn.v_proj.weight"], p[lp + "self_attn.v_proj.bias"])
    q = q.view(B, T, H, hd).transpose(1, 2)
    k = k.view(B, T, H, hd).transpose(1, 2)
    v = v.view(B, T, H, hd).transpose(1, 2)
    # tf:integrations/sdpa_attention.py:154-163, is_causal False (tf:clip:291), no mask
    a = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0,
                                       is_causal=False, scale=hd ** -0.5)
    a = a.transpose(1, 2).reshape(B, T, D)
    a = F.linear(a, p[lp + "self_attn.out_proj.weight"], p[lp + "self_attn.out_proj.bias"])
    x = r + a
    r = x
    h = F.layer_norm(x, (D,), p[lp + "layer_norm2.weight"], p[lp + "layer_norm2.bias"], enc.eps)
    h = F.linear(h, p[lp + "mlp.fc1.weight"], p[lp + "mlp.fc1.bias"])
    h = quick_gelu(h)
    h = F.linear(h, p[lp + "mlp.fc2.weight"], p[lp + "mlp.fc2.bias"])
    return r + h


def clip_vision_forward(p: Params, pixel_values: torch.Tensor, enc: EncoderCfg,
                        pre="encoder.vision_model.") -> List[torch.Tensor]:
    """Returns hidden_states tuple as HF does with output_hidden_states=True
    (entry 0 = after pre_layrnorm, entry -1 = last_hidden_state); tf:clip:641-649."""
    x = clip_embeddings(p, pixel_values, enc, pre)
    D = enc.hidden
    x = F.layer_norm(x, (D,), p[pre + "pre_layrnorm.weight"], p[pre + "pre_layrnorm.bias"], enc.eps)
    hs = [x]
    for i in range(enc.layers):
        x = clip_layer(p, x, i, enc, pre)
        hs.append(x)
    return hs


def zero_forward(p: Params, xs: Sequence[torch.Tensor], enc: EncoderCfg, *,
                 shave_last_n: int = -1, ignore_cls: bool = False) -> List[torch.Tensor]:
    """Zero._forward, adapter.py:66-116."""
    split_sizes = [x.shape[0] for x in xs]
    x = torch.cat(list(xs), dim=0)
    hs = clip_vision_forward(p, x, enc)
    x = hs[-shave_last_n] if shave_last_n > 0 else hs[-1]          # adapter.py:85-88
    if ignore_cls:
        x = x[:, 1:, :]                                            # adapter.py:91-92
    pre = "encoder.vision_model."
    x = F.layer_norm(x, (enc.hidden,), p[pre + "post_layernorm.weight"],
                     p[pre + "post_layernorm.bias"], enc.eps)      # adapter.py:94 (ALL tokens)
    x = F.linear(x, p["adapter.0.weight"], p["adapter.0.bias"])    # adapter.py:95
    x = F.gelu(x)                                                  # nn.GELU() = exact erf
    x = F.linear(x, p["adapter.2.weight"], p["adapter.2.bias"])
    outs = []
    for piece in torch.split(x, split_sizes, dim=0):               # adapter.py:107-113
        piece = piece.reshape(-1, piece.shape[-1])
        outs.append(torch.cat([p["embd_soi"], piece, p["embd_eoi"]], dim=0))
    return outs


# --------------------------------------------------------------------------
# decoder (tf:models/llama/modeling_llama.py)
# --------------------------------------------------------------------------
def llama3_inv_freq(dec: DecoderCfg) -> torch.Tensor:
    """tf:modeling_rope_utils.py:580-662 (llama3 wavelength smoothing), fp32."""
    dim = dec.head_dim
    inv_freq = 1.0 / (dec.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.int64).to(torch.float) / dim))
    low_wl = dec.rope_orig_ctx / dec.rope_low
    high_wl = dec.rope_orig_ctx / dec.rope_high
    wavelen = 2 * math.pi / inv_freq
    inv_l = torch.where(wavelen > low_wl, inv_freq / dec.rope_factor, inv_freq)
    smooth = (dec.rope_orig_ctx / wavelen - dec.rope_low) / (dec.rope_high - dec.rope_low)
    smoothed = (1 - smooth) * inv_l / dec.rope_factor + smooth * inv_l
    medium = ~(wavelen < high_wl) * ~(wavelen > low_wl)
    return torch.where(medium, smoothed, inv_l)


def rope_cos_sin(dec: DecoderCfg, S: int, dtype) -> Tuple[torch.Tensor, torch.Tensor]:
    """tf:llama:113-127 — fp32 outer product, cat(freqs,freqs), cos/sin, cast."""
    inv = llama3_inv_freq(dec)
    pos = torch.arange(S, dtype=torch.float32)
    freqs = (inv[:, None] @ pos[None, :]).transpose(0, 1)          # [S, hd/2]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    # tf:llama:130-134
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    # tf:llama:62-67 — fp32 stats, cast back to input dtype BEFORE the weight multiply.
    dt = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return w * xf.to(dt)


def llama_layer(p: Params, x: torch.Tensor, i: int, dec: DecoderCfg, cos, sin,
                pre="decoder.model.layers.") -> torch.Tensor:
    # tf:llama:295-324 (layer), :243-281 (attention), :174-176 (MLP)
    lp = f"{pre}{i}."
    B, S, d = x.shape
    nh, nkv, hd = dec.heads, dec.kv_heads, dec.head_dim
    r = x
    h = rms_norm(x, p[lp + "input_layernorm.weight"], dec.rms_eps)
    q = F.linear(h, p[lp + "self_attn.q_proj.weight"]).view(B, S, nh, hd).transpose(1, 2)
    k = F.linear(h, p[lp + "self_attn.k_proj.weight"]).view(B, S, nkv, hd).transpose(1, 2)
    v = F.linear(h, p[lp + "self_attn.v_proj.weight"]).view(B, S, nkv, hd).transpose(1, 2)
    c = cos[None, None]
    s = sin[None, None]
    q = (q * c) + (rotate_half(q) * s)                             # tf:llama:137-160
    k = (k * c) + (rotate_half(k) * s)
    # tf:integrations/sdpa_attention.py: no mask given (engine:276-279) -> is_causal, GQA
    rep = nh // nkv
    k = k[:, :, None].expand(B, nkv, rep, S, hd).reshape(B, nh, S, hd)
    v = v[:, :, None].expand(B, nkv, rep, S, hd).reshape(B, nh, S, hd)
    a = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0,
                                       is_causal=S > 1, scale=hd ** -0.5)
    a = a.transpose(1, 2).reshape(B, S, nh * hd)
    x = r + F.linear(a, p[lp + "self_attn.o_proj.weight"])
    r = x
    h = rms_norm(x, p[lp + "post_attention_layernorm.weight"], dec.rms_eps)
    g = F.linear(h, p[lp + "mlp.gate_proj.weight"])
    u = F.linear(h, p[lp + "mlp.up_proj.weight"])
    h = F.linear(F.silu(g) * u, p[lp + "mlp.down_proj.weight"])
    return r + h


def count_decoder_layers(p: Params, pre="decoder.model.layers.") -> int:
    ids = {int(k[len(pre):].split(".", 1)[0]) for k in p if k.startswith(pre)}
    return (max(ids) + 1) if ids else 0


def llama_forward(p: Params, inputs_embeds: torch.Tensor, dec: DecoderCfg,
                  return_hidden: bool = False):
    """LlamaForCausalLM.forward(inputs_embeds=…, use_cache=False); tf:llama:367-417,467-492."""
    B, S, d = inputs_embeds.shape
    cos, sin = rope_cos_sin(dec, S, inputs_embeds.dtype)
    x = inputs_embeds
    hs = [x]
    for i in range(count_decoder_layers(p)):
        x = llama_layer(p, x, i, dec, cos, sin)
        hs.append(x)
    x = rms_norm(x, p["decoder.model.norm.weight"], dec.rms_eps)
    w = p["decoder.model.embed_tokens.weight"] if dec.tie else p["decoder.lm_head.weight"]
    logits = F.linear(x, w)
    return (logits, hs) if return_hidden else logits


# --------------------------------------------------------------------------
# forward_hf: input assembly and loss tail (engine_ar_instruction_llama.py)
# --------------------------------------------------------------------------
def assemble_inputs(p: Params, embds_imgs: Sequence[torch.Tensor], input_ids: Sequence[Sequence[int]],
                    targets: Sequence[Sequence[int]], img_id: int, max_seq_len: int):
    """engine:185-240 — wte gathers, splice image embeddings at the first img token,
    image span targets = IGNORE_INDEX, right-pad (0.0 / IGNORE_INDEX), truncate."""
    wte = p["decoder.model.embed_tokens.weight"]
    embs, tgts = [], []
    for b, (ids, tgt) in enumerate(zip(input_ids, targets)):
        ids = list(ids); tgt = list(tgt)
        if img_id in ids:
            at = ids.index(img_id)
            n_img = embds_imgs[b].shape[0]
            ids1 = ids[:at]
            ids2 = ids[at + 1:]
            ids2 = ids2[: max_seq_len - len(ids1) - n_img]        # engine:198 (plain python slice)
            e1 = F.embedding(torch.tensor(ids1, dtype=torch.long), wte)
            e2 = F.embedding(torch.tensor(ids2, dtype=torch.long), wte)
            embs.append(torch.cat([e1, embds_imgs[b].to(wte.dtype), e2], dim=0))
            tgts.append(tgt[:at] + [IGNORE_INDEX] * n_img + tgt[at + 1:])
        else:
            embs.append(F.embedding(torch.tensor(ids, dtype=torch.long), wte))
            tgts.append(tgt)
    S = max(e.shape[0] for e in embs)
    St = max(len(t) for t in tgts)
    x = torch.zeros(len(embs), S, wte.shape[1], dtype=wte.dtype)
    y = torch.full((len(tgts), St), IGNORE_INDEX, dtype=torch.long)
    for b, (e, t) in enumerate(zip(embs, tgts)):
        x[b, : e.shape[0]] = e
        y[b, : len(t)] = torch.tensor(t, dtype=torch.long)
    return x[:, :max_seq_len, :], y[:, :max_seq_len]


def loss_tail(logits: torch.Tensor, target: torch.Tensor, pad_id: int):
    """engine:280-305 — fp32 logits, shift, pad->ignore, token-sum CE, token count."""
    logits = logits.float()[:, :-1, :].contiguous()
    target = target[:, 1:].contiguous().clone()
    target[target == pad_id] = IGNORE_INDEX
    loss = F.cross_entropy(logits.view(-1, logits.shape[-1]), target.view(-1),
                           ignore_index=IGNORE_INDEX, reduction="none")
    return loss.sum(), target.ne(IGNORE_INDEX).sum()


@dataclass
class StepOut:
    loss: torch.Tensor
    loss_sum: torch.Tensor
    num_tokens: torch.Tensor
    logits: torch.Tensor
    inputs_embeds: torch.Tensor
    target: torch.Tensor
    embds_imgs: List[torch.Tensor]
    grads: Dict[str, torch.Tensor] = field(default_factory=dict)
    dec_hidden: List[torch.Tensor] = field(default_factory=list)


def train_step(p: Params, enc: EncoderCfg, dec: DecoderCfg, imgs: Sequence[torch.Tensor],
               input_ids, targets, *, img_id: int, pad_id: int, max_seq_len: int = 4096,
               encoder_ft_layers_after: int = 16, want_grads: bool = True,
               world_tokens: Optional[int] = None, keep_hidden: bool = False,
               stage: str = "finetune-encoder") -> StepOut:
    """forward_hf (engine:30-318) from token ids on, plus loss.backward() (train.py:567).

    ``world_tokens``: the cross-rank token count after reduce_sum (engine:308-310).
    reduce_sum's backward is the identity (torchshard.py:23-25), so each rank
    back-propagates local_loss_sum / global_tokens; None = 1 rank.
    """
    tids = translator_layer_ids(dec.translators_range, dec.num_translators) if stage == "pretrain" else ()
    names = trainable_names(p, encoder_ft_layers_after, stage, tids) if want_grads else []
    q: Params = {}
    for k, v in p.items():
        if k in names:
            q[k] = v.detach().clone().requires_grad_(True)
        else:
            q[k] = v.detach()
    with torch.set_grad_enabled(want_grads):
        embds = zero_forward(q, imgs, enc)
        x, y = assemble_inputs(q, embds, input_ids, targets, img_id, max_seq_len)
        logits, hs = llama_forward(q, x, dec, return_hidden=True)
        loss_sum, ntok = loss_tail(logits, y, pad_id)
        denom = ntok if world_tokens is None else torch.tensor(world_tokens)
        loss = loss_sum / denom
    out = StepOut(loss=loss.detach(), loss_sum=loss_sum.detach(), num_tokens=ntok,
                  logits=logits.detach(), inputs_embeds=x.detach(), target=y,
                  embds_imgs=[e.detach() for e in embds],
                  dec_hidden=[h.detach() for h in hs] if keep_hidden else [])
    if want_grads:
        loss.backward()
        out.grads = {k: q[k].grad.detach() for k in names if q[k].grad is not None}
    return out


# --------------------------------------------------------------------------
# zero-shot graft inference (inference/vlm_infer.py:201-270): greedy decoding
# --------------------------------------------------------------------------
def greedy_generate(p: Params, inputs_embeds: torch.Tensor, dec: DecoderCfg, max_new_tokens: int,
                    eos_ids: Sequence[int] = (), pad_id: Optional[int] = None, return_logits: bool = False):
    """What tf generate(inputs_embeds=…, do_sample=False, use_cache=True) computes, restated without a
    cache: every step re-runs the full causal forward on the grown sequence (small cases only) and
    takes the argmax of the last position; finished sequences emit ``pad_id``; stops when all are
    finished.  Returns new ids [B, n] (and the per-step last-position logits)."""
    wte = p["decoder.model.embed_tokens.weight"]
    x = inputs_embeds
    B = x.shape[0]
    done = torch.zeros(B, dtype=torch.bool)
    out, step_logits = [], []
    with torch.no_grad():
        for _ in range(max_new_tokens):
            logits = llama_forward(p, x, dec)[:, -1].float()
            step_logits.append(logits)
            nxt = logits.argmax(-1)
            if eos_ids:
                nxt = torch.where(done, torch.full_like(nxt, pad_id if pad_id is not None else eos_ids[0]), nxt)
            out.append(nxt)
            if eos_ids:
                for e in eos_ids:
                    done |= nxt == e
                if bool(done.all()):
                    break
            x = torch.cat([x, F.embedding(nxt, wte)[:, None].to(x.dtype)], dim=1)
    toks = torch.stack(out, dim=1)
    return (toks, step_logits) if return_logits else toks


# --------------------------------------------------------------------------
# step tail: clip + AdamW  (train.py:382-387, 570-580)
# --------------------------------------------------------------------------
def clip_and_adamw(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor],
                   state: Dict[str, Dict[str, torch.Tensor]], *, lr: float, step: int,
                   beta1=0.9, beta2=0.999, eps=1e-8, wd=0.0, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_(…, max_norm) then torch.optim.AdamW.step() restated.

    Returns the total grad norm.  params are updated in place.
    """
    total = torch.sqrt(sum((g.float() ** 2).sum() for g in grads.values()))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for k, g in grads.items():
        g = (g.float() * coef)
        st = state.setdefault(k, {"m": torch.zeros_like(g), "v": torch.zeros_like(g)})
        pf = params[k].float()
        pf = pf * (1 - lr * wd)
        st["m"].mul_(beta1).add_(g, alpha=1 - beta1)
        st["v"].mul_(beta2).addcmul_(g, g, value=1 - beta2)
        bc1 = 1 - beta1 ** step
        bc2 = 1 - beta2 ** step
        denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
        pf = pf - (lr / bc1) * (st["m"] / denom)
        params[k].copy_(pf.to(params[k].dtype))
    return total


# --------------------------------------------------------------------------
# synthetic batch
# --------------------------------------------------------------------------
def synth_batch(enc: EncoderCfg, dec: DecoderCfg, B: int, seed: int, *, n_pre: int = 21,
                n_post: int = 168, n_resp: int = 128, ragged: bool = False, img_id: Optional[int] = None,
                text_only: Sequence[int] = ()):
    """Images ~N(0,1) [1,3,H,W] each; ids uniform in [0, vocab-256) (the top of the
    vocabulary is kept free for the special ids); targets = ids on the last
    ``n_resp`` positions of the suffix, IGNORE_INDEX before.  ``text_only`` lists
    sample indices that carry no image token (their zero image is still encoded,
    loader.py:204-206 / SURVEY.md Q4)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    hi = dec.vocab - 256 if dec.vocab > 512 else dec.vocab - 8
    if img_id is None:
        img_id = dec.vocab - 2
    pad_id = dec.vocab - 1
    imgs, ids_all, tgt_all = [], [], []
    for b in range(B):
        if b in text_only:
            imgs.append(torch.zeros(1, 3, enc.image, enc.image))
        else:
            imgs.append(torch.randn(1, 3, enc.image, enc.image, generator=g))
        npost = n_post
        if ragged:
            npost = int(torch.randint(max(2, n_post // 4), n_post + 1, (1,), generator=g))
        pre = torch.randint(0, hi, (n_pre,), generator=g).tolist()
        post = torch.randint(0, hi, (npost,), generator=g).tolist()
        nresp = min(n_resp, npost)
        if b in text_only:
            ids = pre + post
            tgt = [IGNORE_INDEX] * (n_pre + npost - nresp) + post[npost - nresp:]
        else:
            ids = pre + [img_id] + post
            tgt = [IGNORE_INDEX] * (n_pre + 1 + npost - nresp) + post[npost - nresp:]
        ids_all.append(ids)
        tgt_all.append(tgt)
    return imgs, ids_all, tgt_all, img_id, pad_id
