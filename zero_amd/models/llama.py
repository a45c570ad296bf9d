"""Llama-3 causal LM (surrogate decoder) on HIP kernels.

Drop-in for ``transformers.LlamaForCausalLM`` as the reference uses it
(ref:build_decoder.py:71-146,268-362; ref:methods/engine_ar_instruction_llama.py:186,276-280):
``decoder(inputs_embeds=…, use_cache=False)["logits"]``, ``decoder(input_ids=…)``,
``decoder.model.get_input_embeddings()``, ``decoder.model.layers`` (a ModuleList that
supports index-assign and pop, which the surrogate surgery needs), ``decoder.config``,
``decoder.lm_head`` (tied to ``embed_tokens`` when ``tie_word_embeddings``).
The decoder is frozen on this path, so its parameters are created with
requires_grad=False and the layer node provides dgrad only.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import os

import torch
import torch.nn as nn

from .. import functional as ZF
from .. import ops
from .layers import Embedding, Linear, RMSNorm, pack_params


@dataclass
class LlamaConfig:
    vocab_size: int = 128256
    hidden_size: int = 3072
    intermediate_size: int = 8192
    num_hidden_layers: int = 28
    num_attention_heads: int = 24
    num_key_value_heads: int = 8
    head_dim: int = 128
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_scaling: Dict = field(default_factory=lambda: dict(rope_type="llama3", factor=32.0, low_freq_factor=1.0,
                                                            high_freq_factor=4.0,
                                                            original_max_position_embeddings=8192))
    max_position_embeddings: int = 131072
    tie_word_embeddings: bool = True
    hidden_act: str = "silu"


def llama3_inv_freq(cfg: LlamaConfig) -> torch.Tensor:
    """tf:modeling_rope_utils.py:580-662 (llama3 wavelength smoothing), fp32 on the host."""
    dim = cfg.head_dim
    inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.int64).to(torch.float) / dim))
    rs = cfg.rope_scaling
    if not rs or rs.get("rope_type", "default") == "default":
        return inv_freq
    factor, low, high = rs["factor"], rs["low_freq_factor"], rs["high_freq_factor"]
    old = rs["original_max_position_embeddings"]
    low_wl, high_wl = old / low, old / high
    wavelen = 2 * math.pi / inv_freq
    inv_l = torch.where(wavelen > low_wl, inv_freq / factor, inv_freq)
    smooth = (old / wavelen - low) / (high - low)
    smoothed = (1 - smooth) * inv_l / factor + smooth * inv_l
    medium = ~(wavelen < high_wl) * ~(wavelen > low_wl)
    return torch.where(medium, smoothed, inv_l)


TAIL_ROWS = os.environ.get("ZERO_AMD_TAIL_ROWS", "1") != "0"      # A/B switch: 0 = the last layer computes every row
TAIL_ROWS_USED = 0          # forwards that took the row-restricted tail (tests)


class LlamaRotaryEmbedding(nn.Module):
    """cos/sin tables [S, head_dim/2] in fp32 holding the bf16-rounded values (tf:llama:113-127:
    fp32 outer product, cos/sin, cast to the activation dtype).  Built on the host once per
    sequence length and cached on the device."""

    def __init__(self, cfg: LlamaConfig):
        super().__init__()
        self.cfg = cfg
        self._cache: Dict = {}

    def tables(self, S: int, device):
        key = (S, str(device))
        if key not in self._cache:
            inv = llama3_inv_freq(self.cfg)
            pos = torch.arange(S, dtype=torch.float32)
            freqs = (inv[:, None] @ pos[None, :]).transpose(0, 1)          # [S, hd/2]
            cos = freqs.cos().to(torch.bfloat16).to(torch.float32)
            sin = freqs.sin().to(torch.bfloat16).to(torch.float32)
            self._cache = {key: (cos.contiguous().to(device), sin.contiguous().to(device))}
        return self._cache[key]


class LlamaAttention(nn.Module):
    def __init__(self, cfg: LlamaConfig, device=None):
        super().__init__()
        d, hd = cfg.hidden_size, cfg.head_dim
        self.q_proj = Linear(d, cfg.num_attention_heads * hd, bias=False, device=device)
        self.k_proj = Linear(d, cfg.num_key_value_heads * hd, bias=False, device=device)
        self.v_proj = Linear(d, cfg.num_key_value_heads * hd, bias=False, device=device)
        self.o_proj = Linear(cfg.num_attention_heads * hd, d, bias=False, device=device)
        pack_params([self.q_proj.weight, self.k_proj.weight, self.v_proj.weight])


class LlamaMLP(nn.Module):
    def __init__(self, cfg: LlamaConfig, device=None):
        super().__init__()
        self.gate_proj = Linear(cfg.hidden_size, cfg.intermediate_size, bias=False, device=device)
        self.up_proj = Linear(cfg.hidden_size, cfg.intermediate_size, bias=False, device=device)
        self.down_proj = Linear(cfg.intermediate_size, cfg.hidden_size, bias=False, device=device)
        pack_params([self.gate_proj.weight, self.up_proj.weight])


class LlamaDecoderLayer(nn.Module):
    """tf:llama:295-324 as ONE autograd node (functional.DecoderLayerFn)."""

    def __init__(self, cfg: LlamaConfig, layer_idx: int = 0, device=None):
        super().__init__()
        self.cfg_dims = (cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.rms_norm_eps)
        self.layer_idx = layer_idx
        self.self_attn = LlamaAttention(cfg, device)
        self.mlp = LlamaMLP(cfg, device)
        self.input_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, device=device)
        self.post_attention_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, device=device)

    def forward(self, hidden_states, position_embeddings=None, tail_rows=None, **kwargs):
        cos_t, sin_t = position_embeddings
        a, m = self.self_attn, self.mlp
        rows, inv = tail_rows if tail_rows is not None else (None, None)
        return ZF.DecoderLayerFn.apply(
            hidden_states, self.input_layernorm.weight, a.q_proj.weight, a.k_proj.weight, a.v_proj.weight,
            a.o_proj.weight, self.post_attention_layernorm.weight, m.gate_proj.weight, m.up_proj.weight,
            m.down_proj.weight, cos_t, sin_t, self.cfg_dims, rows, inv)


class LlamaModel(nn.Module):
    def __init__(self, cfg: LlamaConfig, device=None):
        super().__init__()
        self.config = cfg
        self.embed_tokens = Embedding(cfg.vocab_size, cfg.hidden_size, device=device)
        self.layers = nn.ModuleList([LlamaDecoderLayer(cfg, i, device) for i in range(cfg.num_hidden_layers)])
        self.norm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps, device=device)
        self.rotary_emb = LlamaRotaryEmbedding(cfg)

    def get_input_embeddings(self):
        return self.embed_tokens

    def run_layers(self, inputs_embeds, tail_rows=None):
        """``tail_rows`` = (rows, inv_rows): the only positions whose final hidden state the caller reads (the rows a
        loss is formed on).  If the last layer is frozen and fewer than half of the positions are listed, that layer
        computes its row-wise half (o-projection, MLP) for the listed rows only and the result is [1, R, d] — check
        ``.shape`` (functional.DecoderLayerFn); otherwise, and without the argument, [B, S, d] as ever."""
        B, S, _ = inputs_embeds.shape
        pe = self.rotary_emb.tables(S, inputs_embeds.device)
        x = inputs_embeds
        # like tf:llama:402 the loop covers whatever the ModuleList holds after the surgery
        layers = self.layers[: self.config.num_hidden_layers]
        for i, layer in enumerate(layers):
            tail = None
            if (tail_rows is not None and TAIL_ROWS and i == len(layers) - 1 and 0 < tail_rows[0].numel() * 2 <= B * S
                    and not any(p.requires_grad for p in layer.parameters())):
                tail = tail_rows
                global TAIL_ROWS_USED
                TAIL_ROWS_USED += 1
            x = layer(x, position_embeddings=pe, tail_rows=tail)
        return x

    def forward(self, input_ids=None, inputs_embeds=None, **kwargs):
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
        if inputs_embeds is None:
            inputs_embeds = self.embed_tokens(input_ids)
        return self.norm(self.run_layers(inputs_embeds.to(torch.bfloat16)))


class CausalLMOutput(dict):
    """dict with attribute access, like transformers' ModelOutput (engine:280 indexes ["logits"])."""
    __getattr__ = dict.get


class LlamaForCausalLM(nn.Module):
    def __init__(self, config: LlamaConfig, device=None):
        super().__init__()
        self.config = config
        self.model = LlamaModel(config, device)
        self.lm_head = Linear(config.hidden_size, config.vocab_size, bias=False, device=device)
        if config.tie_word_embeddings:
            self.lm_head.weight = self.model.embed_tokens.weight            # tf:llama:422
        for p in self.parameters():
            p.requires_grad = False                                         # frozen surrogate (see module docstring)
        import types
        self.generation_config = types.SimpleNamespace(pad_token_id=None, eos_token_id=None)   # ref:vlm_infer.py:213

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def generate(self, inputs_embeds=None, attention_mask=None, do_sample=False, max_new_tokens=20,
                 use_cache=True, **kwargs):
        """Greedy decoding with a KV cache (ref:inference/vlm_infer.py:259-270); new token ids only."""
        from ..inference.decode import generate
        return generate(self, inputs_embeds, attention_mask=attention_mask, do_sample=do_sample,
                        max_new_tokens=max_new_tokens, use_cache=use_cache, **kwargs)

    def forward(self, input_ids=None, inputs_embeds=None, use_cache=False, **kwargs):
        if use_cache:
            raise NotImplementedError("forward(use_cache=True): use .generate() / inference.decode.forward_cached")
        h = self.model(input_ids=input_ids, inputs_embeds=inputs_embeds)
        logits = ZF.LinearFn.apply(h, self.lm_head.weight, None, ops.EPI_NONE)
        return CausalLMOutput(logits=logits)
