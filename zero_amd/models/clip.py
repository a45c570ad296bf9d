"""CLIP ViT vision tower on HIP kernels.

Drop-in for ``transformers.CLIPVisionModel`` as the reference uses it
(ref:build_encoder.py:15-22, ref:models/adapter.py:84-94): call signature
``encoder(pixel_values, output_hidden_states=True)`` -> object with ``.last_hidden_state`` and
``.hidden_states``; attribute ``encoder.vision_model.post_layernorm``; parameter names in
the transformers-4.45 layout (``vision_model.encoder.layers.N.self_attn.q_proj.weight`` …),
which ref:configs/zero/train_llama.py:282,312-321 string-matches and the released
checkpoints use.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import functional as ZF
from .. import ops
from .layers import LayerNorm, Linear, pack_params


@dataclass
class CLIPVisionConfig:
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    image_size: int = 336
    patch_size: int = 14
    layer_norm_eps: float = 1e-5
    hidden_act: str = "quick_gelu"
    num_channels: int = 3


@dataclass
class VisionOutput:
    last_hidden_state: torch.Tensor
    pooler_output: Optional[torch.Tensor] = None
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None


class PatchEmbedding(nn.Module):
    """Holds the conv weight [D, 3, P, P] (bias-free, tf:clip:148-154)."""

    def __init__(self, cfg: CLIPVisionConfig, device=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cfg.hidden_size, cfg.num_channels, cfg.patch_size, cfg.patch_size,
                                               device=device, dtype=torch.bfloat16).normal_(0, 0.02))


class PositionEmbedding(nn.Module):
    def __init__(self, n: int, dim: int, device=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n, dim, device=device, dtype=torch.bfloat16).normal_(0, 0.02))


class CLIPVisionEmbeddings(nn.Module):
    """tf:clip:202-218: patch conv as im2col + MFMA GEMM, CLS prepend, learned positions.
    Frozen on this path (nothing upstream of encoder layer 16 trains): no backward."""

    def __init__(self, cfg: CLIPVisionConfig, device=None):
        super().__init__()
        self.cfg = cfg
        self.class_embedding = nn.Parameter(torch.empty(cfg.hidden_size, device=device, dtype=torch.bfloat16).normal_(0, 0.02))
        self.patch_embedding = PatchEmbedding(cfg, device)
        self.num_patches = (cfg.image_size // cfg.patch_size) ** 2
        self.num_positions = self.num_patches + 1
        self.position_embedding = PositionEmbedding(self.num_positions, cfg.hidden_size, device)
        self._wpad = None
        self._wpad_version = None

    def _padded_weight(self):
        w = self.patch_embedding.weight
        K = w[0].numel()
        Kpad = (K + 63) // 64 * 64
        ver = (w.data_ptr(), w._version)
        if self._wpad is None or self._wpad_version != ver:
            wp = torch.zeros(w.shape[0], Kpad, device=w.device, dtype=torch.bfloat16)
            wp[:, :K].copy_(w.detach().reshape(w.shape[0], K))
            self._wpad, self._wpad_version = wp, ver
        return self._wpad, Kpad

    def forward(self, pixel_values):
        cfg = self.cfg
        if pixel_values.shape[-1] != cfg.image_size or pixel_values.shape[-2] != cfg.image_size:
            raise ValueError(f"Input image size ({pixel_values.shape[-2]}*{pixel_values.shape[-1]}) doesn't match "
                             f"model ({cfg.image_size}*{cfg.image_size}).")
        if torch.is_grad_enabled() and (self.patch_embedding.weight.requires_grad or self.class_embedding.requires_grad
                                        or self.position_embedding.weight.requires_grad or pixel_values.requires_grad):
            raise NotImplementedError("patch/position embedding gradients: frozen on the encoder-training path")
        px = pixel_values.to(torch.bfloat16).contiguous()
        wp, Kpad = self._padded_weight()
        cols = ops.patch_im2col(px, cfg.patch_size, Kpad)
        patch = ops.gemm(cols, wp)
        return ops.embed_finish(patch, self.class_embedding.detach(), self.position_embedding.weight.detach(),
                                px.shape[0], self.num_positions, cfg.hidden_size)


class CLIPAttention(nn.Module):
    def __init__(self, cfg: CLIPVisionConfig, device=None):
        super().__init__()
        D = cfg.hidden_size
        self.q_proj = Linear(D, D, device=device)
        self.k_proj = Linear(D, D, device=device)
        self.v_proj = Linear(D, D, device=device)
        self.out_proj = Linear(D, D, device=device)
        # q|k|v share one storage so the layer runs ONE [3D, D] projection GEMM
        pack_params([self.q_proj.weight, self.k_proj.weight, self.v_proj.weight])
        pack_params([self.q_proj.bias, self.k_proj.bias, self.v_proj.bias])


class CLIPMLP(nn.Module):
    def __init__(self, cfg: CLIPVisionConfig, device=None):
        super().__init__()
        self.fc1 = Linear(cfg.hidden_size, cfg.intermediate_size, device=device)
        self.fc2 = Linear(cfg.intermediate_size, cfg.hidden_size, device=device)


class CLIPEncoderLayer(nn.Module):
    """tf:clip:362-383 as ONE autograd node (functional.EncoderLayerFn)."""

    def __init__(self, cfg: CLIPVisionConfig, device=None):
        super().__init__()
        self.heads = cfg.num_attention_heads
        self.eps = cfg.layer_norm_eps
        self.self_attn = CLIPAttention(cfg, device)
        self.layer_norm1 = LayerNorm(cfg.hidden_size, cfg.layer_norm_eps, device=device)
        self.mlp = CLIPMLP(cfg, device)
        self.layer_norm2 = LayerNorm(cfg.hidden_size, cfg.layer_norm_eps, device=device)

    def forward(self, hidden_states, attention_mask=None, **kwargs):
        if attention_mask is not None:
            raise NotImplementedError("the CLIP vision tower is called without a mask on this path")
        a, m = self.self_attn, self.mlp
        return ZF.EncoderLayerFn.apply(
            hidden_states, self.layer_norm1.weight, self.layer_norm1.bias,
            a.q_proj.weight, a.q_proj.bias, a.k_proj.weight, a.k_proj.bias, a.v_proj.weight, a.v_proj.bias,
            a.out_proj.weight, a.out_proj.bias, self.layer_norm2.weight, self.layer_norm2.bias,
            m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, self.heads, self.eps)


class CLIPEncoder(nn.Module):
    def __init__(self, cfg: CLIPVisionConfig, device=None):
        super().__init__()
        self.layers = nn.ModuleList([CLIPEncoderLayer(cfg, device) for _ in range(cfg.num_hidden_layers)])


class CLIPVisionTransformer(nn.Module):
    def __init__(self, cfg: CLIPVisionConfig, device=None):
        super().__init__()
        self.config = cfg
        self.embeddings = CLIPVisionEmbeddings(cfg, device)
        self.pre_layrnorm = LayerNorm(cfg.hidden_size, cfg.layer_norm_eps, device=device)   # (sic) HF's spelling
        self.encoder = CLIPEncoder(cfg, device)
        self.post_layernorm = LayerNorm(cfg.hidden_size, cfg.layer_norm_eps, device=device)

    def forward(self, pixel_values, output_hidden_states: bool = False):
        x = self.embeddings(pixel_values)
        x = self.pre_layrnorm(x)
        hs = [x] if output_hidden_states else None
        for layer in self.encoder.layers:
            x = layer(x)
            if output_hidden_states:
                hs.append(x)
        pooled = self.post_layernorm(x[:, 0, :].contiguous()) if not torch.is_grad_enabled() else None
        return VisionOutput(last_hidden_state=x, pooler_output=pooled,
                            hidden_states=tuple(hs) if output_hidden_states else None)


class CLIPVisionModel(nn.Module):
    """Same surface as transformers.CLIPVisionModel (4.45 layout)."""

    def __init__(self, config: CLIPVisionConfig, device=None):
        super().__init__()
        self.config = config
        self.vision_model = CLIPVisionTransformer(config, device)

    def forward(self, pixel_values=None, output_hidden_states: bool = False, **kwargs):
        return self.vision_model(pixel_values, output_hidden_states=output_hidden_states)

    def get_input_embeddings(self):
        return self.vision_model.embeddings.patch_embedding
