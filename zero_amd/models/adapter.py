"""``Zero``: encoder + adapter + decoder container (ref:models/adapter.py:13-116), same
constructor, attributes (state-dict roots ``encoder``, ``adapter``, ``embd_soi``,
``embd_eoi``, ``decoder``) and ``_forward`` contract; arithmetic on HIP kernels."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from .. import functional as ZF
from .. import ops
from .layers import Linear


class AdapterMLP(nn.Sequential):
    """Linear -> GELU -> Linear [-> GELU -> Linear …] with the reference's module indices
    (``adapter.0``, ``adapter.2`` …).  The exact-erf GELU runs in the GEMM epilogue."""

    def forward(self, x):
        lin = [m for m in self if isinstance(m, Linear)]
        if len(lin) == 2:
            return ZF.AdapterFn.apply(x, lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias)
        for m in lin[:-1]:
            x = ZF.LinearFn.apply(x, m.weight, m.bias, ops.EPI_GELU_ERF)
        return ZF.LinearFn.apply(x, lin[-1].weight, lin[-1].bias, ops.EPI_NONE)


class Zero(nn.Module):
    def __init__(self, cfg: object, encoder: Optional[nn.Module] = None, decoder: Optional[nn.Module] = None):
        super().__init__()
        mlp_depth = cfg.adapter_mlp_depth
        num_embd = cfg.num_embd
        device = getattr(cfg, "device", None)
        self.encoder = encoder
        self.encoder_shave_last_n_layers = cfg.encoder_shave_last_n_layers
        if self.encoder_shave_last_n_layers > 0:
            print(f"shave last {self.encoder_shave_last_n_layers} layers of encoder")
        self.ignore_encoder_cls_token = cfg.ignore_encoder_cls_token

        # the reference hard-codes the encoder width 1024 (adapter.py:34)
        modules: List[nn.Module] = [Linear(1024, num_embd, device=device)]
        for _ in range(1, mlp_depth):
            modules.append(nn.GELU())
            modules.append(Linear(num_embd, num_embd, device=device))
        self.adapter = AdapterMLP(*modules)

        if self.encoder is not None:
            std = 1 / num_embd ** 0.5                                     # adapter.py:56-60
            self.embd_soi = nn.Parameter((torch.randn(1, num_embd, device=device) * std).to(torch.bfloat16))
            self.embd_eoi = nn.Parameter((torch.randn(1, num_embd, device=device) * std).to(torch.bfloat16))
        self.decoder = decoder
        if self.encoder is None:
            self.adapter = None                                           # language-only (adapter.py:49-50)

    # the real forward is bound from methods/ (ref:train.py:213-214)
    def forward(self):
        raise NotImplementedError

    def _encode_rows(self, xs: List[torch.Tensor]) -> Tuple[torch.Tensor, List[int], int]:
        """cat -> encoder -> select hidden -> (drop CLS) -> post_layernorm(all tokens) -> adapter.
        Returns ([sum(P)*T, d] rows, patches per sample, tokens per patch)."""
        split_sizes = [x.shape[0] for x in xs]
        x = torch.cat(list(xs), dim=0) if len(xs) > 1 else xs[0]
        need_hs = self.encoder_shave_last_n_layers > 0
        out = self.encoder(x, output_hidden_states=True if need_hs or not _is_zero_encoder(self.encoder) else False)
        if self.encoder_shave_last_n_layers > 0:
            x = out.hidden_states[-self.encoder_shave_last_n_layers]
        else:
            x = out.last_hidden_state
        if self.ignore_encoder_cls_token:
            x = x[:, 1:, :].contiguous()
        x = self.encoder.vision_model.post_layernorm(x)
        x = self.adapter(x)
        return x.reshape(-1, x.shape[-1]), split_sizes, x.shape[1]

    def _forward(self, xs: List[torch.Tensor]) -> List[torch.Tensor]:
        """ref:models/adapter.py:66-116: bs * [1 + num_patches * num_tokens + 1, dim]."""
        rows, split_sizes, T = self._encode_rows(xs)
        pieces = torch.split(rows, [p * T for p in split_sizes], dim=0)
        return [torch.cat([self.embd_soi, piece, self.embd_eoi], dim=0) for piece in pieces]


def _is_zero_encoder(enc) -> bool:
    from .clip import CLIPVisionModel
    return isinstance(enc, CLIPVisionModel)
