"""Small nn.Module building blocks whose forward runs on the HIP kernels.  They keep the
parameter names of torch.nn.Linear / LayerNorm / LlamaRMSNorm so state dicts interoperate."""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn as nn

from .. import functional as ZF
from .. import ops


class Linear(nn.Module):
    """y = x W^T + b.  ``weight``/``bias`` may be views into a packed storage (see pack_params)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None, dtype=torch.bfloat16):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features, device=device, dtype=dtype))
        self.bias = nn.Parameter(torch.empty(out_features, device=device, dtype=dtype)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        with torch.no_grad():
            self.weight.normal_(0.0, 0.02)
            if self.bias is not None:
                self.bias.zero_()

    def forward(self, x):
        return ZF.LinearFn.apply(x, self.weight, self.bias, ops.EPI_NONE)

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}"


class LayerNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-5, device=None, dtype=torch.bfloat16):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim, device=device, dtype=dtype))
        self.bias = nn.Parameter(torch.zeros(dim, device=device, dtype=dtype))

    def forward(self, x):
        return ZF.LayerNormFn.apply(x, self.weight, self.bias, self.eps)


class RMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-5, device=None, dtype=torch.bfloat16):
        super().__init__()
        self.variance_epsilon = eps
        self.weight = nn.Parameter(torch.ones(dim, device=device, dtype=dtype))

    def forward(self, x):
        return ZF.RMSNormFn.apply(x, self.weight, self.variance_epsilon)


class Embedding(nn.Module):
    """Row gather (frozen table): ids (int64, any shape) -> [..., dim]."""

    def __init__(self, num: int, dim: int, device=None, dtype=torch.bfloat16):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num, dim
        self.weight = nn.Parameter(torch.empty(num, dim, device=device, dtype=dtype))
        with torch.no_grad():
            self.weight.normal_(0.0, 0.02)

    def forward(self, ids):
        if self.weight.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("embedding table gradient: the decoder is frozen on this path")
        codes = ids.reshape(-1).to(torch.int32).contiguous()
        out = ops.gather_rows(self.weight.detach(), None, codes, self.embedding_dim)
        return out.view(*ids.shape, self.embedding_dim)


def pack_params(params: Sequence[nn.Parameter]):
    """Re-home the given parameters as back-to-back views of ONE storage so the kernels can
    treat q|k|v (gate|up) as a single packed weight.  Values are preserved."""
    total = sum(p.numel() for p in params)
    flat = torch.empty(total, device=params[0].device, dtype=params[0].dtype)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            off += n
    return flat
