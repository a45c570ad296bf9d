"""Counter-based synthetic weights that come out BIT-IDENTICAL on the CPU (where the full-size golden fixture
is generated from the real transformers modules) and on the GPU (where the parity test rebuilds them on the
device in a fraction of a second instead of drawing 2.5 G values from a CPU generator).

value(name, i) = bf16( std(name) * z ),  z = (u0 + u1 + u2 + u3 - 2 * 65535) / (65536 / sqrt(3))
with u0..u3 the four 16-bit fields of splitmix64(key(name) + i): an Irwin-Hall(4) approximation of N(0, 1)
(exact integer arithmetic, then one fp32 multiply and one round-to-nearest-even to bf16 — the same IEEE
operations on both devices).  std(name) follows oracle.zero_oracle.make_params.  Test infrastructure only."""
import math
import zlib

import torch

from oracle import zero_oracle as zo

_M1 = 0xBF58476D1CE4E5B9 - (1 << 64)     # splitmix64 multipliers as signed 64-bit values
_M2 = 0x94D049BB133111EB - (1 << 64)
_GOLD = 0x9E3779B97F4A7C15 - (1 << 64)


def _lsr(x, n):
    return (x >> n) & ((1 << (64 - n)) - 1)      # logical shift right on int64


def hash_normal(numel: int, key: int, device) -> torch.Tensor:
    """fp32 [numel], approximately N(0, 1), a pure function of (key, index)."""
    out = torch.empty(numel, dtype=torch.float32, device=device)
    step = 1 << 26
    for s in range(0, numel, step):
        n = min(step, numel - s)
        x = torch.arange(s, s + n, dtype=torch.int64, device=device)
        x = (x + key) * _GOLD
        x = (x ^ _lsr(x, 30)) * _M1
        x = (x ^ _lsr(x, 27)) * _M2
        x = x ^ _lsr(x, 31)
        u = (x & 0xFFFF) + (_lsr(x, 16) & 0xFFFF) + (_lsr(x, 32) & 0xFFFF) + (_lsr(x, 48) & 0xFFFF)
        out[s:s + n] = (u - 2 * 65535).to(torch.float32) * (math.sqrt(3.0) / 65536.0)
    return out


def _key(seed: int, name: str) -> int:
    return ((seed & 0xFFFF) << 40) + (zlib.crc32(name.encode()) << 8)


def surrogate_shapes(enc: zo.EncoderCfg, dec: zo.DecoderCfg):
    """name -> shape of the SURROGATE model (post-surgery layer numbering), oracle naming."""
    shapes = {}
    shapes.update(zo.encoder_param_shapes(enc))
    shapes.update(zo.decoder_param_shapes(dec, n_layers=dec.surrogate_layers))
    shapes.update(zo.adapter_param_shapes(dec.hidden, enc.hidden))
    return shapes


_RESIDUAL_OUT = ("self_attn.o_proj.weight", "mlp.down_proj.weight", "self_attn.out_proj.weight", "mlp.fc2.weight")


def hash_params(enc: zo.EncoderCfg, dec: zo.DecoderCfg, seed: int, device, dtype=torch.bfloat16, only=None,
                init: str = "plain"):
    """Surrogate-model parameters (names as oracle.make_params(..., surrogate=True) gives them).
    init "conditioned": the output projection of every residual branch (o_proj / down_proj, out_proj / fc2) is
    drawn 1 / sqrt(2 L) smaller (L = depth of its stack: the GPT-2 / Megatron scaled init), so the residual
    stream keeps unit scale through the depth, the logits stay narrow and the loss starts near ln V — the regime a
    trained model lives in, where bf16 arithmetic holds the 1e-2 class of tolerances."""
    if init not in ("plain", "conditioned"):
        raise ValueError(init)
    out = {}
    for name, shp in sorted(surrogate_shapes(enc, dec).items()):
        if only is not None and not only(name):
            continue
        numel = 1
        for s_ in shp:
            numel *= s_
        z = hash_normal(numel, _key(seed, name), device)
        # every value below is ONE fp32 multiply of z (no a * z + b expression: a GPU fuses that into an FMA with a
        # single rounding, the CPU rounds twice, and a handful of values would land on different bf16 neighbours)
        if name in ("embd_soi", "embd_eoi"):
            t = z * (1.0 / math.sqrt(shp[1]))
        elif "norm" in name and name.endswith(".weight"):
            t = 1.0 + (z * 0.1).to(torch.bfloat16).to(torch.float32)       # exact fp32 add of a bf16 value to 1
        elif name.endswith(".bias") or name.endswith("class_embedding"):
            t = z * 0.02
        elif name.endswith("embed_tokens.weight") or name.endswith("position_embedding.weight"):
            t = z * 0.02
        else:
            fan_in = 1
            for s_ in shp[1:]:
                fan_in *= s_
            std = max(0.02, 0.5 / math.sqrt(fan_in))
            if init == "conditioned" and name.endswith(_RESIDUAL_OUT):
                depth = enc.layers if name.startswith("encoder.") else dec.surrogate_layers
                std = std / math.sqrt(2.0 * depth)
            t = z * std                          # (std is a Python float: still ONE fp32 multiply of z)
        out[name] = t.to(torch.bfloat16).to(dtype).view(shp)
    return out


def checksum(params) -> torch.Tensor:
    """Order-independent integer checksum of the bf16 bit patterns (int64 [2]: xor-free sums that cannot
    round): catches any CPU / GPU divergence of the generator."""
    tot = torch.zeros(2, dtype=torch.int64)
    for name in sorted(params):
        bits = params[name].to(torch.bfloat16).contiguous().view(torch.int16).to(torch.int64).reshape(-1)
        w = (zlib.crc32(name.encode()) & 0xFFFF) + 1
        tot[0] += int(bits.sum().item())
        tot[1] += int(((bits * w) & 0xFFFFFFFF).sum().item())
    return tot
