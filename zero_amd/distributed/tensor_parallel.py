"""Tensor-parallel sharding of the Llama decoder for the zero-shot graft forward on a full LLM
(BASELINE configs[4]: surrogate-trained encoder + translator feeding Llama-3.1-70B, 8 MI355X).

The reference spreads the 80 layers over the GPUs with accelerate's ``device_map="balanced"``
(ref:build_decoder.py:46-50): a sequential layer pipeline in which one GPU works at a time.  On an
xGMI node every GPU holds 1/W of EVERY layer instead (Megatron layout): q/k/v and gate/up are split
by output rows (whole heads; 70B has 8 kv heads = one per GPU at W = 8), o and down by input columns,
so a layer needs two all-reduces (RCCL, [tokens, d] bf16) and all W GPUs compute all the time.
Embedding, final norm and lm_head stay replicated (2.1 GB for 70B).  Inference only.
"""
from __future__ import annotations

import threading
from typing import Callable, List, Optional

import torch
import torch.distributed as dist

from ..models.layers import pack_params


def shard_llama_(decoder, rank: int, world: int):
    """In place: keep this rank's slice of every decoder layer.  Returns the decoder."""
    cfg = decoder.config
    nh, nkv, hd, f = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.intermediate_size
    if nh % world or nkv % world or f % (32 * world):
        raise ValueError(f"cannot split heads {nh}/{nkv} and MLP width {f} over {world} ranks")
    q0, q1 = rank * (nh // world) * hd, (rank + 1) * (nh // world) * hd
    k0, k1 = rank * (nkv // world) * hd, (rank + 1) * (nkv // world) * hd
    f0, f1 = rank * (f // world), (rank + 1) * (f // world)
    with torch.no_grad():
        for layer in decoder.model.layers:
            a, m = layer.self_attn, layer.mlp
            a.q_proj.weight.data = a.q_proj.weight.data[q0:q1].clone()
            a.k_proj.weight.data = a.k_proj.weight.data[k0:k1].clone()
            a.v_proj.weight.data = a.v_proj.weight.data[k0:k1].clone()
            pack_params([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight])
            a.o_proj.weight.data = a.o_proj.weight.data[:, q0:q1].contiguous()
            m.gate_proj.weight.data = m.gate_proj.weight.data[f0:f1].clone()
            m.up_proj.weight.data = m.up_proj.weight.data[f0:f1].clone()
            pack_params([m.gate_proj.weight, m.up_proj.weight])
            m.down_proj.weight.data = m.down_proj.weight.data[:, f0:f1].contiguous()
            for p in (a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, a.o_proj.weight, m.gate_proj.weight,
                      m.up_proj.weight, m.down_proj.weight):
                p.__dict__.pop("_zk_wt", None)       # resident copies of the unsharded weights
                p.__dict__.pop("_zk_gu", None)
    decoder.tp_rank, decoder.tp_world = rank, world
    return decoder


class TPContext:
    """What the layer code needs to know about the group: is this the rank that adds the residual
    (exactly one must), and how partial sums are combined."""

    def __init__(self, rank: int = 0, world: int = 1, process_group=None, reduce_fn: Optional[Callable] = None):
        self.rank, self.world, self.pg = rank, world, process_group
        self._reduce = reduce_fn

    @property
    def lead(self) -> bool:
        return self.rank == 0

    def reduce(self, part: torch.Tensor) -> torch.Tensor:
        if self._reduce is not None:
            return self._reduce(part)
        if self.world > 1:
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.pg)     # RCCL over xGMI ("nccl" backend)
        return part


class ThreadSimGroup:
    """Test harness: W rank threads in ONE process on one device; ``all_reduce`` is a rendezvous that
    sums the ranks' bf16 partials in rank order (what a ring/tree reduction of bf16 values does up to
    association order).  Lets the single-GPU box run the real sharded code path."""

    def __init__(self, world: int):
        self.world = world
        self._bar = threading.Barrier(world)
        self._slots: List[Optional[torch.Tensor]] = [None] * world

    def context(self, rank: int) -> TPContext:
        def reduce(part: torch.Tensor) -> torch.Tensor:
            self._slots[rank] = part
            self._bar.wait()
            total = self._slots[0].float()
            for t in self._slots[1:]:
                total = total + t.float()
            out = total.to(part.dtype)
            self._bar.wait()                     # nobody overwrites a slot before everyone has summed
            part.copy_(out)                      # in place, like dist.all_reduce
            return part
        return TPContext(rank, self.world, reduce_fn=reduce)
