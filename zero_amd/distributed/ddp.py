"""Data-parallel gradient exchange for the trainable set (replaces torch DDP at
ref:train.py:367 and ``no_sync`` at ref:train.py:562-564).

Design (MI355X / RCCL over xGMI):
  * only the TRAINABLE parameters (~113 M for the 3B config) are exchanged — the reference
    registers every parameter with DDP before ``func_optim_filter`` runs and so all-reduces
    the frozen 2.5 B surrogate as well; the result is identical;
  * parameters and gradients live in two flat bf16 buffers laid out in gradient-ready order
    (reverse registration order); packed q|k|v groups stay adjacent.  The wgrad kernels write
    straight into the gradient slots (functional.GradSink), so there is no bucket copy;
  * buckets are contiguous slices (~25 MB, DDP's default, about one encoder layer each);
    a bucket's all-reduce (SUM; the 1/world average is folded into the optimizer kernel) is
    launched asynchronously once its last slot AND every earlier bucket have been written,
    overlapping the remaining backward.  Collectives are therefore issued in bucket-index order on
    every rank, whatever order (or subset) of gradients a rank produced — RCCL pairs collectives by
    issue order, and a rank whose batch leaves a parameter unused (the case the reference passes
    find_unused_parameters=True for) must not reorder them; buckets with unwritten
    slots are flushed, zero-filled, in index order by finish_backward();
  * construction broadcasts rank 0's parameters (what DDP(...) does at ref:train.py:367), so replicas
    start identical even if the ranks were seeded differently;
  * one gradient write per parameter and backward (the layer nodes of functional.py write each slot
    once): a second write after the bucket's all-reduce was launched would be lost, so it raises;
  * gradient accumulation: ``no_sync()`` suppresses the launches; slots accumulate in place.
"""
from __future__ import annotations

import os
from contextlib import contextmanager
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from ..functional import SinkState
from .comm import TorchComm

_DEBUG = os.environ.get("ZERO_AMD_DDP_DEBUG") == "1"
# (gloo on DEVICE tensors is not a usable rehearsal of this path on a one-GPU box: several of its
# host-staged exchanges interleaved with kernels hang the device there, tools/probe_gloo_cuda.py
# reproduces it with torch.matmul alone; RCCL passes the same probe.  gloo rehearsals stay on CPU tensors.)


def _dbg(msg: str):
    import sys, threading
    r = dist.get_rank() if dist.is_initialized() else 0
    print(f"[ddp rank {r} {threading.current_thread().name}] {msg}", file=sys.stderr, flush=True)


def _pack_groups(params: List[nn.Parameter]) -> List[List[nn.Parameter]]:
    """Keep tensors that are currently back-to-back in one storage (pack_params) together."""
    by_storage: Dict[int, List[nn.Parameter]] = {}
    for p in params:
        by_storage.setdefault(p.untyped_storage().data_ptr(), []).append(p)
    group_of: Dict[int, List[nn.Parameter]] = {}
    for ps in by_storage.values():
        ps = sorted(ps, key=lambda t: t.storage_offset())
        run = [ps[0]]
        for a, b in zip(ps, ps[1:]):
            if b.storage_offset() == a.storage_offset() + a.numel():
                run.append(b)
            else:
                for t in run:
                    group_of[id(t)] = run
                run = [b]
        for t in run:
            group_of[id(t)] = run
    out, seen = [], set()
    for p in params:
        if id(p) in seen:
            continue
        g = group_of[id(p)]
        out.append(g)
        seen.update(id(t) for t in g)
    return out


class FlatGradReducer:
    def __init__(self, model: nn.Module, bucket_bytes: int = 25 * 1024 * 1024, process_group=None, comm=None,
                 sync_frozen: str = "broadcast"):
        self.comm = comm if comm is not None else TorchComm(process_group)     # RCCL unless a test passes its own
        self.pg = process_group
        self.world = self.comm.world
        # one-GPU rehearsal of the collective path (bench.py --force-dist): issue the all-reduces
        # of a 1-rank group instead of skipping them
        self._force = dist.is_initialized() and os.environ.get("ZERO_AMD_FORCE_ALLREDUCE") == "1"
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("no trainable parameters: call func_optim_filter before building the reducer")
        self.names = {id(p): n for n, p in named}
        # this replica's gradient-sink state (functional.GradSink finds it through the parameters)
        self.sink = SinkState(on_grad=self._on_grad)
        ordered = [p for _, p in reversed(named)]                 # gradient-ready order
        groups = _pack_groups(ordered)
        self.params: List[nn.Parameter] = [p for g in groups for p in g]
        dev = self.params[0].device
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 7) // 8 * 8                     # 16-byte aligned slots
        self.numel = total
        self.flat_param = torch.zeros(total, dtype=torch.bfloat16, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.bfloat16, device=dev)
        self.offset: Dict[int, int] = {}
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                n = p.numel()
                self.flat_param[o:o + n].copy_(p.reshape(-1))
                p.data = self.flat_param[o:o + n].view(p.shape)
                p._zk_grad = self.flat_grad[o:o + n]
                p._zk_dirty = False
                p._zk_sink = self.sink
                self.offset[id(p)] = o
        if self.world > 1 or self._force:            # (--force-dist: the one-rank rehearsal issues these collectives too)
            # DDP's _sync_module_states (ref:train.py:367): every replica starts from rank 0's parameters AND
            # buffers — the trainable ones in one flat broadcast, the frozen ones (the surrogate decoder, encoder
            # layers 0-15: replicas seeded differently or loaded from different partial checkpoints would train on
            # different functions without any error) tensor by tensor, or verified by a checksum
            self.comm.broadcast(self.flat_param, src=0)
            self._sync_frozen(model, sync_frozen)
        # contiguous buckets
        self.buckets: List[dict] = []
        cur = dict(start=0, end=0, params=[])
        for p, o in zip(self.params, offs):
            end = o + (p.numel() + 7) // 8 * 8
            cur["params"].append(p)
            cur["end"] = end
            if (cur["end"] - cur["start"]) * 2 >= bucket_bytes:
                self.buckets.append(cur)
                cur = dict(start=end, end=end, params=[])
        if cur["params"]:
            self.buckets.append(cur)
        self.bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b["params"]}
        self._pending = [0] * len(self.buckets)
        self._handles: List = []
        self._sync = True
        self._launched = [False] * len(self.buckets)
        self.begin_step()

    def _sync_frozen(self, model: nn.Module, mode: str):
        """mode "broadcast": rank 0's frozen parameters and buffers overwrite everyone's (what DDP does);
        "checksum": nothing is moved, but replicas whose frozen tensors differ raise (4.4-10 GB less traffic at
        start-up when every rank loaded the same checkpoint); "off": trust the caller."""
        if mode not in ("broadcast", "checksum", "off"):
            raise ValueError(f"sync_frozen = {mode!r}")
        if mode == "off":
            return
        mine = {id(p) for p in self.params}
        seen, frozen = set(), []
        for n, t in list(model.named_parameters()) + list(model.named_buffers()):
            key = (t.untyped_storage().data_ptr(), t.storage_offset(), t.numel())
            if id(t) in mine or t.numel() == 0 or key in seen:       # tied weights: once
                continue
            if t.device != self.flat_param.device:                   # a host-side table: not this backend's to move
                continue
            seen.add(key)
            frozen.append((n, t))
        if mode == "broadcast":
            with torch.no_grad():
                for n, t in frozen:
                    d = t.data
                    if d.is_contiguous():
                        self.comm.broadcast(d, src=0)
                    else:
                        c = d.contiguous()
                        self.comm.broadcast(c, src=0)
                        d.copy_(c)
                    # layouts derived from a frozen weight (functional._frozen_t / _frozen_gate_up) follow the
                    # weight: drop whatever was built before the broadcast
                    for k in ("_zk_wt", "_zk_gu"):
                        t.__dict__.pop(k, None)
            return
        # checksum: per tensor an order-independent fp64 sum of the raw bit patterns read as integers (exact for
        # any size this path has: 2^16 * 2^33 elements < 2^53), all in ONE all-reduce of [sum_r0-weighted, sum]
        sums = []
        for n, t in frozen:
            d = t.detach()
            if d.dtype in (torch.bfloat16, torch.float16):
                bits = d.contiguous().view(torch.int16).to(torch.float64)
            elif d.dtype == torch.float32:
                bits = d.contiguous().view(torch.int32).to(torch.float64)
            else:
                bits = d.to(torch.float64)
            sums.append(bits.sum())
        local = torch.stack(sums) if sums else torch.zeros(0, dtype=torch.float64, device=self.flat_param.device)
        ref = local.clone()
        self.comm.broadcast(ref, src=0)
        bad = (ref != local).to(torch.float64)
        self.comm.all_reduce(bad)
        if bool(bad.sum() > 0):
            names = [frozen[i][0] for i in torch.nonzero(bad.cpu()).flatten().tolist()[:8]]
            raise RuntimeError(f"frozen tensors differ between replicas (first: {names}): load the same checkpoint on "
                               "every rank or construct the reducer with sync_frozen='broadcast'")

    # ---- step protocol ---------------------------------------------------------------
    def begin_step(self):
        """zero_grad(set_to_none=True) analogue: slots are overwritten by the first write."""
        for p in self.params:
            p._zk_dirty = False
        self._begin_backward()

    def _begin_backward(self):
        self._pending = [len(b["params"]) for b in self.buckets]
        self._seen = set()
        self._launched = [False] * len(self.buckets)
        self._next = 0                      # first bucket whose all-reduce has not been issued
        self._handles = []

    @contextmanager
    def no_sync(self):
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def _on_grad(self, p):
        i = self.bucket_of.get(id(p))
        if i is None:
            return
        if id(p) in self._seen:
            if self._launched[i] and self._sync:
                raise RuntimeError(f"gradient of {self.names[id(p)]} written again after its bucket's all-reduce "
                                   "was launched (one write per parameter and backward is assumed)")
            return
        self._seen.add(id(p))
        self._pending[i] -= 1
        if self._sync:
            # issue only the contiguous ready prefix: same collective order on every rank
            while self._next < len(self.buckets) and self._pending[self._next] == 0:
                self._launch(self._next)
                self._next += 1

    def _launch(self, i):
        if self._launched[i] or (self.world == 1 and not self._force):
            self._launched[i] = True
            return
        b = self.buckets[i]
        if _DEBUG:
            _dbg(f"launch bucket {i} [{b['start']}:{b['end']}) of {len(self.buckets)}")
        self._handles.append(self.comm.all_reduce_async(self.flat_grad[b["start"]:b["end"]]))
        self._launched[i] = True

    def finish_backward(self):
        """After loss.backward(): zero the slots nothing wrote to, flush the remaining buckets,
        wait for the exchanges (stream-ordered), and get ready for the next backward."""
        for p in self.params:
            if not p._zk_dirty:
                p._zk_grad.zero_()
                p._zk_dirty = True
        if self._sync:
            for i in range(self._next, len(self.buckets)):     # the rest, in index order
                self._launch(i)
            self._next = len(self.buckets)
            if _DEBUG:
                _dbg("device sync before the waits ...")
                torch.cuda.synchronize()
                _dbg("... device idle")
            for k, h in enumerate(self._handles):
                if _DEBUG:
                    _dbg(f"wait handle {k} of {len(self._handles)}")
                h.wait()
        self._begin_backward()

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world                                   # DDP averages

    def close(self):
        self.sink.on_grad = None
        for p in self.params:
            if hasattr(p, "_zk_grad"):
                del p._zk_grad
            if getattr(p, "_zk_sink", None) is self.sink:
                del p._zk_sink
