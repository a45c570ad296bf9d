"""The collectives the data-parallel step needs, behind one small interface so that the SAME Trainer /
FlatGradReducer / reduce_sum_pair code runs over

  * ``TorchComm``  — torch.distributed (backend "nccl" = RCCL over xGMI on MI355X): the product path, one
    process per GPU (ref:utils.py:253-257, ref:train.py:367);
  * ``ThreadComm`` — W rank THREADS of one process on one device (test harness for a one-GPU box: RCCL
    refuses two ranks on one device).  Collectives are rendezvous points; the reduction itself is enqueued on
    the (shared) current stream by whichever rank arrives last, so it is stream-ordered behind every rank's
    producer kernels exactly like an RCCL kernel would be.  ``all_reduce_async`` never blocks the caller:
    gradient buckets are launched from autograd's device thread, which serves every replica of the process —
    a blocking rendezvous there would deadlock.

Only three operations exist on this path: the bucketed gradient all-reduce (async SUM), the construction-time
parameter broadcast, and the blocking 2-element all-reduce of (loss_sum, n_tokens).
"""
from __future__ import annotations

import threading
from typing import List, Optional

import torch
import torch.distributed as dist


class _Done:
    def wait(self):
        return True


class TorchComm:
    def __init__(self, process_group=None):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0

    def all_reduce_async(self, t: torch.Tensor):
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        return t

    def broadcast(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.world > 1:
            gsrc = dist.get_global_rank(self.pg, src) if self.pg is not None else src
            dist.broadcast(t, src=gsrc, group=self.pg)
        return t


class _Rendezvous:
    """One collective call site: the k-th call of every rank meets here."""

    def __init__(self, world: int):
        self.slots: List[Optional[torch.Tensor]] = [None] * world
        self.arrived = 0
        self.done = threading.Event()


class ThreadGroup:
    """Shared state of W rank threads; ``comm(rank)`` gives each thread its endpoint."""

    def __init__(self, world: int):
        self.world = world
        self._lock = threading.Lock()
        self._calls = {}           # (kind, sequence number) -> _Rendezvous
        self.log: List[tuple] = []  # (rank, kind, numel) in arrival order, for tests of the collective ORDER

    def comm(self, rank: int) -> "ThreadComm":
        return ThreadComm(self, rank)

    def _meet(self, kind: str, seq: int, rank: int, t: torch.Tensor, finish):
        with self._lock:
            key = (kind, seq)
            rv = self._calls.get(key)
            if rv is None:
                rv = self._calls[key] = _Rendezvous(self.world)
            if rv.slots[rank] is not None:
                raise RuntimeError(f"rank {rank} entered {kind} #{seq} twice")
            rv.slots[rank] = t
            rv.arrived += 1
            self.log.append((rank, kind, t.numel()))
            last = rv.arrived == self.world
            if last:
                if len({s.numel() for s in rv.slots}) != 1:
                    raise RuntimeError(f"{kind} #{seq}: ranks passed different sizes "
                                       f"{[s.numel() for s in rv.slots]} (collectives issued in different orders)")
                finish(rv.slots)          # enqueued on the current stream, behind every rank's producers
                del self._calls[key]
                rv.done.set()
        return rv


class _Handle:
    def __init__(self, rv: _Rendezvous):
        self._rv = rv

    def wait(self, timeout: float = 120.0):
        if not self._rv.done.wait(timeout):
            raise RuntimeError("collective timed out: a rank never arrived")
        return True


class ThreadComm:
    def __init__(self, group: ThreadGroup, rank: int):
        self.group, self.rank, self.world = group, rank, group.world
        self._seq = {"ar": 0, "bc": 0}

    @staticmethod
    def _sum(slots):
        # rank order, fp32 accumulation, one rounding: what a reduction tree of bf16 values does up to order
        acc = torch.float64 if slots[0].dtype == torch.float64 else torch.float32
        total = slots[0].to(acc)
        for s in slots[1:]:
            total = total + s.to(acc)
        for s in slots:
            s.copy_(total.to(s.dtype))

    def all_reduce_async(self, t: torch.Tensor):
        seq = self._seq["ar"]
        self._seq["ar"] += 1
        return _Handle(self.group._meet("ar", seq, self.rank, t, self._sum))

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        self.all_reduce_async(t).wait()
        return t

    def broadcast(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        seq = self._seq["bc"]
        self._seq["bc"] += 1

        def finish(slots):
            for i, s in enumerate(slots):
                if i != src:
                    s.copy_(slots[src])
        _Handle(self.group._meet("bc", seq, self.rank, t, finish)).wait()
        return t


# ---- the communicator reduce_sum / reduce_sum_pair use: torch.distributed unless a thread installs its own
_TLS = threading.local()


def current_comm():
    c = getattr(_TLS, "comm", None)
    return c if c is not None else TorchComm()


class use_comm:
    """``with use_comm(c):`` — route this THREAD's scalar reductions (ref:engine:308-310) through ``c``."""

    def __init__(self, comm):
        self.comm = comm

    def __enter__(self):
        self.prev = getattr(_TLS, "comm", None)
        _TLS.comm = self.comm
        return self.comm

    def __exit__(self, *exc):
        _TLS.comm = self.prev
        return False
