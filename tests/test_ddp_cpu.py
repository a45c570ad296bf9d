"""N > 1 path on CPU: world_size-2 gloo process group, 127.0.0.1 rendezvous.
Covers reduce_sum / reduce_sum_pair (forward SUM, identity backward: SURVEY.md Q2) and the flat
bucketed gradient reducer (bucket launch on last slot written, no_sync accumulation, zero fill of
unwritten slots).  The wgrad kernels themselves need a GPU, so the slots are filled by hand the way
functional.GradSink does it."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from zero_amd.distributed.ddp import FlatGradReducer
        from zero_amd.distributed.torchshard import reduce_sum, reduce_sum_pair
        from zero_amd.functional import GradSink
        from zero_amd.models.layers import Linear, pack_params

        # ---- scalar reductions (engine:308-313)
        x = torch.tensor(float(rank + 1), requires_grad=True)
        loss_sum = x * 3.0
        tot = reduce_sum(loss_sum)
        ntok = reduce_sum(torch.tensor(10 * (rank + 1)))
        (tot / ntok).backward()
        assert tot.item() == 3.0 * (1 + 2) and ntok.item() == 30
        assert abs(x.grad.item() - 3.0 / 30) < 1e-7          # identity backward: local sum / GLOBAL count
        y = torch.tensor(float(rank + 1), requires_grad=True)
        l2, n2 = reduce_sum_pair(y * 3.0, torch.tensor(10 * (rank + 1)))
        (l2 / n2).backward()
        assert l2.item() == 9.0 and n2.item() == 30 and abs(y.grad.item() - 0.1) < 1e-7

        # ---- flat reducer
        class M(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.q, self.k, self.v = Linear(8, 8), Linear(8, 8), Linear(8, 8)
                pack_params([self.q.weight, self.k.weight, self.v.weight])
                self.o = Linear(8, 16)
                self.frozen = Linear(8, 4)                  # not trainable: DDP syncs it all the same
                self.frozen.weight.requires_grad_(False)
                self.frozen.bias.requires_grad_(False)
                self.register_buffer("table", torch.randn(5))
        torch.manual_seed(100 + rank)                       # replicas seeded DIFFERENTLY on purpose
        m = M()
        # replicas whose frozen tensors differ: "checksum" refuses them, "broadcast" (the default) repairs them
        try:
            FlatGradReducer(m, bucket_bytes=256, sync_frozen="checksum").close()
            raised = False
        except RuntimeError as e:
            raised = "frozen" in str(e)
        assert raised, "differently seeded frozen tensors went unnoticed"
        r = FlatGradReducer(m, bucket_bytes=256)
        for t in (m.frozen.weight, m.frozen.bias, m.table):
            both_f = [torch.empty_like(t.data) for _ in range(2)]
            dist.all_gather(both_f, t.data.clone())
            assert torch.equal(both_f[0], both_f[1]), "frozen tensors differ after construction"
        assert r.world == 2 and len(r.buckets) >= 2
        params = list(r.params)
        # construction broadcast rank 0's parameters (DDP's _sync_module_states, ref:train.py:367)
        mine = r.flat_param.clone()
        both = [torch.empty_like(mine) for _ in range(2)]
        dist.all_gather(both, mine)
        assert torch.equal(both[0], both[1]), "replicas differ after FlatGradReducer construction"
        torch.manual_seed(100)
        m0 = M()
        assert torch.equal(m.o.weight.detach(), m0.o.weight.detach().to(m.o.weight.dtype))

        def write(p, val, accumulate):
            if accumulate and p._zk_dirty:
                p._zk_grad += val
            else:
                p._zk_grad.fill_(val)
            GradSink.done(p)

        # step 1: every slot but one written; that one must come out as zero
        r.begin_step()
        for i, p in enumerate(params[:-1]):
            write(p, float((rank + 1) * (i + 1)), False)
        params[-1]._zk_grad.fill_(123.0)                    # stale data from an earlier step
        r.finish_backward()
        for i, p in enumerate(params[:-1]):
            assert torch.all(p._zk_grad.float() == 3.0 * (i + 1)), (i, p._zk_grad)
        assert torch.all(params[-1]._zk_grad == 0)
        assert abs(r.grad_scale - 0.5) < 1e-12

        # step 2: accumulation over two micro-batches, exchange only on the second
        r.begin_step()
        with r.no_sync():
            for p in params:
                write(p, float(rank + 1), True)
            r.finish_backward()
        for p in params:
            assert torch.all(p._zk_grad.float() == float(rank + 1))      # not exchanged yet
        for p in params:
            write(p, float(rank + 1), True)
        r.finish_backward()
        for p in params:
            assert torch.all(p._zk_grad.float() == 6.0)                   # (1+1) + (2+2)

        # step 3: the ranks produce their gradients in DIFFERENT orders and rank 1 never writes a parameter
        # of the FIRST bucket (an unused parameter): collectives must still pair up bucket by
        # bucket (issued in index order on every rank), the skipped slot counts as zero
        r.begin_step()
        skipped = r.buckets[0]["params"][0]
        order = list(params) if rank == 0 else list(reversed(params))
        for p in order:
            if rank == 1 and p is skipped:
                continue
            write(p, float(rank + 1), False)
        if rank == 1:
            assert not any(r._launched), "a later bucket was launched ahead of the unfinished first one"
        r.finish_backward()
        for p in params:
            want = 1.0 if p is skipped else 3.0
            assert torch.all(p._zk_grad.float() == want), (r.names[id(p)], p._zk_grad)

        # a second write after the bucket went out would be lost: it must raise, not corrupt silently
        r.begin_step()
        first = r.buckets[0]["params"]
        for p in first:
            write(p, 1.0, False)
        try:
            write(first[0], 1.0, True)
            raised = False
        except RuntimeError:
            raised = True
        assert raised
        first_ids = {id(p) for p in first}
        for p in params:
            if id(p) not in first_ids:
                write(p, 1.0, False)
        r.finish_backward()
        r.close()
        FlatGradReducer(m, bucket_bytes=256, sync_frozen="checksum").close()   # frozen tensors identical now: passes
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_thread_comm_rendezvous_cpu():
    """The rank-thread communicator the one-GPU DDP test runs on: async all-reduce never blocks the caller, sums in
    rank order, broadcast copies rank 0; mismatched sizes (collectives issued in different orders) raise."""
    import threading
    from zero_amd.distributed.comm import ThreadGroup, use_comm
    from zero_amd.distributed.torchshard import reduce_sum_pair
    W = 3
    group = ThreadGroup(W)
    out, errs = {}, []

    def main(r):
        try:
            c = group.comm(r)
            b = torch.full((4,), float(r + 1))
            c.broadcast(b, src=0)
            t1, t2 = torch.full((5,), float(r + 1)), torch.full((2,), 10.0 * (r + 1))
            h1 = c.all_reduce_async(t1)          # returns at once even though the other ranks are not there yet
            h2 = c.all_reduce_async(t2)
            h1.wait(); h2.wait()
            with use_comm(c):
                l, n = reduce_sum_pair(torch.tensor(float(r + 1), requires_grad=True) * 2.0, torch.tensor(r + 1))
            out[r] = (b, t1, t2, l.item(), n.item())
        except BaseException as e:  # noqa: BLE001
            errs.append(repr(e))
    ths = [threading.Thread(target=main, args=(r,)) for r in range(W)]
    [t.start() for t in ths]
    [t.join(timeout=60) for t in ths]
    assert not errs, errs
    for r in range(W):
        b, t1, t2, l, n = out[r]
        assert torch.all(b == 1.0) and torch.all(t1 == 6.0) and torch.all(t2 == 60.0) and l == 12.0 and n == 6
    g2 = ThreadGroup(2)
    c0, c1 = g2.comm(0), g2.comm(1)
    c0.all_reduce_async(torch.zeros(3))
    with pytest.raises(RuntimeError):
        c1.all_reduce_async(torch.zeros(4))
