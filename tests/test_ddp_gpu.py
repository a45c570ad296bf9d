"""Two REAL Trainer replicas as rank threads on one device (zero_amd.distributed.comm.ThreadGroup: RCCL refuses
two ranks on one GPU, and the one-GPU box has no second device).  Everything except the transport is the product
path: forward_from_ids with its 2-element (loss_sum, n_tokens) all-reduce, the layer nodes' wgrad kernels writing
into the flat gradient slots, FlatGradReducer's construction-time broadcast and its bucketed asynchronous
all-reduces launched from autograd's device thread, the fused clip + AdamW.

Checks (ref:train.py:367,523-583):
  * replicas seeded DIFFERENTLY hold rank 0's parameters after construction;
  * 2 ranks x B = 2 give the oracle's B = 4 gradient: every rank back-propagates local_loss_sum / GLOBAL tokens,
    the exchange sums, the 1/W of DDP's average is applied by the optimizer kernel (grad_scale);
  * gradients and parameters are bit-identical on both ranks after the step;
  * bucket all-reduces go out in the same order on both ranks, and only on the LAST micro-step under no_sync."""
import threading

import pytest
import torch

from oracle import zero_oracle as zo
from tests.golden.configs import GOLDEN_CONFIGS

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run_ranks(world, fn):
    out, errs = [None] * world, []

    def wrap(r):
        try:
            out[r] = fn(r)
        except BaseException as e:  # noqa: BLE001
            import traceback
            errs.append((r, traceback.format_exc()))
    ths = [threading.Thread(target=wrap, args=(r,), name=f"rank{r}") for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=240)
    assert not any(t.is_alive() for t in ths), "a rank thread hangs (collective never completed)"
    assert not errs, errs[0][1]
    return out


@pytest.mark.parametrize("accum", [1, 2])
def test_two_trainer_replicas_match_oracle_global_batch(accum):
    from tests.helpers import build_model, make_cfg
    from zero_amd.configs.zero.train_llama import func_optim_filter
    from zero_amd.distributed.comm import ThreadGroup
    from zero_amd.methods.engine_ar_instruction_llama import forward_from_ids
    from zero_amd.train import Trainer
    spec = GOLDEN_CONFIGS["tiny"]
    enc, dec = spec["enc"], spec["dec"]
    W, Bl = 2, 2
    params = zo.make_params(enc, dec, spec["seed"], surrogate=True)
    names = zo.trainable_names(params, spec["ft_after"])
    nb = W * Bl * accum
    imgs, ids, tgts, img_id, pad_id = zo.synth_batch(enc, dec, nb, 23, n_pre=5, n_post=24, n_resp=12, ragged=True,
                                                     text_only=(1,))
    imgs_r = [i.to(torch.bfloat16).float() for i in imgs]
    # sample -> (micro-step, rank): micro-step m holds samples [m * W * Bl, (m + 1) * W * Bl), rank r its r-th half
    def mine(m, r):
        lo = m * W * Bl + r * Bl
        return list(range(lo, lo + Bl))

    # oracle: per micro-step the GLOBAL batch of W * Bl samples, mean over its tokens; micro-steps averaged (/accum)
    ref = {n: torch.zeros_like(params[n]) for n in names}
    for m in range(accum):
        sel = [i for r in range(W) for i in mine(m, r)]
        o = zo.train_step(params, enc, dec, [imgs_r[i] for i in sel], [ids[i] for i in sel], [tgts[i] for i in sel],
                          img_id=img_id, pad_id=pad_id, max_seq_len=spec["max_seq_len"],
                          encoder_ft_layers_after=spec["ft_after"])
        for n in names:
            ref[n] += o.grads[n] / accum

    group = ThreadGroup(W)
    models = []
    for r in range(W):
        cfg = make_cfg(enc, dec, DEV, spec["ft_after"], spec["max_seq_len"])
        cfg.gradient_accumulation_steps = accum
        cfg.check_nan_loss = False
        p_r = dict(params)
        if r > 0:      # a replica that was initialised differently: the construction broadcast must overwrite it
            g = torch.Generator().manual_seed(1000 + r)
            p_r = {k: (v + 0.05 * torch.randn(v.shape, generator=g) if k in names else v) for k, v in params.items()}
        model = build_model(cfg, p_r)
        groups = func_optim_filter(cfg, model)
        model.train()
        models.append((cfg, model, groups))

    def rank_main(r):
        cfg, model, groups = models[r]
        tr = Trainer(cfg, model, forward_from_ids, groups, total_steps=10, comm=group.comm(r), bucket_bytes=1 << 20)
        try:
            start = tr.reducer.flat_param.clone()
            launched = []
            for m in range(accum):
                sel = mine(m, r)
                before = len([1 for e in group.log if e[0] == r and e[1] == "ar"])
                tr.reducer_probe = None
                # stop before the optimizer on the last micro-step so the exchanged gradients can be read
                if m == accum - 1:
                    orig = tr.optimizer_step
                    grads = {}

                    def probe():
                        torch.cuda.synchronize()
                        for p in tr.reducer.params:
                            grads[tr.reducer.names[id(p)]] = p._zk_grad.detach().view(p.shape).clone()
                        orig()
                    tr.optimizer_step = probe
                loss = tr.micro_step([imgs[i].to(DEV).to(torch.bfloat16) for i in sel], [ids[i] for i in sel],
                                     [tgts[i] for i in sel], img_id, pad_id)
                launched.append(len([1 for e in group.log if e[0] == r and e[1] == "ar"]) - before)
            torch.cuda.synchronize()
            return dict(start=start, end=tr.reducer.flat_param.clone(), grads=grads, loss=float(loss),
                        launched=launched, nbuckets=len(tr.reducer.buckets), scale=tr.reducer.grad_scale)
        finally:
            tr.reducer.close()

    res = _run_ranks(W, rank_main)
    a, b = res
    # construction: both replicas start from rank 0's parameters
    assert torch.equal(a["start"], b["start"])
    # the loss every rank reports is the GLOBAL mean of the last micro-step (engine:308-313)
    assert a["loss"] == b["loss"]
    # exchanged gradients: identical on both ranks, equal to the oracle's global-batch gradient (x accum: micro-step
    # losses are divided by accum before backward, train.py:527, and so is `ref`)
    assert set(a["grads"]) == set(names)
    scale = max(v.norm().item() / v.numel() ** 0.5 for v in ref.values())
    worst = ("", 0.0)
    for n in names:
        assert torch.equal(a["grads"][n], b["grads"][n]), n
        g, r_ = a["grads"][n].float().cpu(), ref[n]
        e = ((g - r_).norm() / max(r_.norm().item(), 1e-3 * scale * r_.numel() ** 0.5)).item()
        if e > worst[1]:
            worst = (n, e)
    print("two replicas vs oracle global batch: worst gradient", worst)
    assert worst[1] <= 3e-2, worst
    assert a["scale"] == 0.5                               # DDP's average: 1 / world, applied in the optimizer kernel
    # after the step: identical parameters on both ranks, and they moved
    assert torch.equal(a["end"], b["end"]) and not torch.equal(a["start"], a["end"])
    # collectives per micro-step: 1 (the scalar pair) while accumulating, 1 + every bucket on the last one
    assert a["nbuckets"] >= 2
    want = [1] * (accum - 1) + [1 + a["nbuckets"]]
    assert a["launched"] == want and b["launched"] == want, (a["launched"], b["launched"], want)
    # same collective sizes in the same order on both ranks
    seq = {r: [e[2] for e in group.log if e[0] == r and e[1] == "ar"] for r in range(W)}
    assert seq[0] == seq[1]
