"""Checkpoint files in the reference's format (ref:utils.py:301-520): which tensors are saved, under
which names, in which file; loading tolerates wrapper prefixes, refuses unexpected keys, can skip the
adapter.  CPU only: the modules are built, never run."""
import os
import types

import pytest
import torch

from oracle import zero_oracle as zo
from tests.golden.configs import GOLDEN_CONFIGS
from tests.helpers import build_model, make_cfg
from zero_amd.configs.zero.train_llama import func_optim_filter
from zero_amd.utils import _get_raw_param_name, load_checkpoint, save_checkpoint


def _model(stage="finetune-encoder", name="tiny"):
    spec = GOLDEN_CONFIGS[name]
    enc, dec = spec["enc"], spec["dec"]
    params = zo.make_params(enc, dec, spec["seed"], surrogate=True)
    cfg = make_cfg(enc, dec, "cpu", spec["ft_after"], spec["max_seq_len"])
    cfg.training_stage = stage
    model = build_model(cfg, params)
    func_optim_filter(cfg, model)
    return cfg, model, params


def test_raw_param_name():
    assert _get_raw_param_name("module._fsdp_wrapped_module.decoder._checkpoint_wrapped_module.x._orig_mod.w") == "decoder.x.w"


@pytest.mark.parametrize("stage,name", [("finetune-encoder", "tiny"), ("pretrain", "pretrain")])
def test_save_load_roundtrip(tmp_path, stage, name):
    cfg, model, _ = _model(stage, name)
    args = types.SimpleNamespace(ckpt_dir=str(tmp_path), exp_code="exp0")
    path = save_checkpoint(args, model, None, None, epoch=1, global_step=4880)
    assert path.endswith(os.path.join("exp0", "ckpt_01_0004880.pth"))         # ref:utils.py:322-325
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    assert set(ckpt) == {"epoch", "global_step", "model", "optimizer", "scheduler"}
    trainable = {n for n, p in model.named_parameters() if p.requires_grad}
    assert set(ckpt["model"]) == trainable and "adapter.0.weight" in trainable and "embd_soi" in trainable
    if stage == "pretrain":
        assert any(k.startswith("decoder.model.layers.1.") for k in ckpt["model"])
    else:
        assert any(k.startswith("encoder.vision_model.encoder.layers.") for k in ckpt["model"])
        assert not any(k.startswith("decoder.") for k in ckpt["model"])
    saved = {k: v.clone() for k, v in ckpt["model"].items()}
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.requires_grad:
                p.add_(1.0)
    epoch, step = load_checkpoint(path, model, strict=False)
    assert (epoch, step) == (1, 4880)
    for n, p in model.named_parameters():
        if n in saved:
            assert torch.equal(p.detach(), saved[n]), n


def test_load_reference_style_keys_and_guards(tmp_path):
    cfg, model, _ = _model()
    sd = {"module." + n: p.detach().clone() + 2.0 for n, p in model.named_parameters()
          if n.startswith("adapter.") or n == "embd_soi"}
    wrapped = {"module.encoder._fsdp_wrapped_module.vision_model.post_layernorm.weight":
               model.encoder.vision_model.post_layernorm.weight.detach().clone() + 3.0}
    path = str(tmp_path / "ref_style.pth")
    torch.save({"model": {**sd, **wrapped}, "epoch": 2, "global_step": 7}, path)
    before = model.adapter[0].weight.detach().clone()
    assert load_checkpoint(path, model, strict=False, ignore_adapter_keys=True) == (2, 7)
    assert torch.equal(model.adapter[0].weight.detach(), before)                      # skipped
    assert torch.equal(model.encoder.vision_model.post_layernorm.weight.detach(), wrapped[next(iter(wrapped))])
    load_checkpoint(path, model, strict=False)
    assert torch.equal(model.adapter[0].weight.detach(), before + 2.0)
    bad = str(tmp_path / "bad.pth")
    torch.save({"model": {"adapter.9.weight": torch.zeros(2)}}, bad)
    with pytest.raises(ValueError):
        load_checkpoint(bad, model, strict=False)
    assert load_checkpoint(bad, model, strict=False, ignore_nonzero_unexpected_keys=True) == (1, 1)
    assert load_checkpoint("/nonexistent.pth", model, dry_load_mode=True) == (0, 0)
    with pytest.raises(AssertionError):
        load_checkpoint("/nonexistent.pth", model)


def test_trainer_state_dict_is_torch_adamw_layout(tmp_path):
    """The reference saves ``optimizer.state_dict()`` of torch.optim.AdamW and ``scheduler.state_dict()`` of
    CosineAnnealingLR (ref:train.py:382-414, ref:utils.py:357-377).  The Trainer's state dicts must load into those
    torch objects, and theirs into the Trainer, through a real checkpoint file (weights_only=True)."""
    from zero_amd.train import Trainer
    cfg, model, _ = _model()
    groups = func_optim_filter(cfg, model)
    tr = Trainer(cfg, model, lambda self, c: None, groups, total_steps=40)
    try:
        # the reference's side: torch AdamW over the same groups, two steps with synthetic gradients
        ref_groups = func_optim_filter(cfg, model)       # same parameters, same order
        opt = torch.optim.AdamW(ref_groups, betas=(cfg.adamw_beta1, cfg.adamw_beta2), eps=1e-8)
        sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=40, eta_min=0.0)
        g = torch.Generator().manual_seed(5)
        for _ in range(2):
            for grp in ref_groups:
                for p in grp["params"]:
                    p.grad = (0.01 * torch.randn(p.shape, generator=g)).to(p.dtype)
            opt.step(); sch.step()
        for grp in ref_groups:
            for p in grp["params"]:
                p.grad = None
        args = types.SimpleNamespace(ckpt_dir=str(tmp_path), exp_code="interop")
        path = save_checkpoint(args, model, opt, sch, epoch=0, global_step=2)        # a reference-style file
        # ... loads into the Trainer
        assert load_checkpoint(path, model, optimizer=tr, scheduler=tr.scheduler, strict=False) == (0, 2)
        assert tr.global_step == 2
        flat = [p for grp in ref_groups for p in grp["params"]]
        for i, p in enumerate(flat):
            o, n = tr.reducer.offset[id(p)], p.numel()
            assert torch.equal(tr.m[o:o + n], opt.state[p]["exp_avg"].reshape(-1).float()), i
            assert torch.equal(tr.v[o:o + n], opt.state[p]["exp_avg_sq"].reshape(-1).float()), i
        # the Trainer's file loads into fresh torch objects
        path2 = save_checkpoint(args, model, tr, tr.scheduler, epoch=0, global_step=tr.global_step + 1000)
        opt2 = torch.optim.AdamW(func_optim_filter(cfg, model), betas=(cfg.adamw_beta1, cfg.adamw_beta2), eps=1e-8)
        sch2 = torch.optim.lr_scheduler.CosineAnnealingLR(opt2, T_max=40, eta_min=0.0)
        ck = torch.load(path2, map_location="cpu", weights_only=True)
        assert set(ck["optimizer"]) == {"state", "param_groups"}
        assert [g_["params"] for g_ in ck["optimizer"]["param_groups"]] == [g_["params"] for g_ in opt.state_dict()["param_groups"]]
        opt2.load_state_dict(ck["optimizer"])
        sch2.load_state_dict(ck["scheduler"])
        assert sch2.last_epoch == 2
        for p in flat:
            assert torch.equal(opt2.state[p]["exp_avg"].float(), opt.state[p]["exp_avg"].float())
            assert float(opt2.state[p]["step"]) == 2.0
        # the cosine factor both sides apply at step 2 is the same
        assert abs(sch.get_last_lr()[0] / sch.base_lrs[0] - tr.lr_factor(2)) < 1e-6
        # a structurally different optimizer state is rejected loudly and the schedule still resumes at the file's step
        bad = {"state": {}, "param_groups": [{"params": [0, 1]}]}
        with pytest.raises(ValueError):
            tr.load_state_dict(bad)
        torch.save({"epoch": 3, "global_step": 77, "model": {}, "optimizer": bad, "scheduler": None},
                   os.path.join(str(tmp_path), "bad.pth"))
        assert load_checkpoint(os.path.join(str(tmp_path), "bad.pth"), model, optimizer=tr, strict=False) == (3, 77)
        assert tr.global_step == 77
    finally:
        tr.reducer.close()


def test_saved_lr_matches_cosine_annealing_with_min_lr():
    """``cfg.min_lr > 0``: the rate the Trainer applies, the ``lr`` it saves in ``param_groups`` and the scheduler
    view's ``_last_lr`` must all equal torch's CosineAnnealingLR(eta_min=min_lr) after N steps — torch's scheduler is
    recursive on ``group['lr']``, so a checkpoint with ``lr = base * factor`` would resume from a wrong rate."""
    from zero_amd.train import Trainer
    cfg, model, _ = _model()
    cfg.min_lr = 1e-6
    groups = func_optim_filter(cfg, model)
    T, N = 40, 7
    tr = Trainer(cfg, model, lambda self, c: None, groups, total_steps=T)
    try:
        ref_groups = func_optim_filter(cfg, model)
        opt = torch.optim.AdamW(ref_groups, betas=(cfg.adamw_beta1, cfg.adamw_beta2), eps=1e-8)
        sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=T, eta_min=cfg.min_lr)
        for _ in range(N):
            opt.step(); sch.step()
        tr.global_step = N
        want = sch.get_last_lr()
        sd, ssd = tr.state_dict(), tr.scheduler.state_dict()
        for i, g in enumerate(tr._groups):
            assert abs(tr.current_lr(g["lr"], N) - want[i]) <= 1e-9 * want[i]
            assert abs(sd["param_groups"][i]["lr"] - want[i]) <= 1e-9 * want[i]
            assert abs(ssd["_last_lr"][i] - want[i]) <= 1e-9 * want[i]
        assert ssd["eta_min"] == cfg.min_lr
        # loaded into fresh torch objects, the next scheduler step continues exactly as the uninterrupted run does
        opt2 = torch.optim.AdamW(func_optim_filter(cfg, model), betas=(cfg.adamw_beta1, cfg.adamw_beta2), eps=1e-8)
        sch2 = torch.optim.lr_scheduler.CosineAnnealingLR(opt2, T_max=T, eta_min=cfg.min_lr)
        opt2.load_state_dict(sd)
        sch2.load_state_dict(ssd)
        opt.step(); sch.step(); opt2.step(); sch2.step()
        for a, b in zip(sch.get_last_lr(), sch2.get_last_lr()):
            assert abs(a - b) <= 1e-9 * a
    finally:
        tr.reducer.close()
