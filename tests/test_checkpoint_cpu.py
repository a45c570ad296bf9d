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
