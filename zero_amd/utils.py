"""Checkpoint files in the reference's format (ref:utils.py:301-520), so checkpoints move both ways
between facebookresearch/zero and zero_amd (SURVEY.md §5 "key-name contract", §8f row 3).

File:   <ckpt_dir>/<exp_code>/ckpt_{epoch:02}_{global_step:07}.pth          (ref:utils.py:320-325)
Layout: {"epoch", "global_step", "model", "optimizer", "scheduler"}           (ref:utils.py:368-377)
Model:  only the tensors whose parameter has requires_grad (ref:utils.py:344-355), under raw names —
        wrapper prefixes of DDP / FSDP / activation checkpointing / torch.compile removed
        (ref:utils.py:301-306).  zero_amd modules keep the transformers-4.45 names the released
        checkpoints use (``encoder.vision_model.encoder.layers.N…``, ``adapter.0.weight``, ``embd_soi``,
        ``decoder.model.layers.N…``), so nothing is renamed here.
Loading executes nothing from the file (``weights_only=True``, ref:utils.py:404).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

_WRAPPER_PREFIXES = ("_fsdp_wrapped_module.", "_checkpoint_wrapped_module.", "_orig_mod.", "module.")


def _get_raw_param_name(n: str) -> str:
    for pre in _WRAPPER_PREFIXES:
        n = n.replace(pre, "")
    return n


def checkpoint_path(args, epoch: int, global_step: int) -> str:
    return os.path.join(args.ckpt_dir, args.exp_code, f"ckpt_{str(epoch).zfill(2)}_{str(global_step).zfill(7)}.pth")


def save_checkpoint(args, model, optimizer, scheduler, epoch, global_step, only_save_trainable_params=True,
                    is_master=True) -> Optional[str]:
    """Rank-0 save.  ``optimizer`` / ``scheduler``: anything with ``state_dict()``, or None.  A Trainer passes
    itself and ``trainer.scheduler``: their state dicts have torch.optim.AdamW's / CosineAnnealingLR's layouts, so
    the file loads into the reference's optimizer and the reference's files load here."""
    if not is_master:
        return None
    path = checkpoint_path(args, epoch, global_step)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    raw_model = model.module if hasattr(model, "module") else model
    full = {_get_raw_param_name(k): v for k, v in raw_model.state_dict().items()}
    if only_save_trainable_params:
        state = {}
        for n, p in raw_model.named_parameters():
            if p.requires_grad:
                n = _get_raw_param_name(n)
                state[n] = full[n]
    else:
        state = full
    # detached CPU copies: parameters of a Trainer are views into one flat buffer, and a view would
    # drag the whole buffer into the file
    state = {k: v.detach().to("cpu", copy=True).contiguous() for k, v in state.items()}
    torch.save({"epoch": epoch, "global_step": global_step, "model": state,
                "optimizer": optimizer.state_dict() if optimizer is not None else None,
                "scheduler": scheduler.state_dict() if scheduler is not None else None}, path)
    return path


def load_checkpoint(path, model, optimizer=None, scheduler=None, strict=True, verbose=False,
                    ignore_nonzero_unexpected_keys=False, ignore_adapter_keys=False,
                    force_to_use_raw_param_name=False, dry_load_mode=False) -> Tuple[int, int]:
    """Returns (epoch, global_step).  Keys saved from a wrapped reference model (``module.``, FSDP,
    ``_orig_mod.``) are mapped onto this model's raw names; unexpected keys raise ValueError unless
    told otherwise (ref:utils.py:500-502); ``ignore_adapter_keys`` skips adapter / soi / eoi tensors
    (ref:utils.py:482-488).  ``force_to_use_raw_param_name`` is accepted for signature parity (the
    modules here are never wrapped, raw names are always used)."""
    if dry_load_mode:
        return 0, 0
    if not os.path.exists(path):
        raise AssertionError(f"ckpt not found: {path}")
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    states = ckpt["model"] if "model" in ckpt else ckpt
    target = model.module if hasattr(model, "module") else model
    wanted = {}
    for n, v in states.items():
        n = _get_raw_param_name(n)
        if ignore_adapter_keys and ("adapter" in n or "embd_soi" in n or "embd_eoi" in n):
            if verbose:
                print(f". [ignore] adapter keys: {n}: {tuple(v.shape)}")
            continue
        wanted[n] = v
        if verbose:
            print(f"- {n}: {tuple(v.shape)}")
    msgs = target.load_state_dict(wanted, strict=strict)
    if len(msgs.unexpected_keys) > 0 and not ignore_nonzero_unexpected_keys:
        raise ValueError(f"unexpected keys found in the model state dict: {path}: {msgs.unexpected_keys}")
    epoch = int(ckpt["epoch"]) if "epoch" in ckpt else 1
    global_step = ckpt["global_step"] if "global_step" in ckpt else 1
    for obj, key in ((optimizer, "optimizer"), (scheduler, "scheduler")):
        if obj is not None and ckpt.get(key) is not None:
            try:
                obj.load_state_dict(ckpt[key])
            except Exception as e:        # the reference only reports this (ref:utils.py:504-514)
                print(f"failed to load {key} state dict: {e}")
                if hasattr(obj, "global_step"):
                    # a Trainer whose optimizer state was rejected still resumes the schedule at the
                    # checkpoint's step (the reference replays its scheduler to global_step, ref:train.py:474-493)
                    obj.global_step = int(global_step)
    return epoch, global_step
