"""The DDP training step of the reference (ref:train.py:367,381-387,523-583) on HIP kernels.

``Trainer`` owns what train.py's loop owns: the bound forward, the gradient exchange
(distributed/ddp.py), grad-norm clipping and AdamW (ONE fused kernel over the flat trainable
buffer: clip coefficient from a device-side sum of squares, no host sync), the cosine
schedule with linear warm-up (train.py:410-414,495-521) and gradient accumulation with
``no_sync``.  bf16 parameters without an fp32 master copy, like the reference
(``model.to(bf16)`` then AdamW on those tensors, train.py:209,382); the Adam moments are kept in
fp32."""
from __future__ import annotations

import math
from typing import Callable, List, Optional

import torch
import torch.nn as nn

from . import ops
from .distributed.comm import use_comm
from .distributed.ddp import FlatGradReducer
from .functional import ZeroArena, bump_weights_epoch


class Trainer:
    def __init__(self, cfg, model: nn.Module, forward_fn: Callable, optim_groups: List[dict], *,
                 total_steps: int = 1000, warmup_steps: int = 0, process_group=None, bucket_bytes=25 << 20,
                 comm=None, sync_frozen: str = "broadcast"):
        self.cfg, self.model = cfg, model
        self.forward = forward_fn.__get__(model, model.__class__)          # train.py:213-214
        self.comm = comm             # None: torch.distributed (RCCL); tests pass a rank-thread communicator
        self.reducer = FlatGradReducer(model, bucket_bytes=bucket_bytes, process_group=process_group, comm=comm,
                                       sync_frozen=sync_frozen)
        r = self.reducer
        dev = r.flat_param.device
        self.m = torch.zeros(r.numel, dtype=torch.float32, device=dev)
        self.v = torch.zeros(r.numel, dtype=torch.float32, device=dev)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.arena = ZeroArena(dev)             # per-backward fp32 accumulators of the vector gradients
        r.sink.arena = self.arena               # this replica's own (functional.SinkState): nothing process-wide
        # (lr, wd) per parameter from the reference-style groups, merged into contiguous segments
        hp = {}
        for g in optim_groups:
            for p in g["params"]:
                hp[id(p)] = (float(g["lr"]), float(g.get("weight_decay", 0.0)))
        segs = []
        for p in r.params:
            o = r.offset[id(p)]
            e = o + (p.numel() + 7) // 8 * 8
            h = hp.get(id(p), (float(cfg.lr), 0.0))
            if segs and segs[-1][2] == h and segs[-1][1] == o:
                segs[-1][1] = e
            else:
                segs.append([o, e, h])
        self.segments = segs
        # torch.optim layout bookkeeping (state_dict / load_state_dict): parameter index = position in the
        # concatenation of the reference-style groups, as torch.optim.Optimizer numbers them
        self._groups = [dict(params=[p for p in g["params"] if id(p) in r.offset], lr=float(g["lr"]),
                             weight_decay=float(g.get("weight_decay", 0.0))) for g in optim_groups]
        self.total_steps, self.warmup_steps = total_steps, warmup_steps
        self.global_step = 0
        self.accum = max(1, int(cfg.gradient_accumulation_steps))
        self._micro = 0

    def lr_factor(self, step: int) -> float:
        """Linear warm-up then cosine to cfg.min_lr (CosineAnnealingLR, train.py:410-414,495-521)."""
        if self.warmup_steps > 0 and step <= self.warmup_steps:
            return step / self.warmup_steps
        t = min(step, self.total_steps)
        return 0.5 * (1 + math.cos(math.pi * t / max(1, self.total_steps)))

    def current_lr(self, group_lr: float, step: int) -> float:
        """The learning rate step ``step`` applies to a group whose base rate is ``group_lr``: linear warm-up, then
        CosineAnnealingLR's closed form ``eta_min + (base - eta_min) (1 + cos(pi t / T_max)) / 2``.  The ONE place
        the schedule is evaluated: the optimizer step, ``state_dict()['param_groups'][i]['lr']`` and the scheduler
        view's ``_last_lr`` all come from here (torch's scheduler recurses on ``group['lr']``, so a checkpoint
        must carry the rate that was really applied)."""
        f = self.lr_factor(step)
        if self.warmup_steps > 0 and step <= self.warmup_steps:
            return group_lr * f
        min_lr = float(getattr(self.cfg, "min_lr", 0.0))
        return min_lr + (group_lr - min_lr) * f

    def micro_step(self, *fwd_args, **fwd_kwargs) -> torch.Tensor:
        """forward + backward of one micro-batch; the optimizer runs on the last one."""
        r = self.reducer
        last = (self._micro + 1) % self.accum == 0
        if self._micro % self.accum == 0:
            r.begin_step()
        if self.comm is not None:
            with use_comm(self.comm):                                       # the forward's scalar reductions
                loss = self.forward(self.cfg, *fwd_args, **fwd_kwargs)
        else:
            loss = self.forward(self.cfg, *fwd_args, **fwd_kwargs)         # train.py:524
        scaled = loss / self.accum                                          # train.py:527
        self.arena.reset()
        if last:
            scaled.backward()                                               # train.py:567
            r.finish_backward()
            self.optimizer_step()
        else:
            with r.no_sync():                                               # train.py:562-564
                scaled.backward()
                r.finish_backward()
        self._micro += 1
        return loss.detach()

    def optimizer_step(self):
        r, cfg = self.reducer, self.cfg
        for p in r.params:
            # every trainable gradient must have gone through the flat slots (the ZF layer nodes); a parameter that
            # was used through plain autograd (e.g. Zero._forward's torch.cat on embd_soi) would be silently ignored
            if p.grad is not None:
                raise RuntimeError(f"{r.names[id(p)]} received an autograd .grad outside the flat gradient buffer: "
                                   "drive the Trainer through forward_hf / forward_from_ids, not Zero._forward")
        self.global_step += 1
        bump_weights_epoch()           # the fused AdamW below writes the parameters through raw pointers
        self.sumsq.zero_()
        max_norm = float(cfg.grad_clip)
        if max_norm > 0:
            ops.sumsq_(r.flat_grad, self.sumsq, scale=r.grad_scale)         # clip_grad_norm_ (train.py:574-576)
        for o, e, (lr, wd) in self.segments:
            cur = self.current_lr(lr, self.global_step)
            ops.adamw_step_(r.flat_param[o:e], r.flat_grad[o:e], self.m[o:e], self.v[o:e], lr=cur,
                            beta1=cfg.adamw_beta1, beta2=cfg.adamw_beta2, eps=1e-8, weight_decay=wd,
                            step=self.global_step, max_norm=max_norm, sumsq=self.sumsq, grad_scale=r.grad_scale)

    def grad_norm(self) -> torch.Tensor:
        return self.sumsq.sqrt()

    # ---- optimizer / scheduler state in torch's own layouts (ref:utils.py:357-377,504-514 save and load
    # ``optimizer.state_dict()`` of torch.optim.AdamW and ``scheduler.state_dict()`` of CosineAnnealingLR)
    def state_dict(self) -> dict:
        """torch.optim.AdamW layout: ``state[i] = {step, exp_avg, exp_avg_sq}`` with i numbering the parameters of
        the reference-style groups in order, ``param_groups`` with the hyper-parameters and index lists.  The
        moments are this Trainer's fp32 ones (torch casts them to the parameter dtype on load)."""
        r, cfg = self.reducer, self.cfg
        state, groups, i = {}, [], 0
        lr_now = lambda base: (self.current_lr(base, self.global_step) if self.global_step > 0
                               else (0.0 if self.warmup_steps > 0 else base))
        for g in self._groups:
            idx = []
            for p in g["params"]:
                if self.global_step > 0:
                    o, n = r.offset[id(p)], p.numel()
                    state[i] = {"step": torch.tensor(float(self.global_step)),
                                "exp_avg": self.m[o:o + n].detach().to("cpu", copy=True).view(p.shape),
                                "exp_avg_sq": self.v[o:o + n].detach().to("cpu", copy=True).view(p.shape)}
                idx.append(i)
                i += 1
            groups.append({"lr": lr_now(g["lr"]), "initial_lr": g["lr"], "betas": (float(cfg.adamw_beta1), float(cfg.adamw_beta2)),
                           "eps": 1e-8, "weight_decay": g["weight_decay"], "amsgrad": False, "maximize": False,
                           "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                           "decoupled_weight_decay": True, "params": idx})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd: dict):
        """Accepts torch.optim.AdamW's layout (a checkpoint written by the reference, moments in any float dtype)
        and the name-keyed layout of round 1.  Raises on a structural mismatch (utils.load_checkpoint reports it
        and falls back to the checkpoint's global_step, as the reference's try / except does)."""
        r = self.reducer
        if "param_groups" in sd:
            flat = [p for g in self._groups for p in g["params"]]
            n_saved = sum(len(g["params"]) for g in sd["param_groups"])
            if n_saved != len(flat):
                raise ValueError(f"optimizer state has {n_saved} parameters, this run trains {len(flat)}")
            step = 0
            for i, p in enumerate(flat):
                st = sd["state"].get(i, sd["state"].get(str(i)))
                if st is None:
                    continue
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state {i}: shape {tuple(st['exp_avg'].shape)} != {tuple(p.shape)}")
                o, n = r.offset[id(p)], p.numel()
                self.m[o:o + n].copy_(st["exp_avg"].reshape(-1).to(torch.float32))
                self.v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1).to(torch.float32))
                step = max(step, int(float(st["step"])))
            self.global_step = step
            self._micro = step * self.accum
        else:
            for p in r.params:
                o, n, name = r.offset[id(p)], p.numel(), r.names[id(p)]
                if name in sd["exp_avg"]:
                    self.m[o:o + n].copy_(sd["exp_avg"][name].reshape(-1))
                    self.v[o:o + n].copy_(sd["exp_avg_sq"][name].reshape(-1))
            self.global_step = int(sd.get("global_step", 0))
            self._micro = int(sd.get("micro", 0))
        bump_weights_epoch()

    @property
    def scheduler(self):
        """CosineAnnealingLR-shaped view of the schedule for save_checkpoint / load_checkpoint."""
        return _SchedulerView(self)


class _SchedulerView:
    def __init__(self, trainer: Trainer):
        self.t = trainer

    def state_dict(self) -> dict:
        t = self.t
        return {"T_max": t.total_steps, "eta_min": float(getattr(t.cfg, "min_lr", 0.0)),
                "base_lrs": [g["lr"] for g in t._groups], "last_epoch": t.global_step,
                "_step_count": t.global_step + 1, "_last_lr": [t.current_lr(g["lr"], max(t.global_step, 1)) for g in t._groups]}

    def load_state_dict(self, sd: dict):
        self.t.global_step = int(sd["last_epoch"])
        self.t._micro = self.t.global_step * self.t.accum
