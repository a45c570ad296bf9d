"""``forward_hf(self, cfg, data, tokenizer) -> loss`` — the method the reference binds onto the
``Zero`` instance (ref:train.py:213-214; ref:methods/engine_ar_instruction_llama.py:30-318).

Stage 1 (host, strings): chat-template tokenisation + target masking, the same rules as
engine:59-183, driven through whatever ``tokenizer`` is passed.
Stage 2 (``forward_from_ids``, the hot path): encoder+adapter rows -> ONE fused
gather that builds ``inputs_embeds`` (wte rows | image rows | soi/eoi | zero padding,
engine:186-240) -> surrogate decoder -> final norm + lm_head + shifted token-sum cross
entropy on the non-ignored rows only -> cross-rank sums -> loss (engine:275-316).
"""
from __future__ import annotations

import random
from copy import deepcopy
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .. import functional as ZF
from ..distributed.torchshard import reduce_sum, reduce_sum_pair

IGNORE_INDEX = -100

CHAT_TEMPLATE = (
    "{% if not add_generation_prompt is defined %}{% set add_generation_prompt = false %}{% endif %}"
    "{% for message in messages %}{{'<|start_header_id|>' + message['role'] + '<|end_header_id|>' + '\n' + "
    "message['content'] + '<|eot_id|>'}}{% endfor %}"
    "{% if add_generation_prompt %}{{ '<|start_header_id|>assistant<|end_header_id|>\n' }}{% endif %}")


def tokenize_conversations(cfg, conversations, tokenizer) -> Tuple[List[List[int]], List[List[int]], int, int]:
    """engine:41-45,59-183: per-sample ``input_id`` / ``target`` lists, plus (img_id, pad_id)."""
    img_token_str = "<|reserved_special_token_0|>"
    img_id = tokenizer.convert_tokens_to_ids(img_token_str)
    bot_id = tokenizer.convert_tokens_to_ids("<|begin_of_text|>")
    pad_id = tokenizer.convert_tokens_to_ids("<|end_of_text|>")
    tokenizer = deepcopy(tokenizer)
    tokenizer.chat_template = CHAT_TEMPLATE
    templates = cfg.image_prompt_templates
    input_ids, targets = [], []
    for conv in conversations:
        input_id, target = [bot_id], []
        target += [IGNORE_INDEX] * len(input_id) if cfg.ignore_instruction_token_ids else list(input_id)
        for usr, ast in zip(conv[::2], conv[1::2]):
            assert usr["role"] in ["human", "user"]
            assert ast["role"] in ["model", "assistant"]
            usr_text = usr["content"]
            has_img = "<image>" in usr_text
            apply_img_prompt = False
            if has_img and cfg.enable_image_prompt:
                apply_img_prompt = True
                usr_text = usr_text.replace("<image>\n", "").replace("\n<image>", "").strip()
                image_prompt = random.choice(templates) if cfg.enable_random_image_prompt else templates[0]
                usr_text = usr_text[0].lower() + usr_text[1:]
                if cfg.split_image_tokens_in_prompt_with_newline:
                    usr_text = f"{image_prompt}\n{img_token_str},\n{usr_text}"
                else:
                    usr_text = f"{image_prompt}{img_token_str}, {usr_text}"
            if has_img and not apply_img_prompt:
                usr_text = img_token_str.join(usr_text.split("<image>"))
            encode_id = tokenizer.apply_chat_template([{"role": "user", "content": usr_text}])
            input_id += encode_id
            if cfg.ignore_instruction_token_ids:
                if cfg.include_special_tokens_in_masking_instruction_tokens:
                    target += ([encode_id[0]] + [IGNORE_INDEX] + [encode_id[2]]
                               + [IGNORE_INDEX] * (len(encode_id) - 4) + [encode_id[-1]])
                else:
                    target += [IGNORE_INDEX] * len(encode_id)
            else:
                target += encode_id
            encode_id = tokenizer.apply_chat_template([{"role": "assistant", "content": ast["content"]}])
            input_id += encode_id
            if cfg.ignore_instruction_token_ids:
                if cfg.include_special_tokens_in_masking_instruction_tokens:
                    target += [encode_id[0]] + [IGNORE_INDEX] + [encode_id[2]] + encode_id[3:]
                else:
                    target += [IGNORE_INDEX] * 3 + encode_id[3:]
            else:
                target += encode_id
        assert len(input_id) == len(target), f"{len(input_id)} != {len(target)}"
        input_ids.append(input_id)
        targets.append(target)
    return input_ids, targets, img_id, pad_id


def build_plan(input_ids, targets, n_img_rows: Sequence[int], img_id: int, pad_id: int, max_seq_len: int):
    """Host-side index plan of engine:186-240 + :283-285 (numpy, no device work).

    n_img_rows[b] = adapter rows of sample b (P_b * T).  The image table on the device is
    [all adapter rows ; soi ; eoi]; codes follow ops.gather_rows."""
    B = len(input_ids)
    total_rows = int(sum(n_img_rows))
    soi_code, eoi_code = -2 - total_rows, -2 - (total_rows + 1)
    seqs, tgts = [], []
    row0 = 0
    for b, (ids, tgt) in enumerate(zip(input_ids, targets)):
        ids, tgt = list(ids), list(tgt)
        n = int(n_img_rows[b])
        if img_id in ids:
            at = ids.index(img_id)
            ids1 = ids[:at]
            ids2 = ids[at + 1:]
            ids2 = ids2[: max_seq_len - len(ids1) - (n + 2)]
            img_codes = [soi_code] + [-2 - (row0 + j) for j in range(n)] + [eoi_code]
            seqs.append(ids1 + img_codes + ids2)
            tgts.append(tgt[:at] + [IGNORE_INDEX] * (n + 2) + tgt[at + 1:])
        else:
            seqs.append(ids)
            tgts.append(tgt)
        row0 += n
    S = min(max(len(s) for s in seqs), max_seq_len)
    St = min(max(len(t) for t in tgts), max_seq_len)
    if S != St:
        raise ValueError(f"input/target length mismatch after assembly: {S} vs {St}")
    codes = np.full((B, S), -1, dtype=np.int32)
    tgt_arr = np.full((B, S), IGNORE_INDEX, dtype=np.int64)
    for b in range(B):
        s = seqs[b][:S]
        codes[b, :len(s)] = s
        t = tgts[b][:S]
        tgt_arr[b, :len(t)] = t
    # shifted labels: row (b, s) predicts target[b, s+1]; pad -> ignore (engine:283-285)
    labels = np.full((B, S), IGNORE_INDEX, dtype=np.int64)
    labels[:, :-1] = tgt_arr[:, 1:]
    labels[labels == pad_id] = IGNORE_INDEX
    flat = labels.reshape(-1)
    rows = np.nonzero(flat != IGNORE_INDEX)[0].astype(np.int32)
    row_labels = flat[rows].astype(np.int32)
    inv_rows = np.full(B * S, -1, dtype=np.int32)
    inv_rows[rows] = np.arange(rows.size, dtype=np.int32)
    # backward maps of the splice
    fc = codes.reshape(-1)
    inv_codes = np.full(total_rows, -1, dtype=np.int32)
    is_img = (fc <= -2) & (fc > soi_code)
    pos = np.nonzero(is_img)[0]
    inv_codes[-2 - fc[pos]] = pos
    soi_pos = np.full(B, -1, dtype=np.int32)
    eoi_pos = np.full(B, -1, dtype=np.int32)
    for arr, code in ((soi_pos, soi_code), (eoi_pos, eoi_code)):
        p = np.nonzero(fc == code)[0]
        arr[p // S] = p
    return dict(B=B, S=S, codes=fc, target=tgt_arr, rows=rows, row_labels=row_labels, inv_rows=inv_rows,
                inv_codes=inv_codes, soi_pos=soi_pos, eoi_pos=eoi_pos, n_valid=int(rows.size),
                labels_full=flat.astype(np.int32))


class _NanWatch:
    """engine:315-316 raises ValueError when the loss is NaN.  Reading the loss on the host right
    after the forward drains the GPU once per step (the backward cannot be enqueued behind it), so
    the flag travels to pinned host memory asynchronously and is examined, without blocking, at the
    next forward(s): the error is raised one or two steps late instead of stalling every step.
    ``cfg.check_nan_loss = "sync"`` restores the reference's immediate check."""

    def __init__(self):
        self.pending = []      # (event, pinned flag)
        self.free = []         # pinned one-byte flags ready for reuse (no per-step host allocation)

    def check_and_arm(self, loss, device, sync: bool = False):
        keep = []
        for ev, flag in self.pending:
            if ev.query():
                if bool(flag.item()):
                    self.pending = []
                    rank = dist.get_rank() if dist.is_initialized() else 0
                    raise ValueError(f"loss is NaN on rank {rank}")
                self.free.append((ev, flag))
            else:
                keep.append((ev, flag))
        self.pending = keep
        if device.type != "cuda":
            if bool(loss.isnan()):
                rank = dist.get_rank() if dist.is_initialized() else 0
                raise ValueError(f"loss is NaN on rank {rank}")
            return
        ev, flag = self.free.pop() if self.free else (torch.cuda.Event(), torch.empty(1, dtype=torch.bool).pin_memory())
        flag.copy_(loss.detach().isnan().reshape(1), non_blocking=True)
        ev.record()
        self.pending.append((ev, flag))

    def flush(self):
        """Blocking check of everything in flight (end of training / tests)."""
        pend, self.pending = self.pending, []
        for ev, flag in pend:
            ev.synchronize()
            if bool(flag.item()):
                rank = dist.get_rank() if dist.is_initialized() else 0
                raise ValueError(f"loss is NaN on rank {rank}")


_NAN_WATCH = _NanWatch()


def _upload(plan, device, keys):
    """One host->device copy for all int32 index arrays."""
    parts = [plan[k] for k in keys]
    sizes = [p.size for p in parts]
    buf = torch.from_numpy(np.concatenate(parts).astype(np.int32, copy=False))
    if device.type == "cuda":
        buf = buf.pin_memory().to(device, non_blocking=True)
    return dict(zip(keys, torch.split(buf, sizes)))


def forward_from_ids(self, cfg, imgs: List[torch.Tensor], input_ids, targets, img_id: int, pad_id: int,
                     return_aux: bool = False):
    if getattr(cfg, "enable_balance_loss_weights", False):
        # engine:243-255,299-301,460-550 multiply the per-token loss by batch-mean response weights (two extra
        # MAX / SUM all-reduces); off in every shipped config (config_base.py:140) and not built here — a config
        # that turns it on must not silently train on a different objective
        raise NotImplementedError("enable_balance_loss_weights (_calc_loss_weights_with_batchmean) is not implemented")
    device = imgs[0].device
    rows, split_sizes, T = self._encode_rows(imgs)                       # engine:49
    plan = build_plan(input_ids, targets, [p * T for p in split_sizes], img_id, pad_id, cfg.max_seq_len)
    B, S = plan["B"], plan["S"]
    plan["n_valid_arr"] = np.asarray([plan["n_valid"]], dtype=np.int32)
    dev = _upload(plan, device, ["codes", "rows", "row_labels", "inv_rows", "inv_codes", "soi_pos", "eoi_pos",
                                 "labels_full", "n_valid_arr"])
    wte = self.decoder.model.get_input_embeddings().weight                # engine:186
    inputs_embeds = ZF.SpliceFn.apply(rows, self.embd_soi, self.embd_eoi, wte.detach(), dev["codes"],
                                      dev["inv_codes"], dev["soi_pos"], dev["eoi_pos"], B, S)
    dec = self.decoder
    if hasattr(dec.model, "run_layers"):
        # only the labelled rows of the final hidden state are read: the last (frozen) layer need not form the others
        hidden = dec.model.run_layers(inputs_embeds, tail_rows=(dev["rows"], dev["inv_rows"]))
        compact = hidden.shape[:2] != inputs_embeds.shape[:2]
        loss = ZF.LMHeadLossFn.apply(hidden, dec.model.norm.weight.detach(), dec.lm_head.weight.detach(),
                                     None if compact else dev["rows"], dev["row_labels"],
                                     None if compact else dev["inv_rows"],
                                     dec.model.norm.variance_epsilon, (dec.lm_head.weight,))
        logits = None
    else:   # any module with the HF call contract (engine:276-280)
        logits = dec(inputs_embeds=inputs_embeds, use_cache=False)["logits"]
        loss = ZF.CrossEntropySumFn.apply(logits.reshape(B * S, -1), dev["labels_full"])
    num_tokens = dev["n_valid_arr"][0].to(torch.int64)                    # rides in the one pinned upload: no sync
    loss_sum_local = loss.detach()
    if cfg.reduce_batch_loss:                                             # engine:308-310
        loss, num_tokens = reduce_sum_pair(loss, num_tokens)
    loss = loss / num_tokens                                              # engine:313
    chk = getattr(cfg, "check_nan_loss", True)
    if chk == "sync":                                                     # engine:315-316, as the reference does it
        if loss.isnan():
            rank = dist.get_rank() if dist.is_initialized() else 0
            raise ValueError(f"loss is NaN on rank {rank}")
    elif chk:
        _NAN_WATCH.check_and_arm(loss, device)
    if return_aux:
        return loss, dict(inputs_embeds=inputs_embeds, target=torch.from_numpy(plan["target"]), logits=logits,
                          loss_sum=loss_sum_local, num_tokens=num_tokens, plan=plan, img_rows=rows)
    return loss


def forward_hf(self, cfg: object, data, tokenizer: object):
    imgs, conversations, keys = data
    device = cfg.device
    imgs = [img.to(device) for img in imgs]                               # engine:48
    input_ids, targets, img_id, pad_id = tokenize_conversations(cfg, conversations, tokenizer)
    return forward_from_ids(self, cfg, imgs, input_ids, targets, img_id, pad_id)
