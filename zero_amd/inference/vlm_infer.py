"""Zero-shot grafting inference (ref:inference/vlm_infer.py:191-277) from token ids on: encoder +
adapter prefix (``Zero._forward``, the same function as on the training path), splice into the prompt
at the image token, pad, greedy ``generate`` on the FULL decoder (``enable_translator=False``,
ref:vlm_infer.py:57) — on one GPU or tensor parallel (distributed/tensor_parallel.py).
The tokenizer / chat template and the image file preprocessing stay outside (no tokenizer offline,
SURVEY.md §8c): callers pass pixel tensors and id lists that contain ``img_id`` once."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from ..distributed.tensor_parallel import TPContext
from .decode import generate


@torch.no_grad()
def graft_inputs(model, imgs: List[torch.Tensor], input_ids: Sequence[Sequence[int]], img_id: int) -> torch.Tensor:
    """ref:vlm_infer.py:201,231-257: [wte(ids before) | soi, image tokens, eoi | wte(ids after)], right-padded with 0."""
    device = imgs[0].device
    embds_imgs = model._forward(imgs)                                    # vlm_infer.py:201
    wte = model.decoder.model.get_input_embeddings()                     # vlm_infer.py:214
    rows = []
    for i, ids in enumerate(input_ids):
        ids = list(ids)
        at = ids.index(img_id)                                           # vlm_infer.py:234
        e1 = wte(torch.tensor(ids[:at], dtype=torch.long, device=device))
        e2 = wte(torch.tensor(ids[at + 1:], dtype=torch.long, device=device))
        rows.append(torch.cat([e1, embds_imgs[i].to(e1.dtype), e2], dim=0))
    S = max(r.shape[0] for r in rows)
    out = torch.zeros((len(rows), S, rows[0].shape[1]), dtype=rows[0].dtype, device=device)   # pad_sequence(0.0)
    for i, r in enumerate(rows):
        out[i, : r.shape[0]] = r
    return out


@torch.no_grad()
def graft_generate(model, imgs: List[torch.Tensor], input_ids: Sequence[Sequence[int]], img_id: int,
                   max_new_tokens: int = 1024, tp: Optional[TPContext] = None, **gen_kwargs) -> torch.Tensor:
    """New token ids [B, <= max_new_tokens] (greedy, KV cache), ref:vlm_infer.py:259-270."""
    x = graft_inputs(model, imgs, input_ids, img_id)
    mask = torch.ones(x.shape[:2], device=x.device)                      # vlm_infer.py:260-263
    return generate(model.decoder, x, attention_mask=mask, do_sample=False, max_new_tokens=max_new_tokens,
                    use_cache=True, tp=tp, **gen_kwargs)
