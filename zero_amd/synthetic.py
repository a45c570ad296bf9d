"""Synthetic benchmark inputs of SURVEY.md §8d (product side: bench.py and examples draw their batches here, not
from the test oracle).  Shapes follow the shipped config: 336-pixel images (ref:configs/config_base.py:111-112), one
image token per sample (pad mode), a response-only loss mask (ref:configs/zero/train_llama.py:119)."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

IGNORE_INDEX = -100          # ref:methods/engine_ar_instruction_llama.py:22


def synth_batch(image: int, vocab: int, B: int, seed: int, *, n_pre: int = 21, n_post: int = 168, n_resp: int = 128,
                ragged: bool = False, img_id: Optional[int] = None,
                text_only: Sequence[int] = ()) -> Tuple[List[torch.Tensor], List[List[int]], List[List[int]], int, int]:
    """B samples: an image ~ N(0, 1) [1, 3, image, image] (CLIP-normalised pixels have about unit scale), ``n_pre``
    token ids before the image slot and ``n_post`` after it (uniform below the special ids at the top of the
    vocabulary); labels = the ids on the last ``n_resp`` positions, IGNORE_INDEX elsewhere.  ``ragged`` draws a
    different suffix length per sample (padding parity); ``text_only`` samples carry no image token but still come
    with a zero image (ref:loader.py:204-206).  Returns (imgs, ids, targets, img_id, pad_id)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    hi = vocab - 256 if vocab > 512 else vocab - 8
    img_id = vocab - 2 if img_id is None else img_id
    pad_id = vocab - 1
    imgs, ids_all, tgt_all = [], [], []
    for b in range(B):
        text = b in text_only
        imgs.append(torch.zeros(1, 3, image, image) if text else torch.randn(1, 3, image, image, generator=g))
        npost = int(torch.randint(max(2, n_post // 4), n_post + 1, (1,), generator=g)) if ragged else n_post
        pre = torch.randint(0, hi, (n_pre,), generator=g).tolist()
        post = torch.randint(0, hi, (npost,), generator=g).tolist()
        nresp = min(n_resp, npost)
        slot = [] if text else [img_id]
        ids_all.append(pre + slot + post)
        tgt_all.append([IGNORE_INDEX] * (n_pre + len(slot) + npost - nresp) + post[npost - nresp:])
    return imgs, ids_all, tgt_all, img_id, pad_id
