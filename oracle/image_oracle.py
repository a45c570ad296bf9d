"""CPU restatement of the image data path.  TEST INFRASTRUCTURE ONLY: imported by
tests/ (and nothing in zero_amd/); the product path is zero_amd/loader.py + csrc/image.hip.

What it restates
  * ref:loader.py:609-620 ``expand2square``; :691-728 ``select_best_resolution``; :731-767
    ``resize_and_pad_image``; :770-789 ``divide_to_patches``; :623-688 ``process_anyres_image``;
    :171-175 the "pad" branch of ``JsonDataset.__getitem__``.
  * the CLIPImageProcessor the reference gets from transformers (PIL backend): resize to the shortest
    edge (tf:image_processing_backends.py:521-570, tf:image_transforms.py get_resize_output_image_size),
    centre crop (:602-617), rescale (tf:image_transforms.py:118-122: float64 multiply, cast to float32),
    normalize (:417-437: float32 subtract / divide).
  * ``PIL.Image.resize`` for 8-bit RGB with the BICUBIC filter.  Third-party, absent from the reference
    tree: Pillow (unpinned by the reference; 12.2.0 here), src/libImaging/Resample.c —
    ``precompute_coeffs``, ``normalize_coeffs_8bpc``, ``ImagingResampleHorizontal_8bpc``,
    ``ImagingResampleVertical_8bpc``, ``ImagingResample`` (horizontal pass first, u8 rounding between the
    passes) and ``Image.resize``'s same-size shortcut.

Pinned by tests/golden/image.safetensors, written by tests/golden/make_golden_image.py from the real PIL
and the real transformers CLIPImageProcessor (PIL backend) in the build container; bit-exact.
Scalar Python loops over the output index, numpy over the other axis: small cases only.
"""
from __future__ import annotations

import math
import re
from typing import List, Sequence, Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def bicubic_filter(x: float) -> float:
    """Resample.c bicubic_filter, a = -0.5 (Keys)."""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, in0: float, in1: float, out_size: int, support_: float = 2.0):
    """Resample.c precompute_coeffs: per output index (first source index, tap count) and ksize doubles."""
    filterscale = scale = float(in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support_ * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds, kk = [], []
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        ww = 0.0
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)          # C (int): truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [0.0] * ksize
        for x in range(xmax):
            w = bicubic_filter((x + xmin - center + 0.5) * ss)
            k[x] = w
            ww += w
        for x in range(xmax):
            if ww != 0.0:
                k[x] /= ww
        bounds.append((xmin, xmax))
        kk.append(k)
    return ksize, bounds, kk


def normalize_coeffs_8bpc(kk: List[List[float]]) -> List[List[int]]:
    out = []
    for k in kk:
        out.append([int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)) for v in k])
    return out


def _clip8(acc: np.ndarray) -> np.ndarray:
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resample_horizontal(img: np.ndarray, out_w: int) -> np.ndarray:
    h, w, c = img.shape
    _, bounds, kk = precompute_coeffs(w, 0.0, float(w), out_w)
    ik = normalize_coeffs_8bpc(kk)
    out = np.empty((h, out_w, c), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_w):
        xmin, n = bounds[xx]
        acc = np.full((h, c), 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(n):
            acc += src[:, xmin + x, :] * ik[xx][x]
        out[:, xx, :] = _clip8(acc)
    return out


def resample_vertical(img: np.ndarray, out_h: int) -> np.ndarray:
    return resample_horizontal(img.transpose(1, 0, 2), out_h).transpose(1, 0, 2).copy()


def pil_resize(img: np.ndarray, size: Tuple[int, int]) -> np.ndarray:
    """``Image.resize((w, h))`` with the default BICUBIC filter on an RGB u8 image [h][w][3]."""
    w, h = size
    ih, iw, _ = img.shape
    if (iw, ih) == (w, h):
        return img.copy()
    out = img
    if w != iw:
        out = resample_horizontal(out, w)
    if h != ih:
        out = resample_vertical(out, h)
    return out


def expand2square(img: np.ndarray, bg: Sequence[int]) -> np.ndarray:
    """ref:loader.py:609-620: centre the image on a square canvas of the background colour."""
    h, w, _ = img.shape
    if w == h:
        return img
    s = max(w, h)
    out = np.empty((s, s, 3), dtype=np.uint8)
    out[:] = np.asarray(bg, dtype=np.uint8)
    if w > h:
        out[(w - h) // 2:(w - h) // 2 + h] = img
    else:
        out[:, (h - w) // 2:(h - w) // 2 + w] = img
    return out


def normalize_u8(img: np.ndarray, mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    """u8 [h][w][3] -> float32 [3][h][w]: rescale by 1/255 in float64, cast, then float32 (x - mean) / std."""
    x = (img.astype(np.float64) * (1 / 255)).astype(np.float32).transpose(2, 0, 1)
    m = np.array(mean, dtype=np.float32)[:, None, None]
    s = np.array(std, dtype=np.float32)[:, None, None]
    return (x - m) / s


def clip_preprocess(img: np.ndarray, size: int = 336, crop: int = 336, mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    """CLIPImageProcessor.preprocess on one RGB u8 image -> float32 [1][3][crop][crop]."""
    h, w, _ = img.shape
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
    r = pil_resize(img, (nw, nh))
    top, left = (nh - crop) // 2, (nw - crop) // 2
    if top < 0 or left < 0:
        raise ValueError("crop larger than the resized image")
    r = r[top:top + crop, left:left + crop]
    return normalize_u8(r, mean, std)[None]


def pad_mode_pixels(img: np.ndarray, size: int = 336, mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    """ref:loader.py:171-175 with bg_value from ref:loader.py:108."""
    bg = tuple(int(x * 255) for x in mean)
    return clip_preprocess(expand2square(img, bg), size, size, mean, std)


def parse_grid_pinpoints(grid_pinpoints, patch_size: int) -> List[Tuple[int, int]]:
    """ref:loader.py:636-664: "(1x1),...,(6x6)" -> every (i, j) in the range, times the patch size."""
    if isinstance(grid_pinpoints, str) and "x" in grid_pinpoints:
        m = re.findall(r"\((\d+)x(\d+)\)", grid_pinpoints)
        (a0, b0), (a1, b1) = tuple(map(int, m[0])), tuple(map(int, m[-1]))
        return [(i * patch_size, j * patch_size) for i in range(a0, a1 + 1) for j in range(b0, b1 + 1)]
    if isinstance(grid_pinpoints, str):
        import ast
        grid_pinpoints = ast.literal_eval(grid_pinpoints)
    return [tuple(p) for p in grid_pinpoints]


def select_best_resolution(original_size, possible_resolutions):
    """ref:loader.py:691-728: largest effective resolution, ties broken by least waste."""
    ow, oh = original_size
    best, best_eff, best_waste = None, 0, float("inf")
    for w, h in possible_resolutions:
        scale = min(w / ow, h / oh)
        dw, dh = int(ow * scale), int(oh * scale)
        eff = min(dw * dh, ow * oh)
        waste = w * h - eff
        if eff > best_eff or (eff == best_eff and waste < best_waste):
            best, best_eff, best_waste = (w, h), eff, waste
    return best


def resize_and_pad_image(img: np.ndarray, target: Tuple[int, int], bg) -> np.ndarray:
    """ref:loader.py:731-767."""
    oh, ow, _ = img.shape
    tw, th = target
    sw, sh = tw / ow, th / oh
    if sw < sh:
        nw, nh = tw, min(math.ceil(oh * sw), th)
    else:
        nh, nw = th, min(math.ceil(ow * sh), tw)
    r = pil_resize(img, (nw, nh))
    out = np.empty((th, tw, 3), dtype=np.uint8)
    out[:] = np.asarray(bg, dtype=np.uint8)
    px, py = (tw - nw) // 2, (th - nh) // 2
    out[py:py + nh, px:px + nw] = r
    return out


def divide_to_patches(img: np.ndarray, ps: int) -> List[np.ndarray]:
    """ref:loader.py:770-789 (row-major tiles)."""
    h, w, _ = img.shape
    return [img[i:i + ps, j:j + ps] for i in range(0, h, ps) for j in range(0, w, ps)]


def process_anyres_image(img: np.ndarray, size: int, grid_pinpoints, bg, mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    """ref:loader.py:623-688 -> float32 [1 + tiles][3][size][size]: the whole image squashed to size x size
    first (the reference's acknowledged resize-instead-of-pad, :675-677), then the tiles of the padded canvas."""
    res = parse_grid_pinpoints(grid_pinpoints, size)
    h, w, _ = img.shape
    best = select_best_resolution((w, h), res)
    padded = resize_and_pad_image(img, best, bg)
    tiles = divide_to_patches(padded, size)
    base = pil_resize(img, (size, size))
    return np.concatenate([clip_preprocess(t, size, size, mean, std) for t in [base] + tiles], axis=0)
