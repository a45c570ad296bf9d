"""The image half of the reference's data path on the device.

The reference prepares every sample on CPU DataLoader workers with PIL (ref:loader.py:163-208):
``expand2square`` with the mean colour, then ``CLIPImageProcessor`` (bicubic resize to 336, centre crop,
1/255 rescale, mean/std normalisation), or the LLaVA-NeXT "anyres" tiling (ref:loader.py:623-789).  Here the
decoded u8 RGB image is uploaded once and the same steps run as three byte kernels (csrc/image.hip).
Function and attribute names follow ref:loader.py so a dataset can swap them in; images travel as
``DeviceImage`` (a u8 tensor seen through a paste/crop window) instead of ``PIL.Image``.

Results are BIT-IDENTICAL to PIL + the PIL-backend CLIPImageProcessor: the resampler is Pillow's 8-bit
fixed-point one, and its coefficient tables are made here in float64 exactly as Pillow's
``precompute_coeffs`` / ``normalize_coeffs_8bpc`` (Resample.c) make them.  JSON parsing, file decoding,
samplers and collation (ref:loader.py:47-118,240-601) stay host-side concerns and are not rebuilt.
"""
from __future__ import annotations

import ast
import functools
import math
import re
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import check, lib

OPENAI_CLIP_MEAN = [0.48145466, 0.4578275, 0.40821073]
OPENAI_CLIP_STD = [0.26862954, 0.26130258, 0.27577711]
PRECISION_BITS = 32 - 8 - 2


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# ---- Pillow's coefficient tables (Resample.c precompute_coeffs + normalize_coeffs_8bpc, bicubic) ----------

def _bicubic(x: np.ndarray) -> np.ndarray:
    x = np.abs(x)
    near = ((1.5 * x - 2.5) * x) * x + 1
    far = (((x - 5) * x + 8) * x - 4) * -0.5
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


@functools.lru_cache(maxsize=256)
def resample_coeffs(in_size: int, out_size: int) -> Tuple[int, np.ndarray, np.ndarray]:
    """(ksize, bounds int32 [out, 2] = (first source index, taps), kk int32 [out, ksize]) of a bicubic
    resize in_size -> out_size over the whole axis.  float64 throughout, operation for operation what the
    C code does, so the integer coefficients are the ones PIL uses."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    ss = 1.0 / filterscale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size)
    n = xmax - xmin
    taps = np.arange(ksize, dtype=np.int64)
    w = _bicubic(((taps[None, :] + xmin[:, None]).astype(np.float64) - center[:, None] + 0.5) * ss)
    w = np.where(taps[None, :] < n[:, None], w, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for i in range(ksize):                      # the C loop's summation order
        ww = ww + w[:, i]
    k = np.where(ww[:, None] != 0.0, w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    fixed = np.where(k < 0, np.trunc(-0.5 + k * (1 << PRECISION_BITS)), np.trunc(0.5 + k * (1 << PRECISION_BITS)))
    bounds = np.stack([xmin, n], axis=1).astype(np.int32)
    return ksize, np.ascontiguousarray(bounds), np.ascontiguousarray(fixed.astype(np.int32))


_DEV_TABLES: Dict[tuple, tuple] = {}


def _device_coeffs(in_size: int, out_size: int, device):
    key = (in_size, out_size, str(device))
    hit = _DEV_TABLES.get(key)
    if hit is None:
        ksize, bounds, kk = resample_coeffs(in_size, out_size)
        if len(_DEV_TABLES) > 512:
            _DEV_TABLES.clear()
        hit = _DEV_TABLES[key] = (ksize, torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device))
    return hit


def _pack_rgb(bg) -> int:
    if bg is None:
        bg = (0, 0, 0)
    if isinstance(bg, int):
        bg = (bg, bg, bg)
    r, g, b = (int(v) & 255 for v in bg)
    return r | (g << 8) | (b << 16)


class DeviceImage:
    """An RGB u8 image in HBM: ``data`` [h, w, 3] pasted at ``paste`` = (x, y) onto a ``canvas`` = (W, H) of
    colour ``bg`` (negative paste = crop).  Padding, pasting and cropping only edit the window; the kernels
    read through it.  ``size`` is PIL's (width, height)."""

    mode = "RGB"

    def __init__(self, data: torch.Tensor, canvas: Optional[Tuple[int, int]] = None, paste: Tuple[int, int] = (0, 0),
                 bg=(0, 0, 0)):
        if data.dtype != torch.uint8 or data.dim() != 3 or data.shape[2] != 3:
            raise ValueError(f"DeviceImage: expected uint8 [h, w, 3], got {data.dtype} {tuple(data.shape)}")
        if not data.is_cuda:
            raise ValueError("DeviceImage: zero_amd kernels need a ROCm device tensor (no CPU path exists)")
        if data.stride(2) != 1 or data.stride(1) != 3:
            data = data.contiguous()
        self.data = data
        self.canvas = tuple(canvas) if canvas is not None else (data.shape[1], data.shape[0])
        self.paste = tuple(paste)
        self.bg = bg

    @classmethod
    def from_any(cls, img, device=None) -> "DeviceImage":
        """PIL image / numpy / torch u8 [h, w, 3] -> DeviceImage on ``device`` (one host-to-device copy)."""
        if isinstance(img, DeviceImage):
            return img
        if isinstance(img, torch.Tensor):
            t = img
        else:
            if hasattr(img, "convert"):                         # PIL.Image (ref:loader.py:163 .convert("RGB"))
                img = img.convert("RGB")
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(img)))
        device = device if device is not None else (t.device if t.is_cuda else torch.device("cuda"))
        return cls(t.to(device, non_blocking=True))

    @property
    def size(self) -> Tuple[int, int]:
        return self.canvas

    @property
    def plain(self) -> bool:
        return self.paste == (0, 0) and self.canvas == (self.data.shape[1], self.data.shape[0])

    def crop(self, box) -> "DeviceImage":
        """``Image.crop((left, upper, right, lower))``; PIL fills what lies outside the image with zeros."""
        l, u, r, b = (int(v) for v in box)
        W, H = self.canvas
        if l < 0 or u < 0 or r > W or b > H:
            if self.plain or _pack_rgb(self.bg) == 0:
                return DeviceImage(self.data, (r - l, b - u), (self.paste[0] - l, self.paste[1] - u), (0, 0, 0))
            raise NotImplementedError("crop beyond a coloured canvas")
        return DeviceImage(self.data, (r - l, b - u), (self.paste[0] - l, self.paste[1] - u), self.bg)

    def materialize(self) -> torch.Tensor:
        """The window as a dense u8 [H, W, 3] tensor (tests / debugging; the kernels never need it)."""
        if self.plain:
            return self.data
        W, H = self.canvas
        out = torch.empty((H, W, 3), dtype=torch.uint8, device=self.data.device)
        bgc = _pack_rgb(self.bg)
        out[:] = torch.tensor([bgc & 255, (bgc >> 8) & 255, (bgc >> 16) & 255], dtype=torch.uint8, device=out.device)
        h, w = self.data.shape[:2]
        px, py = self.paste
        x0, y0, x1, y1 = max(px, 0), max(py, 0), min(px + w, W), min(py + h, H)
        if x1 > x0 and y1 > y0:
            out[y0:y1, x0:x1] = self.data[y0 - py:y1 - py, x0 - px:x1 - px]
        return out

    def resize(self, size: Tuple[int, int], resample=None) -> "DeviceImage":
        """``Image.resize((w, h))`` with PIL's default filter for RGB (BICUBIC), two passes like
        ImagingResample: horizontal first, u8 between the passes."""
        if resample not in (None, 3):
            raise NotImplementedError("only BICUBIC (PIL's default, CLIP's resample=3)")
        w, h = int(size[0]), int(size[1])
        W, H = self.canvas
        if (w, h) == (W, H):
            return self
        if w <= 0 or h <= 0:
            raise ValueError("height and width must be > 0")
        dev = self.data.device
        cur = self
        if w != W:
            ksize, bounds, kk = _device_coeffs(W, w, dev)
            out = torch.empty((H, w, 3), dtype=torch.uint8, device=dev)
            d = self.data
            check(lib.zk_u8_resample_rows(d.data_ptr(), d.shape[0], d.shape[1], d.stride(0), H, self.paste[0],
                                          self.paste[1], _pack_rgb(self.bg), bounds.data_ptr(), kk.data_ptr(), ksize, w,
                                          out.data_ptr(), _stream()), "zk_u8_resample_rows")
            cur = DeviceImage(out)
        if h != H:
            src = cur.materialize()
            if src.stride(0) != src.shape[1] * 3:
                src = src.contiguous()
            ksize, bounds, kk = _device_coeffs(H, h, dev)
            out = torch.empty((h, src.shape[1], 3), dtype=torch.uint8, device=dev)
            check(lib.zk_u8_resample_cols(src.data_ptr(), H, src.shape[1] * 3, bounds.data_ptr(), kk.data_ptr(), ksize,
                                          h, out.data_ptr(), _stream()), "zk_u8_resample_cols")
            cur = DeviceImage(out)
        return cur


def normalize_lut(image_mean, image_std, rescale_factor: float = 1 / 255) -> np.ndarray:
    """[3, 256] float32: what tf:image_transforms.py rescale (float64 multiply, cast to float32) followed by
    normalize (float32 subtract, divide) give for every byte value."""
    x = (np.arange(256).astype(np.float64) * rescale_factor).astype(np.float32)
    mean = np.array(image_mean, dtype=np.float32)
    std = np.array(image_std, dtype=np.float32)
    return np.ascontiguousarray(((x[None, :] - mean[:, None]) / std[:, None]).astype(np.float32))


def patches_normalize(img: DeviceImage, lut: torch.Tensor, patch: int, grid: Tuple[int, int],
                      out: Optional[torch.Tensor] = None, dtype=torch.float32) -> torch.Tensor:
    """[gx * gy, 3, patch, patch] normalised tiles of the window (row-major, ref:loader.py:770-789)."""
    gx, gy = grid
    P = gx * gy
    d = img.data
    if out is None:
        out = torch.empty((P, 3, patch, patch), dtype=dtype, device=d.device)
    if out.shape != (P, 3, patch, patch) or not out.is_contiguous() or out.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("patches_normalize: bad output buffer")
    check(lib.zk_u8_patches_normalize(d.data_ptr(), d.shape[0], d.shape[1], d.stride(0), img.paste[0], img.paste[1],
                                      _pack_rgb(img.bg), gx, patch, P, lut.data_ptr(), out.data_ptr(),
                                      int(out.dtype == torch.bfloat16), _stream()), "zk_u8_patches_normalize")
    return out


class CLIPImageProcessorHip:
    """``CLIPImageProcessor`` (PIL backend) for DeviceImages: same attributes, same pipeline (resize to the
    shortest edge, centre crop, rescale, normalize; tf:image_processing_backends.py:619-653), same bits.
    ``preprocess(...)["pixel_values"]`` is a device tensor [n, 3, crop, crop] (float32 like the reference's
    CPU tensor, or ``dtype=torch.bfloat16`` to skip the engine's later cast, ref:engine:48)."""

    resample = 3
    do_resize = do_center_crop = do_rescale = do_normalize = do_convert_rgb = True

    def __init__(self, size: int = 336, image_mean=None, image_std=None, device=None):
        self.size = {"shortest_edge": size}
        self.crop_size = {"height": size, "width": size}
        self.image_mean = list(image_mean or OPENAI_CLIP_MEAN)
        self.image_std = list(image_std or OPENAI_CLIP_STD)
        self.rescale_factor = 1 / 255
        self.device = device
        self._luts: Dict[str, torch.Tensor] = {}

    def lut(self, device) -> torch.Tensor:
        t = self._luts.get(str(device))
        if t is None:
            t = self._luts[str(device)] = torch.from_numpy(
                normalize_lut(self.image_mean, self.image_std, self.rescale_factor)).to(device)
        return t

    def _resized_window(self, img: DeviceImage) -> DeviceImage:
        s, ch, cw = self.size["shortest_edge"], self.crop_size["height"], self.crop_size["width"]
        W, H = img.size
        short, long = (W, H) if W <= H else (H, W)
        new_short, new_long = s, int(s * long / short)          # tf get_resize_output_image_size
        nw, nh = (new_short, new_long) if W <= H else (new_long, new_short)
        r = img.resize((nw, nh))
        top, left = (nh - ch) // 2, (nw - cw) // 2              # tf center_crop
        if top < 0 or left < 0:
            raise NotImplementedError("centre crop larger than the resized image")
        return r.crop((left, top, left + cw, top + ch))

    def preprocess(self, images, return_tensors: str = "pt", dtype=torch.float32, **_):
        if return_tensors not in ("pt", None):
            raise NotImplementedError("device tensors only")
        single = not isinstance(images, (list, tuple))
        imgs = [DeviceImage.from_any(i, self.device) for i in ([images] if single else images)]
        ps = self.crop_size["height"]
        if self.crop_size["width"] != ps:
            raise NotImplementedError("square crops only (CLIP)")
        dev = imgs[0].data.device
        out = torch.empty((len(imgs), 3, ps, ps), dtype=dtype, device=dev)
        for i, img in enumerate(imgs):
            patches_normalize(self._resized_window(img), self.lut(dev), ps, (1, 1), out=out[i:i + 1])
        return {"pixel_values": out}

    __call__ = preprocess


# ---- ref:loader.py:609-789 --------------------------------------------------------------------------------

def expand2square(img, bg_value) -> DeviceImage:
    """ref:loader.py:609-620: centre on a square canvas of ``bg_value`` (no copy: the window is edited)."""
    img = DeviceImage.from_any(img)
    if not img.plain:
        img = DeviceImage(img.materialize())
    w, h = img.size
    if w == h:
        return img
    if w > h:
        return DeviceImage(img.data, (w, w), (0, (w - h) // 2), bg_value)
    return DeviceImage(img.data, (h, h), ((h - w) // 2, 0), bg_value)


def select_best_resolution(original_size, possible_resolutions):
    """ref:loader.py:691-728: the (width, height) with the largest effective resolution, least waste on ties."""
    ow, oh = original_size
    best, best_eff, best_waste = None, 0, float("inf")
    for w, h in possible_resolutions:
        scale = min(w / ow, h / oh)
        eff = min(int(ow * scale) * int(oh * scale), ow * oh)
        waste = w * h - eff
        if eff > best_eff or (eff == best_eff and waste < best_waste):
            best, best_eff, best_waste = (w, h), eff, waste
    return best


def resize_and_pad_image(image, target_resolution, bg_value=(0, 0, 0)) -> DeviceImage:
    """ref:loader.py:731-767: aspect-preserving resize, centred on a ``target_resolution`` canvas."""
    image = DeviceImage.from_any(image)
    ow, oh = image.size
    tw, th = target_resolution
    sw, sh = tw / ow, th / oh
    if sw < sh:
        nw, nh = tw, min(math.ceil(oh * sw), th)
    else:
        nh, nw = th, min(math.ceil(ow * sh), tw)
    r = image.resize((nw, nh))
    if not r.plain:
        r = DeviceImage(r.materialize())
    return DeviceImage(r.data, (tw, th), ((tw - nw) // 2, (th - nh) // 2), bg_value)


def divide_to_patches(image: DeviceImage, patch_size: int) -> List[DeviceImage]:
    """ref:loader.py:770-789: row-major tiles (windows onto the same buffer)."""
    w, h = image.size
    return [image.crop((j, i, j + patch_size, i + patch_size)) for i in range(0, h, patch_size)
            for j in range(0, w, patch_size)]


def parse_grid_pinpoints(grid_pinpoints, patch_size: int):
    """ref:loader.py:636-664."""
    if isinstance(grid_pinpoints, str) and "x" in grid_pinpoints:
        if patch_size not in (224, 336, 384, 448, 512):
            raise AssertionError("patch_size should be in [224, 336, 384, 448, 512]")
        m = re.findall(r"\((\d+)x(\d+)\)", grid_pinpoints)
        (a0, b0), (a1, b1) = tuple(map(int, m[0])), tuple(map(int, m[-1]))
        return [[i * patch_size, j * patch_size] for i in range(a0, a1 + 1) for j in range(b0, b1 + 1)]
    return grid_pinpoints if isinstance(grid_pinpoints, list) else ast.literal_eval(grid_pinpoints)


def process_anyres_image(image, processor: CLIPImageProcessorHip, grid_pinpoints, bg_value=None,
                         dtype=torch.float32) -> torch.Tensor:
    """ref:loader.py:623-688 -> [1 + tiles, 3, s, s]: the whole image squashed to s x s, then the tiles of the
    padded best-resolution canvas.  Two normalisation launches write one tensor (the reference runs the
    processor once per tile and concatenates)."""
    image = DeviceImage.from_any(image, processor.device)
    s = processor.size["shortest_edge"]
    ps = processor.crop_size["height"]
    res = parse_grid_pinpoints(grid_pinpoints, s)
    best = select_best_resolution(image.size, res)
    padded = resize_and_pad_image(image, best, bg_value)
    if ps != s or best[0] % ps or best[1] % ps:
        tiles = divide_to_patches(padded, ps)
        base = image.resize((s, s))
        return torch.cat([processor.preprocess(t, dtype=dtype)["pixel_values"] for t in [base] + tiles], dim=0)
    gx, gy = best[0] // ps, best[1] // ps
    dev = image.data.device
    out = torch.empty((1 + gx * gy, 3, ps, ps), dtype=dtype, device=dev)
    base = image.resize((s, s))                                   # ref:loader.py:677 (resize, not pad)
    patches_normalize(base, processor.lut(dev), ps, (1, 1), out=out[0:1])
    patches_normalize(padded, processor.lut(dev), ps, (gx, gy), out=out[1:])
    return out


def load_image(image, processor: CLIPImageProcessorHip, image_ratio_type: str = "pad", anyres_grid_pinpoints=None,
               bg_value=None, dtype=torch.float32) -> torch.Tensor:
    """The image branch of ``JsonDataset.__getitem__`` (ref:loader.py:171-197) -> [patches, 3, h, w]."""
    if bg_value is None:
        bg_value = tuple(int(x * 255) for x in processor.image_mean)          # ref:loader.py:108
    if image_ratio_type == "pad":
        return processor.preprocess(expand2square(DeviceImage.from_any(image, processor.device), bg_value),
                                    dtype=dtype)["pixel_values"]
    if image_ratio_type == "anyres":
        return process_anyres_image(image, processor, anyres_grid_pinpoints, bg_value, dtype=dtype)
    raise ValueError(f"unknown image_ratio_type: {image_ratio_type}")
