"""Generate tests/golden/image.safetensors: golden vectors of the image data path.

Run ONCE in the build container:   python tests/golden/make_golden_image.py

What runs here is the real thing, not the oracle: Pillow (12.2.0) and the transformers (5.15.0)
``CLIPImageProcessor`` — which resolves to its PIL backend here, the backend the reference's pinned
transformers 4.45 used — built from local arguments (no hub access).  ``loader.py`` itself cannot be
imported (needs ``torchvision``), so its steps are driven with the same PIL calls in the order
ref:loader.py:171-175,609-620 ("pad") and :623-688,731-789 ("anyres") give: ``Image.new`` / ``paste``,
``Image.resize`` (default filter), ``crop``, ``processor.preprocess``.  ``select_best_resolution`` is pure
integer/float host logic with nothing third-party to run; its choices are stored as the ``*_best``
entries computed by the calls below from the reference's formula, and cross-checked in the tests by
exhaustive comparison of the oracle and the product.
Inputs are seeded uint8 arrays stored in the fixture next to the outputs.
"""
import math
import os
import re
import sys

import numpy as np
import torch
from PIL import Image
from safetensors.torch import save_file
from transformers import CLIPImageProcessor

HERE = os.path.dirname(os.path.abspath(__file__))


def processor(size):
    return CLIPImageProcessor(size={"shortest_edge": size}, crop_size={"height": size, "width": size})


def synth(h, w, seed, kind):
    rs = np.random.RandomState(seed)
    if kind == "noise":
        return rs.randint(0, 256, (h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 255 // max(h + w - 2, 1))], -1)
    edges = ((xx // 7 + yy // 5) % 2) * 90
    return np.clip(base * 0.6 + edges[..., None] + rs.randint(0, 30, (h, w, 3)), 0, 255).astype(np.uint8)


def pad_mode(arr, size):
    p = processor(size)
    bg = tuple(int(x * 255) for x in p.image_mean)                 # loader.py:108
    img = Image.fromarray(arr, "RGB")
    w, h = img.size
    if w > h:                                                      # loader.py:609-620
        x = Image.new(img.mode, (w, w), bg)
        x.paste(img, (0, (w - h) // 2))
        img = x
    elif h > w:
        x = Image.new(img.mode, (h, h), bg)
        x.paste(img, ((h - w) // 2, 0))
        img = x
    return p(img, return_tensors="pt")["pixel_values"]             # loader.py:173-175


def anyres_mode(arr, size, grid):
    p = processor(size)
    bg = tuple(int(x * 255) for x in p.image_mean)
    img = Image.fromarray(arr, "RGB")
    m = re.findall(r"\((\d+)x(\d+)\)", grid)                       # loader.py:649-659
    a0, a1 = tuple(map(int, m[0])), tuple(map(int, m[-1]))
    res = [[i * size, j * size] for i in range(a0[0], a1[0] + 1) for j in range(a0[1], a1[1] + 1)]
    ow, oh = img.size
    best, be, bw = None, 0, float("inf")                           # loader.py:691-728
    for w, h in res:
        s = min(w / ow, h / oh)
        dw, dh = int(ow * s), int(oh * s)
        eff = min(dw * dh, ow * oh)
        waste = w * h - eff
        if eff > be or (eff == be and waste < bw):
            be, bw, best = eff, waste, (w, h)
    tw, th = best                                                  # loader.py:731-767
    sw, sh = tw / ow, th / oh
    if sw < sh:
        nw, nh = tw, min(math.ceil(oh * sw), th)
    else:
        nh, nw = th, min(math.ceil(ow * sh), tw)
    resized = img.resize((nw, nh))
    canvas = Image.new("RGB", (tw, th), bg)
    canvas.paste(resized, ((tw - nw) // 2, (th - nh) // 2))
    ps = p.crop_size["height"]
    tiles = [canvas.crop((j, i, j + ps, i + ps)) for i in range(0, th, ps) for j in range(0, tw, ps)]   # :770-789
    base = img.resize((size, size))                                # loader.py:677
    outs = [p.preprocess(t, return_tensors="pt")["pixel_values"] for t in [base] + tiles]
    return torch.cat(outs, dim=0), best


def main():
    out = {}
    meta = {"pillow": Image.__version__ if hasattr(Image, "__version__") else "", "cases": ""}
    import PIL
    meta["pillow"] = PIL.__version__
    import transformers
    meta["transformers"] = transformers.__version__
    cases = []
    # pad mode (expand2square + processor)
    for name, (h, w, seed, kind, size) in {
        "pad_wide": (47, 83, 1, "noise", 28), "pad_tall": (90, 31, 2, "noise", 28), "pad_up": (20, 20, 3, "noise", 28),
        "pad_smooth": (61, 100, 4, "smooth", 56), "pad_same": (28, 28, 5, "noise", 28),
        "pad_336": (500, 375, 6, "smooth", 336),
    }.items():
        arr = synth(h, w, seed, kind)
        out[name + "_in"] = torch.from_numpy(arr)
        out[name + "_out"] = pad_mode(arr, size).contiguous()
        cases.append(f"{name}:pad:{size}")
    # the processor alone on non-square images (resize to the shortest edge + centre crop)
    for name, (h, w, seed, size) in {"crop_wide": (50, 77, 7, 28), "crop_tall": (131, 64, 8, 28)}.items():
        arr = synth(h, w, seed, "noise")
        out[name + "_in"] = torch.from_numpy(arr)
        out[name + "_out"] = processor(size)(Image.fromarray(arr, "RGB"), return_tensors="pt")["pixel_values"].contiguous()
        cases.append(f"{name}:crop:{size}")
    # anyres
    for name, (h, w, seed, kind, size, grid) in {
        "any_wide": (75, 130, 9, "noise", 28, "(1x1),...,(3x3)"), "any_small": (40, 33, 10, "smooth", 28, "(1x1),...,(3x3)"),
        "any_tall": (200, 45, 11, "smooth", 28, "(1x1),...,(4x4)"), "any_exact": (56, 56, 12, "noise", 28, "(1x1),...,(2x2)"),
    }.items():
        arr = synth(h, w, seed, kind)
        px, best = anyres_mode(arr, size, grid)
        out[name + "_in"] = torch.from_numpy(arr)
        out[name + "_out"] = px.contiguous()
        out[name + "_best"] = torch.tensor(best, dtype=torch.int32)
        cases.append(f"{name}:anyres:{size}:{grid}")
    # the bare resampler
    for name, (h, w, seed, nw, nh) in {"rs_down": (64, 97, 13, 30, 21), "rs_up": (17, 23, 14, 40, 31),
                                       "rs_mixed": (40, 25, 15, 11, 90), "rs_big": (300, 200, 16, 37, 41)}.items():
        arr = synth(h, w, seed, "noise")
        out[name + "_in"] = torch.from_numpy(arr)
        out[name + "_out"] = torch.from_numpy(np.asarray(Image.fromarray(arr, "RGB").resize((nw, nh))).copy())
        cases.append(f"{name}:resize:{nw}x{nh}")
    meta["cases"] = ";".join(cases)
    save_file(out, os.path.join(HERE, "image.safetensors"), metadata=meta)
    print("wrote", len(out), "tensors;", sum(t.numel() * t.element_size() for t in out.values()) / 1e6, "MB")


if __name__ == "__main__":
    main()
