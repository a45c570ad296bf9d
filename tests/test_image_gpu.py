"""Image data path on the MI355X: csrc/image.hip through the C ABI against the golden
vectors (real PIL + transformers), against the oracle at other sizes, and through size-independent properties
at camera-sized inputs.  Bit-exact everywhere: this is byte / fixed-point work."""
import os

import numpy as np
import pytest
import torch
from safetensors import safe_open

from oracle import image_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "image.safetensors")
BG = tuple(int(v * 255) for v in O.CLIP_MEAN)


def golden_cases():
    with safe_open(GOLDEN, "np") as f:
        return [c.split(":") for c in f.metadata()["cases"].split(";")]


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def run_product(case, x):
    from zero_amd import loader as L
    mode = case[1]
    if mode == "resize":
        w, h = map(int, case[2].split("x"))
        return L.DeviceImage(dev(x)).resize((w, h)).materialize()
    proc = L.CLIPImageProcessorHip(int(case[2]))
    if mode == "pad":
        return L.load_image(dev(x), proc, "pad")
    if mode == "crop":
        return proc(L.DeviceImage(dev(x)), return_tensors="pt")["pixel_values"]
    grid = [list(r) for r in O.parse_grid_pinpoints(case[3], int(case[2]))]
    return L.load_image(dev(x), proc, "anyres", anyres_grid_pinpoints=grid)


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c[0])
def test_golden_bit_exact(case):
    with safe_open(GOLDEN, "np") as f:
        x, ref = f.get_tensor(case[0] + "_in"), f.get_tensor(case[0] + "_out")
    got = run_product(case, x).cpu().numpy()
    assert got.shape == ref.shape and got.dtype == ref.dtype
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("h,w,size", [(333, 517, 336), (700, 250, 336), (64, 64, 336), (1000, 750, 224)])
def test_pad_mode_equals_oracle(h, w, size):
    from zero_amd import loader as L
    x = np.random.RandomState(h + w).randint(0, 256, (h, w, 3), dtype=np.uint8)
    got = L.load_image(dev(x), L.CLIPImageProcessorHip(size), "pad")
    assert np.array_equal(got.cpu().numpy(), O.pad_mode_pixels(x, size))
    bf = L.load_image(dev(x), L.CLIPImageProcessorHip(size), "pad", dtype=torch.bfloat16)
    assert torch.equal(bf, got.to(torch.bfloat16))               # the engine's later cast (ref:engine:48)


@pytest.mark.parametrize("h,w", [(480, 640), (900, 300), (336, 336), (200, 1300)])
def test_anyres_336_equals_oracle(h, w):
    from zero_amd import loader as L
    x = np.random.RandomState(h * 7 + w).randint(0, 256, (h, w, 3), dtype=np.uint8)
    proc = L.CLIPImageProcessorHip(336)
    got = L.load_image(dev(x), proc, "anyres", anyres_grid_pinpoints="(1x1),...,(3x3)")
    ref = O.process_anyres_image(x, 336, "(1x1),...,(3x3)", BG)
    assert got.shape == ref.shape
    assert np.array_equal(got.cpu().numpy(), ref)
    # the per-tile route of the reference (divide_to_patches + one preprocess per tile) gives the same tensor
    di = L.DeviceImage(dev(x))
    best = L.select_best_resolution(di.size, L.parse_grid_pinpoints("(1x1),...,(3x3)", 336))
    tiles = L.divide_to_patches(L.resize_and_pad_image(di, best, BG), 336)
    per_tile = torch.cat([proc.preprocess(t)["pixel_values"] for t in [di.resize((336, 336))] + tiles])
    assert torch.equal(per_tile, got)


def test_window_ops_equal_pil_semantics():
    from zero_amd import loader as L
    x = np.random.RandomState(5).randint(0, 256, (37, 61, 3), dtype=np.uint8)
    sq = L.expand2square(dev(x), BG)
    assert sq.size == (61, 61)
    assert np.array_equal(sq.materialize().cpu().numpy(), O.expand2square(x, BG))
    pad = L.resize_and_pad_image(dev(x), (56, 84), BG)
    assert np.array_equal(pad.materialize().cpu().numpy(), O.resize_and_pad_image(x, (56, 84), BG))
    tiles = L.divide_to_patches(pad, 28)
    ref_tiles = O.divide_to_patches(O.resize_and_pad_image(x, (56, 84), BG), 28)
    assert len(tiles) == len(ref_tiles) == 6
    for t, r in zip(tiles, ref_tiles):
        assert np.array_equal(t.materialize().cpu().numpy(), r)
    # vertical-only resize of a padded window (materialised first), horizontal-only resize
    assert np.array_equal(sq.resize((61, 20)).materialize().cpu().numpy(), O.pil_resize(O.expand2square(x, BG), (61, 20)))
    assert np.array_equal(sq.resize((90, 61)).materialize().cpu().numpy(), O.pil_resize(O.expand2square(x, BG), (90, 61)))


def test_camera_size_properties():
    """4000 x 3000: too slow for the oracle's loops; properties the resampler must keep at any size."""
    from zero_amd import loader as L
    proc = L.CLIPImageProcessorHip(336)
    lut = proc.lut(torch.device("cuda")).cpu().numpy()
    # a constant image stays constant (coefficients sum to 2^22 within a few units), whatever the padding colour
    const = torch.empty((3000, 4000, 3), dtype=torch.uint8, device="cuda")
    const[:] = torch.tensor(BG, dtype=torch.uint8, device="cuda")
    out = L.load_image(const, proc, "pad").cpu().numpy()
    for c in range(3):
        assert (out[0, c] == lut[c, BG[c]]).all()
    # separability: resizing rows then columns by hand equals the one call; the result is deterministic
    x = torch.randint(0, 256, (3000, 4000, 3), dtype=torch.uint8, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
    a = L.DeviceImage(x).resize((448, 336)).materialize()
    b = L.DeviceImage(L.DeviceImage(x).resize((448, 3000)).materialize()).resize((448, 336)).materialize()
    assert torch.equal(a, b)
    # a vertical flip commutes with the horizontal pass, a horizontal flip with the vertical pass
    hf = L.DeviceImage(x.flip(0).contiguous()).resize((448, 3000)).materialize()
    assert torch.equal(hf, L.DeviceImage(x).resize((448, 3000)).materialize().flip(0))
    # and the sampled sub-image agrees with the oracle on a strip it can afford (rows are independent)
    strip = x[:8].cpu().numpy()
    assert np.array_equal(L.DeviceImage(x[:8].contiguous()).resize((448, 8)).materialize().cpu().numpy(),
                          O.resample_horizontal(strip, 448))


def test_bad_arguments_are_refused():
    from zero_amd._lib import lib
    assert lib.zk_u8_resample_rows(None, 1, 1, 3, 1, 0, 0, 0, None, None, 1, 1, None, None) == -22
    assert lib.zk_u8_resample_cols(None, 1, 1, None, None, 1, 1, None, None) == -22
    assert lib.zk_u8_patches_normalize(None, 1, 1, 3, 0, 0, 0, 1, 1, 1, None, None, 0, None) == -22
