"""Image data path, CPU side: the oracle against the golden vectors made by the real
PIL + transformers CLIPImageProcessor (tests/golden/make_golden_image.py), and the product's host logic
(coefficient tables, resolution choice, normalisation table) against the oracle.  Everything bit-exact."""
import os

import numpy as np
import pytest
from safetensors import safe_open

from oracle import image_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "image.safetensors")
BG = tuple(int(v * 255) for v in O.CLIP_MEAN)


def golden_cases():
    with safe_open(GOLDEN, "np") as f:
        return [c.split(":") for c in f.metadata()["cases"].split(";")]


def run_oracle(case, x):
    mode = case[1]
    if mode == "pad":
        return O.pad_mode_pixels(x, int(case[2]))
    if mode == "crop":
        return O.clip_preprocess(x, int(case[2]), int(case[2]))
    if mode == "anyres":
        return O.process_anyres_image(x, int(case[2]), case[3], BG)
    w, h = map(int, case[2].split("x"))
    return O.pil_resize(x, (w, h))


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c[0])
def test_oracle_matches_pil_and_transformers(case):
    with safe_open(GOLDEN, "np") as f:
        x, ref = f.get_tensor(case[0] + "_in"), f.get_tensor(case[0] + "_out")
        got = run_oracle(case, x)
        assert got.shape == ref.shape and got.dtype == ref.dtype
        assert np.array_equal(got, ref)
        if case[1] == "anyres":
            h, w, _ = x.shape
            best = O.select_best_resolution((w, h), O.parse_grid_pinpoints(case[3], int(case[2])))
            assert tuple(f.get_tensor(case[0] + "_best")) == best


def test_product_coefficient_tables_equal_oracle():
    from zero_amd.loader import resample_coeffs
    pairs = [(83, 28), (90, 28), (20, 28), (500, 336), (375, 336), (28, 28), (97, 30), (17, 40), (25, 11), (40, 90),
             (1, 5), (5, 1), (2, 3), (1333, 336), (4000, 336), (336, 672), (337, 336), (335, 336)]
    rs = np.random.RandomState(0)
    pairs += [(int(a), int(b)) for a, b in rs.randint(1, 700, size=(60, 2))]
    for n_in, n_out in pairs:
        ksize, bounds, kk = O.precompute_coeffs(n_in, 0.0, float(n_in), n_out)
        ik = np.array(O.normalize_coeffs_8bpc(kk), dtype=np.int32)
        pk, pb, pkk = resample_coeffs(n_in, n_out)
        assert pk == ksize, (n_in, n_out)
        assert np.array_equal(pb, np.array(bounds, dtype=np.int32)), (n_in, n_out)
        assert np.array_equal(pkk, ik), (n_in, n_out)
        assert (pb[:, 0] >= 0).all() and (pb[:, 0] + pb[:, 1] <= n_in).all() and (pb[:, 1] <= ksize).all()


def test_product_resolution_choice_equals_oracle():
    from zero_amd import loader as L
    res = O.parse_grid_pinpoints("(1x1),...,(6x6)", 336)
    assert L.parse_grid_pinpoints("(1x1),...,(6x6)", 336) == [list(r) for r in res]
    rs = np.random.RandomState(1)
    for w, h in rs.randint(8, 3000, size=(400, 2)):
        assert L.select_best_resolution((int(w), int(h)), res) == O.select_best_resolution((int(w), int(h)), res)
    with pytest.raises(AssertionError):                        # ref:loader.py:641-647
        L.parse_grid_pinpoints("(1x1),...,(2x2)", 28)
    assert L.parse_grid_pinpoints("[(336, 672), (672, 336)]", 336) == [(336, 672), (672, 336)]


def test_product_normalize_table_equals_oracle():
    from zero_amd.loader import normalize_lut
    lut = normalize_lut(O.CLIP_MEAN, O.CLIP_STD)
    img = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)          # [1, 256, 3]
    assert np.array_equal(lut, O.normalize_u8(img)[:, 0, :])


def test_device_image_refuses_cpu_tensors():
    import torch
    from zero_amd.loader import DeviceImage
    with pytest.raises(ValueError):
        DeviceImage(torch.zeros(4, 4, 3, dtype=torch.uint8))
