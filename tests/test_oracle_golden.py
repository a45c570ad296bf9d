"""The oracle (oracle/zero_oracle.py) against the golden vectors that the real
transformers modules + the reference's Zero/reduce_sum produced
(tests/golden/make_golden.py).  CPU only."""
import os

import pytest
import torch
from safetensors.torch import load_file

from oracle import zero_oracle as zo
from tests.golden.configs import GOLDEN_CONFIGS, shrink


def _close(a, b, rtol, atol, what):
    a = a.float(); b = b.float()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, f"{what}: max err {err:.3e} vs ref max {ref:.3e}"


@pytest.mark.parametrize("name", [n for n, s_ in GOLDEN_CONFIGS.items() if not s_.get("graft")])   # graft: tests/test_graft_cpu.py
def test_oracle_fp32_matches_golden(name, golden_dir):
    spec = GOLDEN_CONFIGS[name]
    gold = load_file(os.path.join(golden_dir, f"{name}.safetensors"))
    enc, dec = spec["enc"], spec["dec"]
    full = zo.make_params(enc, dec, spec["seed"], surrogate=False)
    chk = torch.stack([v.double().sum() for _, v in sorted(full.items())]).sum()
    assert abs(chk.item() - gold["weights_checksum"].item()) < 1e-6 * max(1.0, abs(chk.item())), "RNG drift"
    p = zo.apply_surgery(full, dec)
    imgs, ids, tgts, img_id, pad_id = zo.synth_batch(enc, dec, **spec["batch"])
    out = zo.train_step(p, enc, dec, imgs, ids, tgts, img_id=img_id, pad_id=pad_id,
                        max_seq_len=spec["max_seq_len"], encoder_ft_layers_after=spec["ft_after"],
                        keep_hidden=True, stage=spec.get("stage", "finetune-encoder"))
    assert int(out.num_tokens) == int(gold["fp32.num_tokens"])
    assert torch.equal(out.target, gold["fp32.target"])
    _close(out.loss_sum, gold["fp32.loss_sum"][0], 1e-5, 1e-5, "loss_sum")
    _close(out.loss, gold["fp32.loss"][0], 1e-5, 1e-6, "loss")
    _close(shrink(torch.stack(out.embds_imgs)), gold["fp32.embds_imgs"], 1e-4, 1e-5, "embds_imgs")
    _close(shrink(out.inputs_embeds), gold["fp32.inputs_embeds"], 1e-4, 1e-5, "inputs_embeds")
    _close(shrink(out.logits), gold["fp32.logits"], 1e-4, 1e-5, "logits")
    # HF's last hidden_states entry is taken after the final norm; compare the pre-norm ones
    for i, h in enumerate(out.dec_hidden[:-1]):
        _close(shrink(h), gold[f"fp32.dec_hidden.{i}"], 1e-4, 1e-5, f"dec_hidden.{i}")
    assert f"fp32.dec_hidden.{len(out.dec_hidden) - 1}" in gold
    n_checked = 0
    for k, v in gold.items():
        if k.startswith("fp32.grad."):
            n = k[len("fp32.grad."):]
            _close(shrink(out.grads[n]), v, 2e-4, 1e-7, k)
            _close(out.grads[n].norm(), gold["fp32.gradnorm." + n][0], 2e-4, 1e-8, "norm " + n)
            n_checked += 1
    assert n_checked == len(out.grads) and n_checked > 6
    if spec.get("stage") == "pretrain":   # translator layers 1 and 3 of the 5-layer surrogate, no encoder layer
        assert any(k.startswith("decoder.model.layers.1.") for k in out.grads)
        assert any(k.startswith("decoder.model.layers.3.mlp.down_proj") for k in out.grads)
        assert not any(".encoder.layers." in k for k in out.grads)
    # the eager attention path of transformers agrees with sdpa, and so with the oracle
    _close(shrink(out.logits), gold["eager.logits"], 1e-4, 1e-5, "eager logits")
    # encoder hidden states (output_hidden_states=True)
    hs = zo.clip_vision_forward(p, torch.cat(imgs, 0), enc)
    for i, h in enumerate(hs):
        _close(shrink(h), gold[f"fp32.enc_hidden.{i}"], 1e-4, 1e-5, f"enc_hidden.{i}")


@pytest.mark.parametrize("name", ["tiny", "trunc"])
def test_oracle_bf16_close_to_golden_bf16(name, golden_dir):
    """bf16 mode of the oracle vs the bf16 run of transformers: same cast points, so
    agreement is far tighter than bf16 resolution would demand."""
    spec = GOLDEN_CONFIGS[name]
    gold = load_file(os.path.join(golden_dir, f"{name}.safetensors"))
    enc, dec = spec["enc"], spec["dec"]
    p = zo.make_params(enc, dec, spec["seed"], surrogate=True, dtype=torch.bfloat16)
    imgs, ids, tgts, img_id, pad_id = zo.synth_batch(enc, dec, **spec["batch"])
    out = zo.train_step(p, enc, dec, [i.to(torch.bfloat16) for i in imgs], ids, tgts, img_id=img_id,
                        pad_id=pad_id, max_seq_len=spec["max_seq_len"],
                        encoder_ft_layers_after=spec["ft_after"])
    _close(out.loss, gold["bf16.loss"][0], 2e-3, 1e-3, "bf16 loss")
    _close(shrink(out.logits), gold["bf16.logits"], 2e-2, 1e-3, "bf16 logits")


def test_unit_vectors(golden_dir):
    u = load_file(os.path.join(golden_dir, "unit.safetensors"))
    for tag, dec in (("3b", zo.LLAMA32_3B), ("8b", zo.LLAMA31_8B)):
        inv = zo.llama3_inv_freq(dec)
        assert torch.equal(inv, u[f"inv_freq.{tag}"]), tag
        cos, sin = zo.rope_cos_sin(dec, 4096, torch.bfloat16)
        pos = u[f"rope_pos.{tag}"]
        assert torch.equal(cos[pos].float(), u[f"rope_cos_bf16.{tag}"])
        assert torch.equal(sin[pos].float(), u[f"rope_sin_bf16.{tag}"])
    cos, sin = zo.rope_cos_sin(zo.LLAMA32_3B, 7, torch.float32)
    q, k = u["rope.q"], u["rope.k"]
    qe = q * cos[None, None] + zo.rotate_half(q) * sin[None, None]
    ke = k * cos[None, None] + zo.rotate_half(k) * sin[None, None]
    _close(qe, u["rope.qe"], 1e-6, 1e-6, "rope q")
    _close(ke, u["rope.ke"], 1e-6, 1e-6, "rope k")
    _close(zo.rms_norm(u["rms.x"], u["rms.w"], 1e-5), u["rms.y"], 1e-6, 1e-6, "rms")
    yb = zo.rms_norm(u["rms.x"].bfloat16(), u["rms.w"].bfloat16(), 1e-5)
    assert torch.equal(yb.float(), u["rms.y_bf16"])
    _close(zo.quick_gelu(u["act.x"]), u["act.quick_gelu"], 1e-6, 1e-7, "quick_gelu")
    _close(torch.nn.functional.gelu(u["act.x"]), u["act.gelu"], 1e-6, 1e-7, "gelu")


def test_surgery_layer_counts():
    # 28->18, 32->19, 80->42
    assert zo.LLAMA32_3B.surrogate_layers == 18
    assert zo.LLAMA31_8B.surrogate_layers == 19
    assert len(zo.surrogate_layer_map(80, (tuple(range(40, 79)),), 1)) == 42
    assert zo.surrogate_layer_map(28, (tuple(range(16, 27)),), 1) == list(range(17)) + [27]


def test_trainable_set_3b():
    shapes = {}
    shapes.update(zo.encoder_param_shapes(zo.CLIP_L_336))
    shapes.update(zo.adapter_param_shapes(3072))
    names = zo.trainable_names({k: None for k in shapes}, 16)
    n = 0
    for k in names:
        c = 1
        for s in shapes[k]:
            c *= s
        n += c
    # 8 x 12,596,224 + 2,048 + 12.59 M + 2 d
    assert n == 8 * 12596224 + 2048 + (1024 * 3072 + 3072 + 3072 * 3072 + 3072) + 2 * 3072
