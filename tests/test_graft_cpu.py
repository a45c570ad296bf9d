"""Zero-shot graft inference, CPU side: the oracle's greedy decoding (full recompute per step) against
the golden run of transformers' own generate() with a KV cache on the real modules + the reference's
Zero (tests/golden/make_golden.py run_graft); tensor-parallel sharding arithmetic on CPU tensors."""
import os

import torch
from safetensors.torch import load_file

from oracle import zero_oracle as zo
from tests.golden.configs import GOLDEN_CONFIGS, shrink


def test_oracle_greedy_matches_transformers_generate(golden_dir):
    spec = GOLDEN_CONFIGS["graft"]
    gold = load_file(os.path.join(golden_dir, "graft.safetensors"))
    enc, dec = spec["enc"], spec["dec"]
    p = zo.make_params(enc, dec, spec["seed"], surrogate=False)
    assert zo.count_decoder_layers(p) == dec.layers                   # graft = the FULL decoder (vlm_infer.py:57)
    imgs, ids, tgts, img_id, _ = zo.synth_batch(enc, dec, **spec["batch"])
    embds = zo.zero_forward(p, imgs, enc)
    x, _ = zo.assemble_inputs(p, embds, ids, tgts, img_id, spec["max_seq_len"])
    assert (shrink(x) - gold["fp32.inputs_embeds"]).abs().max() <= 1e-5
    toks, logits = zo.greedy_generate(p, x, dec, spec["max_new_tokens"], return_logits=True)
    assert torch.equal(toks, gold["fp32.gen_tokens"])
    ref = gold["fp32.prompt_last_logits"]
    assert (logits[0] - ref).abs().max() <= 1e-4 * ref.abs().max() + 1e-5
    # EOS semantics of generate(): a finished sequence emits pad, the loop ends when all are finished
    eos = int(toks[0, 2])
    t2 = zo.greedy_generate(p, x, dec, spec["max_new_tokens"], eos_ids=(eos,), pad_id=7)
    first = (toks[0] == eos).nonzero()[0, 0].item()
    assert t2[0, first] == eos and (t2[0, first + 1:] == 7).all()


def test_tp_shard_shapes_and_partial_sums():
    """Column / row splits reproduce the unsharded products (fp32, CPU tensors only)."""
    torch.manual_seed(0)
    d, nh, nkv, hd, f, W = 64, 4, 2, 16, 128, 2
    x = torch.randn(5, d)
    wq, wo = torch.randn(nh * hd, d), torch.randn(d, nh * hd)
    wg, wu, wd = torch.randn(f, d), torch.randn(f, d), torch.randn(d, f)
    full_attn_like = (x @ wq.t()) @ wo.t()
    full_mlp = (torch.nn.functional.silu(x @ wg.t()) * (x @ wu.t())) @ wd.t()
    pa = pm = 0
    for r in range(W):
        q0, q1 = r * (nh // W) * hd, (r + 1) * (nh // W) * hd
        f0, f1 = r * (f // W), (r + 1) * (f // W)
        pa = pa + (x @ wq[q0:q1].t()) @ wo[:, q0:q1].t()
        pm = pm + (torch.nn.functional.silu(x @ wg[f0:f1].t()) * (x @ wu[f0:f1].t())) @ wd[:, f0:f1].t()
    assert torch.allclose(pa, full_attn_like, rtol=1e-4, atol=1e-3)
    assert torch.allclose(pm, full_mlp, rtol=1e-4, atol=1e-3)
