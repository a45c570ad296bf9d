"""Parity at BASELINE.json's FULL size (configs[2]: CLIP-L/14-336 encoder + adapter + the 18-layer
Llama-3.2-3B surrogate, S = 768, V = 128256), where the fp32 oracle cannot run a whole batch in seconds:

  * size-independent properties of the step —
      - batch additivity (the data-parallel identity the DDP step rests on):
        loss_sum and token count of a batch are the sums over its halves, and the batch gradient is the
        token-weighted mean of the halves' gradients;
      - run-to-run reproducibility: same inputs -> bit-identical loss and gradients, vector gradients included
        (no floating-point atomics anywhere: ordered partial sums);
  * the oracle itself on ONE full-size image with the same weights (loss and the adapter / last encoder
    layer gradients; loss within 1e-2, gradients within 5e-2 per tensor at this depth, see the test).

Weights: seeded random init built on the device exactly as bench.py does (no checkpoint exists offline).
"""
import copy
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def full():
    from oracle import zero_oracle as zo
    from zero_amd.build_decoder import load_hf_llama3
    from zero_amd.build_encoder import load_clip
    from zero_amd.configs.zero import train_llama as run_cfg
    from zero_amd.models.adapter import Zero
    cfg = copy.deepcopy(run_cfg.config)
    cfg.device = torch.device(DEV)
    cfg.batch_size = 4
    cfg.reduce_batch_loss = False
    torch.manual_seed(1337)
    stdout, sys.stdout = sys.stdout, open(os.devnull, "w")
    try:
        encoder, _, img_size = load_clip(cfg)
        num_embd, decoder, _, _, _ = load_hf_llama3(cfg)
    finally:
        sys.stdout = stdout
    cfg.num_embd = num_embd
    model = Zero(cfg, encoder, decoder)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "norm" in n and n.endswith("weight"):
                p.fill_(1.0)
    run_cfg.func_optim_filter(cfg, model)
    model.train()
    ecfg, dcfg = zo.EncoderCfg(), zo.DecoderCfg(vocab=128256)
    imgs, ids, tgts, img_id, pad_id = zo.synth_batch(ecfg, dcfg, 4, 4242, n_pre=21, n_post=168, n_resp=128)
    assert 21 + ((img_size // 14) ** 2 + 1 + 2) + 168 == 768
    return dict(cfg=cfg, model=model, imgs=imgs, ids=ids, tgts=tgts, img_id=img_id, pad_id=pad_id, ecfg=ecfg, zo=zo)


def _step(full, sel):
    """forward + backward of the samples `sel`; returns (loss_sum, n_tokens, {name: grad})."""
    from zero_amd.methods.engine_ar_instruction_llama import forward_from_ids
    model, cfg = full["model"], full["cfg"]
    model.zero_grad(set_to_none=True)
    dimgs = [full["imgs"][i].to(DEV).to(torch.bfloat16) for i in sel]
    loss, aux = forward_from_ids(model, cfg, dimgs, [full["ids"][i] for i in sel], [full["tgts"][i] for i in sel],
                                 full["img_id"], full["pad_id"], return_aux=True)
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    return float(aux["loss_sum"].float()), int(aux["num_tokens"]), float(loss.detach()), grads


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def test_full_size_reproducible(full):
    ls1, n1, l1, g1 = _step(full, [0, 1, 2, 3])
    ls2, n2, l2, g2 = _step(full, [0, 1, 2, 3])
    assert n1 == n2 == 4 * 128
    # no floating-point atomics anywhere on the path (round 2): the loss is an ORDERED sum of the row losses, weight
    # gradients are GEMMs with split-K slabs summed in order, vector gradients (biases, LayerNorm affine, soi / eoi)
    # are per-workgroup partial sums added in index order -> everything is bit-identical run to run
    assert ls1 == ls2 and l1 == l2
    assert len(g1) > 100
    for n in g1:
        assert torch.equal(g1[n], g2[n]), n


def test_full_size_batch_additivity(full):
    ls, n, loss, g = _step(full, [0, 1, 2, 3])
    ls_a, n_a, loss_a, g_a = _step(full, [0, 1])
    ls_b, n_b, loss_b, g_b = _step(full, [2, 3])
    assert n == n_a + n_b
    assert abs(ls - (ls_a + ls_b)) <= 2e-4 * abs(ls), (ls, ls_a, ls_b)        # fp32 sums of 512 token losses
    assert abs(loss - ls / n) <= 1e-5 * abs(loss)
    worst = ("", 0.0)
    scale = max(v.float().norm().item() / v.numel() ** 0.5 for v in g.values())
    for name, gv in g.items():
        mix = (g_a[name].float() * n_a + g_b[name].float() * n_b) / n
        err = ((gv.float() - mix).norm() / max(mix.norm().item(), 1e-3 * scale * gv.numel() ** 0.5)).item()
        if err > worst[1]:
            worst = (name, err)
    print("batch additivity: worst gradient", worst)
    # Not a bit-exact identity: M = 3072 and M = 1536 rows take different GEMM tilings / split-K plans, so the
    # two sides are two bf16 runs with different rounding through 18 + 8 layers.  Same bound as the parity of
    # one bf16 run against fp32 (tests/test_model_gpu.py): 3e-2 per tensor; measured worst 1.4e-2 (embd_eoi,
    # a sum of 4 single rows).
    assert worst[1] <= 3e-2, worst


def test_full_size_one_image_against_oracle(full):
    zo, model, cfg = full["zo"], full["model"], full["cfg"]
    ls, n, loss, g = _step(full, [0])
    torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16"))))
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    dc = model.decoder.config
    if "decoder.lm_head.weight" in sd and dc.tie_word_embeddings:
        del sd["decoder.lm_head.weight"]
    odec = zo.DecoderCfg(hidden=dc.hidden_size, heads=dc.num_attention_heads, kv_heads=dc.num_key_value_heads,
                         head_dim=dc.head_dim, intermediate=dc.intermediate_size, layers=dc.num_hidden_layers,
                         vocab=dc.vocab_size, rope_factor=dc.rope_scaling["factor"], tie=dc.tie_word_embeddings)
    ref = zo.train_step(sd, full["ecfg"], odec, [full["imgs"][0].to(torch.bfloat16).float()], full["ids"][:1],
                        full["tgts"][:1], img_id=full["img_id"], pad_id=full["pad_id"],
                        encoder_ft_layers_after=cfg.encoder_ft_layers_after)
    assert int(ref.num_tokens) == n
    d = abs(loss - float(ref.loss))
    assert d <= 1e-2 and d <= 1e-2 * abs(float(ref.loss)), (loss, float(ref.loss))
    assert set(g) == set(ref.grads)
    scale = max(v.float().norm().item() / v.numel() ** 0.5 for v in ref.grads.values())
    errs = {}
    for name, gv in g.items():
        r = ref.grads[name].float()
        errs[name] = ((gv.float().cpu() - r).norm() / max(r.norm().item(), 1e-3 * scale * r.numel() ** 0.5)).item()
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print("full-size oracle parity: |dloss|", d, "largest gradient errors", top)
    worst = top[0]
    # Tolerance at FULL depth: the bf16 path rounds after every op and the per-tensor gradient error grows
    # with the number of layers the gradient crosses (golden configs, 4-6 decoder layers: 1.3-2.1e-2 against
    # the 3e-2 bound; here 18 decoder + 8 encoder layers: 3.0-3.7e-2, a smooth spread led by the softmax-
    # sensitive q/k projections and the single-row soi embedding, no outlier).  Bound 5e-2; loss stays 1e-2.
    assert worst[1] <= 5e-2, worst
    assert sorted(errs.values())[len(errs) // 2] <= 3e-2, "median gradient error"
