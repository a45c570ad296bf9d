"""Zero-shot graft inference on the GPU: single-query attention kernel, KV-cache decoding against a
full recompute (HIP training-path forward and the fp32 oracle), greedy generation against the golden
transformers run, and the tensor-parallel path (W = 2 rank threads on one device)."""
import os
import threading

import pytest
import torch
from safetensors.torch import load_file

from oracle import zero_oracle as zo
from tests.golden.configs import GOLDEN_CONFIGS

pytestmark = pytest.mark.gpu
DEV = "cuda"


def maxrel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def _model(spec):
    from tests.helpers import build_model, make_cfg
    enc, dec = spec["enc"], spec["dec"]
    params = zo.make_params(enc, dec, spec["seed"], surrogate=False)
    cfg = make_cfg(enc, dec, DEV, spec["ft_after"], spec["max_seq_len"])
    cfg.enable_translator = False                                     # ref:vlm_infer.py:57
    model = build_model(cfg, params)
    model.eval()
    return params, cfg, model


@pytest.mark.parametrize("B,Hq,Hkv,hd,rows,lens", [(2, 4, 2, 128, 300, (300, 123)), (1, 8, 1, 128, 1030, (1027,)),
                                                   (3, 6, 6, 64, 70, (1, 64, 70))])
def test_attn_decode_kernel(B, Hq, Hkv, hd, rows, lens):
    from zero_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    q = torch.randn(B, Hq * hd, generator=g).to(torch.bfloat16).to(DEV)
    kc = torch.randn(B, rows, Hkv * hd, generator=g).to(torch.bfloat16).to(DEV)
    vc = torch.randn(B, rows, Hkv * hd, generator=g).to(torch.bfloat16).to(DEV)
    ln = torch.tensor(lens, dtype=torch.int32, device=DEV)
    out = ops.attn_decode(q, kc, vc, ln, max(lens), Hq, Hkv, hd, hd ** -0.5)
    rep = Hq // Hkv
    for b in range(B):
        L = lens[b]
        qf = q[b].float().view(Hq, hd)
        kf = kc[b, :L].float().view(L, Hkv, hd).repeat_interleave(rep, dim=1)
        vf = vc[b, :L].float().view(L, Hkv, hd).repeat_interleave(rep, dim=1)
        p = torch.softmax(torch.einsum("hd,lhd->hl", qf, kf) * hd ** -0.5, dim=-1)
        ref = torch.einsum("hl,lhd->hd", p, vf).reshape(-1)
        assert maxrel(out[b], ref) < 8e-3, (b, L)


@pytest.mark.parametrize("B,Hq,Hkv,hd,rows,pos", [(1, 8, 1, 128, 1100, 1027), (2, 24, 8, 128, 300, 63), (2, 24, 8, 128, 300, 64),
                                                  (3, 6, 6, 64, 70, 0), (4, 32, 2, 64, 200, 199), (1, 64, 8, 128, 900, 800)])
def test_attn_decode_fused_kernel(B, Hq, Hkv, hd, rows, pos):
    """RoPE + cache append + split-cache attention + merge (zk_attn_decode_fused) against fp32 torch, and
    against the separate kernels it replaces (zk_rope, append, zk_attn_decode)."""
    from zero_amd import ops
    g = torch.Generator(device="cpu").manual_seed(11)
    W = (Hq + 2 * Hkv) * hd
    qkv = torch.randn(B, W, generator=g).to(torch.bfloat16).to(DEV)
    kc = torch.randn(B, rows, Hkv * hd, generator=g).to(torch.bfloat16).to(DEV)
    vc = torch.randn(B, rows, Hkv * hd, generator=g).to(torch.bfloat16).to(DEV)
    ang = torch.rand(rows, hd // 2, generator=g) * 6.28
    cos_t, sin_t = ang.cos().to(DEV).contiguous(), ang.sin().to(DEV).contiguous()
    kc0, vc0, qkv0 = kc.clone(), vc.clone(), qkv.clone()
    out = ops.attn_decode_fused(qkv, cos_t, sin_t, kc, vc, pos, Hq, Hkv, hd, hd ** -0.5)
    assert torch.equal(qkv, qkv0)
    # the separate kernels
    q2 = qkv0.clone()
    ops.rope_(q2, cos_t[pos:pos + 1], sin_t[pos:pos + 1], 1, Hq + Hkv, hd)
    k_new, v_new = q2[:, Hq * hd:(Hq + Hkv) * hd], q2[:, (Hq + Hkv) * hd:]
    kc2, vc2 = kc0.clone(), vc0.clone()
    kc2[:, pos], vc2[:, pos] = k_new, v_new
    ln = torch.full((B,), pos + 1, dtype=torch.int32, device=DEV)
    old = ops.attn_decode(q2[:, :Hq * hd], kc2, vc2, ln, pos + 1, Hq, Hkv, hd, hd ** -0.5)
    # cache: row pos appended (same values up to the RoPE's last-bit rounding), every other row untouched
    assert torch.equal(vc, vc2)
    assert maxrel(kc[:, pos], kc2[:, pos]) < 8e-3
    keep = torch.ones(rows, dtype=torch.bool, device=DEV)
    keep[pos] = False
    assert torch.equal(kc[:, keep], kc0[:, keep])
    assert maxrel(out, old) < 8e-3
    rep = Hq // Hkv
    for b in range(B):
        L = pos + 1
        qf = q2[b, :Hq * hd].float().view(Hq, hd)
        kf = kc2[b, :L].float().view(L, Hkv, hd).repeat_interleave(rep, dim=1)
        vf = vc2[b, :L].float().view(L, Hkv, hd).repeat_interleave(rep, dim=1)
        p = torch.softmax(torch.einsum("hd,lhd->hl", qf, kf) * hd ** -0.5, dim=-1)
        ref = torch.einsum("hl,lhd->hd", p, vf).reshape(-1)
        assert maxrel(out[b], ref) < 8e-3, (b, L)
    with pytest.raises(ValueError):
        ops.attn_decode_fused(qkv, cos_t, sin_t, kc, vc, pos, Hq, Hkv, hd, hd ** -0.5, ws=torch.empty(8, device=DEV))


@pytest.mark.parametrize("native", [True, False, "unsplit"], ids=["native-layer-driver", "python-launches", "native-unsplit-attention"])
def test_kv_cache_decode_matches_full_recompute(native, monkeypatch):
    """Prefill + 5 cached single-token steps on unit-scale random embeddings == the causal forward over
    the whole grown sequence (HIP training path, and the fp32 oracle): catches cache / position bugs."""
    from zero_amd.inference import decode
    from zero_amd.inference.decode import KVCache, forward_cached
    if native == "unsplit":
        native = True
        monkeypatch.setattr(decode, "USE_SPLIT_ATTENTION", False)
    monkeypatch.setattr(decode, "USE_NATIVE_DECODE", native)
    spec = GOLDEN_CONFIGS["graft"]
    params, cfg, model = _model(spec)
    dec = spec["dec"]
    g = torch.Generator(device="cpu").manual_seed(3)
    B, S0, steps = 2, 37, 5
    xs = torch.randn(B, S0 + steps, dec.hidden, generator=g).to(torch.bfloat16)
    cache = KVCache(len(model.decoder.model.layers), B, S0 + steps, dec.kv_heads * dec.head_dim, DEV)
    got = [forward_cached(model.decoder, xs[:, :S0].to(DEV), cache)[:, 0]]
    for t in range(steps):
        got.append(forward_cached(model.decoder, xs[:, S0 + t:S0 + t + 1].to(DEV), cache)[:, 0])
    assert cache.length == S0 + steps and (getattr(cache, "_plan", None) is not None) == native
    with torch.no_grad():
        full = model.decoder(inputs_embeds=xs.to(DEV), use_cache=False)["logits"]
    ref = zo.llama_forward(params, xs.float(), dec)
    for t, lg in enumerate(got):
        pos = S0 - 1 + t
        assert maxrel(lg, full[:, pos]) <= 1e-2, t
        assert maxrel(lg, ref[:, pos]) <= 2e-2, t


def test_kv_cache_decode_at_the_70b_shard_shape():
    """BASELINE configs[4] at the shape ONE rank of the 8-way split holds (d = 8192, 8 q heads / 1 kv head x 128,
    f = 28672 / 8 = 3584, V = 128256; 2 layers): prefill + 4 cached single-token steps through the native decode
    drivers (GEMV with fused norm / SwiGLU operands, split-cache attention with one kv head) == the causal
    training-path forward over the whole grown sequence, and == the fp32 oracle on the same weights.  The collectives
    of the split are identities on one rank (TPContext()): the per-rank arithmetic is what is checked."""
    from zero_amd.configs.config_base import LLAMA_SHAPES
    from zero_amd.inference import decode
    from zero_amd.inference.decode import KVCache, forward_cached
    from zero_amd.models.llama import LlamaConfig, LlamaForCausalLM
    assert decode.USE_NATIVE_DECODE and decode.USE_SPLIT_ATTENTION
    sh = dict(LLAMA_SHAPES[("3.1", "70")])
    sh.pop("translators_range")
    factor = sh.pop("rope_factor")
    W, L = 8, 2
    lcfg = LlamaConfig(**{**sh, "num_attention_heads": sh["num_attention_heads"] // W,
                          "num_key_value_heads": sh["num_key_value_heads"] // W,
                          "intermediate_size": sh["intermediate_size"] // W, "num_hidden_layers": L})
    lcfg.rope_scaling = dict(rope_type="llama3", factor=factor, low_freq_factor=1.0, high_freq_factor=4.0,
                             original_max_position_embeddings=8192)
    assert (lcfg.hidden_size, lcfg.num_attention_heads, lcfg.num_key_value_heads, lcfg.intermediate_size) == (8192, 8, 1, 3584)
    torch.manual_seed(70)
    dec = LlamaForCausalLM(lcfg, device=DEV)
    with torch.no_grad():
        for n, p in dec.named_parameters():
            if "norm" in n:
                p.copy_(1.0 + 0.1 * torch.randn_like(p.float()))
            elif "o_proj" in n or "down_proj" in n:
                p.normal_(0.0, 0.02 / (2 * L) ** 0.5)
            else:
                p.normal_(0.0, 0.02)
    dec.eval()
    odec = zo.DecoderCfg(hidden=lcfg.hidden_size, heads=lcfg.num_attention_heads, kv_heads=lcfg.num_key_value_heads,
                         head_dim=lcfg.head_dim, intermediate=lcfg.intermediate_size, layers=L, vocab=lcfg.vocab_size,
                         rope_factor=factor, tie=lcfg.tie_word_embeddings)
    params = {"decoder." + k: v.detach().float().cpu() for k, v in dec.state_dict().items()}
    g = torch.Generator(device="cpu").manual_seed(7)
    B, S0, steps = 1, 203, 4                          # the cache crosses a 64-key chunk boundary at step 2... 3 chunks + 1
    xs = torch.randn(B, S0 + steps, lcfg.hidden_size, generator=g).to(torch.bfloat16)
    cache = KVCache(L, B, S0 + steps, lcfg.num_key_value_heads * lcfg.head_dim, DEV)
    got = [forward_cached(dec, xs[:, :S0].to(DEV), cache)[:, 0]]
    for t in range(steps):
        got.append(forward_cached(dec, xs[:, S0 + t:S0 + t + 1].to(DEV), cache)[:, 0])
    assert cache.length == S0 + steps and getattr(cache, "_plan", None) is not None      # the native drivers ran
    with torch.no_grad():
        full = dec(inputs_embeds=xs.to(DEV), use_cache=False)["logits"]
    rows = [S0 - 1 + t for t in range(steps + 1)]
    torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16"))))
    ref = zo.llama_forward(params, xs.float(), odec)[:, rows]
    for t, lg in enumerate(got):
        assert maxrel(lg, full[:, rows[t]]) <= 1e-2, t
        assert maxrel(lg, ref[:, t]) <= 2e-2, t
        assert torch.equal(lg.float().argmax(-1).cpu(), ref[:, t].argmax(-1)) or maxrel(lg, ref[:, t]) <= 5e-3


def test_graft_generate_matches_golden(golden_dir):
    from zero_amd.inference.vlm_infer import graft_generate, graft_inputs
    spec = GOLDEN_CONFIGS["graft"]
    gold = load_file(os.path.join(golden_dir, "graft.safetensors"))
    params, cfg, model = _model(spec)
    imgs, ids, _, img_id, _ = zo.synth_batch(spec["enc"], spec["dec"], **spec["batch"])
    dimgs = [i.to(DEV).to(torch.bfloat16) for i in imgs]
    x = graft_inputs(model, dimgs, ids, img_id)
    assert tuple(x.shape) == (2, 6 + spec["enc"].tokens + 2 + 9, spec["dec"].hidden)
    toks = graft_generate(model, dimgs, ids, img_id, max_new_tokens=spec["max_new_tokens"])
    assert torch.equal(toks.cpu(), gold["fp32.gen_tokens"])
    # the drop-in call of the reference (vlm_infer.py:264-270)
    t2 = model.decoder.generate(inputs_embeds=x, attention_mask=torch.ones(x.shape[:2], device=DEV), do_sample=False,
                                max_new_tokens=4, use_cache=True)
    assert torch.equal(t2.cpu(), gold["fp32.gen_tokens"][:, :4])
    # EOS: stop + pad
    eos = int(gold["fp32.gen_tokens"][0, 0])
    t3 = model.decoder.generate(inputs_embeds=x[:1], max_new_tokens=6, eos_token_id=eos, pad_token_id=3)
    assert t3.shape[1] == 1 and int(t3[0, 0]) == eos


def test_tensor_parallel_two_rank_threads():
    """shard_llama_ + TPContext through the real layer code with W = 2 (rank threads + rendezvous
    all-reduce on one device) against the unsharded decoder."""
    from zero_amd.distributed.tensor_parallel import ThreadSimGroup, shard_llama_
    from zero_amd.inference.decode import KVCache, forward_cached, generate
    spec = GOLDEN_CONFIGS["graft"]
    dec = spec["dec"]
    g = torch.Generator(device="cpu").manual_seed(9)
    B, S0 = 2, 29
    xs = torch.randn(B, S0 + 2, dec.hidden, generator=g).to(torch.bfloat16).to(DEV)
    _, _, base = _model(spec)
    cache = KVCache(dec.layers, B, S0 + 2, dec.kv_heads * dec.head_dim, DEV)
    want = [forward_cached(base.decoder, xs[:, :S0], cache)[:, 0], forward_cached(base.decoder, xs[:, S0:S0 + 1], cache)[:, 0]]
    want_toks = generate(base.decoder, xs[:, :S0], max_new_tokens=5)
    W = 2
    group = ThreadSimGroup(W)
    shards = []
    for r in range(W):
        _, _, m = _model(spec)
        shard_llama_(m.decoder, r, W)
        assert m.decoder.model.layers[0].self_attn.q_proj.weight.shape[0] == dec.heads // W * dec.head_dim
        assert m.decoder.model.layers[0].mlp.down_proj.weight.shape[1] == dec.intermediate // W
        shards.append(m.decoder)
    results, errors = [None] * W, []

    def run(r):
        try:
            torch.cuda.set_device(0)
            tp = group.context(r)
            c = KVCache(dec.layers, B, S0 + 2, dec.kv_heads // W * dec.head_dim, DEV)
            a = forward_cached(shards[r], xs[:, :S0], c, tp)[:, 0]
            b = forward_cached(shards[r], xs[:, S0:S0 + 1], c, tp)[:, 0]
            results[r] = (a, b, generate(shards[r], xs[:, :S0], max_new_tokens=5, tp=tp))
        except Exception as e:      # a dead rank must not leave the other one waiting at the rendezvous
            errors.append(e)
            group._bar.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=100)
    assert not errors, errors
    for r in range(W):
        a, b, toks = results[r]
        assert maxrel(a, want[0]) <= 1.5e-2 and maxrel(b, want[1]) <= 1.5e-2, r
        assert torch.equal(toks, want_toks), r
    assert torch.equal(results[0][0], results[1][0])                  # every rank ends with the same logits
