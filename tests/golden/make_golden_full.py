"""Generate tests/golden/full_c3.safetensors and full_c4.safetensors: ONE sample at BASELINE.json's full width
AND depth (configs[2]: CLIP-L/14-336 + adapter + the 18-layer Llama-3.2-3B surrogate; configs[3]: the 19-layer
Llama-3.1-8B surrogate; S = 768, V = 128256) through the REAL transformers modules and the reference's own
``Zero`` / ``reduce_sum`` — the same drivers as make_golden.py, fp32 and bf16.

    python tests/golden/make_golden_full.py c3        (≈ 25 GB of RAM, a few minutes on 8 cores)
    python tests/golden/make_golden_full.py c4        (≈ 45 GB of RAM)
    python tests/golden/make_golden_full.py c3c c4c   (the same models with the conditioned init, configs.FULL_CONFIGS)

Purpose ("anchor the tolerances"): the bf16-vs-fp32 distance of transformers itself at full depth
is the floor the HIP path's gradient tolerance is stated against (tests/test_fullsize_gpu.py).  Weights come
from tests/golden/hashweights.py (bit-identical on CPU and GPU), stored here only as a checksum; activations
and 2-D gradients are stored sub-sampled (configs.shrink), vectors whole.
"""
import gc
import os
import sys
import time

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import tests.golden.make_golden as mg  # noqa: E402  (imports transformers + the reference's Zero / reduce_sum)
from oracle import zero_oracle as zo  # noqa: E402
from tests.golden.configs import FULL_CONFIGS, shrink  # noqa: E402
from tests.golden.hashweights import checksum, hash_params  # noqa: E402
from transformers import CLIPVisionConfig, CLIPVisionModel, LlamaConfig, LlamaForCausalLM  # noqa: E402
import torch.distributed as dist  # noqa: E402
import types  # noqa: E402

def build(enc, dec, params, dtype):
    """Real modules with the surrogate's weights: the ORIGINAL-depth LlamaForCausalLM is built, the surviving
    layers get the fixture's weights in their original slots, then the surgery runs on the real ModuleList in
    the reference's call order (as make_golden.build_models)."""
    ccfg = CLIPVisionConfig(hidden_size=enc.hidden, intermediate_size=enc.intermediate,
                            num_hidden_layers=enc.layers, num_attention_heads=enc.heads, image_size=enc.image,
                            patch_size=enc.patch, hidden_act="quick_gelu", layer_norm_eps=enc.eps,
                            attention_dropout=0.0)
    ccfg._attn_implementation = "sdpa"
    encoder = CLIPVisionModel(ccfg)
    sd = {k[len("encoder.vision_model."):]: v.float() for k, v in params.items() if k.startswith("encoder.vision_model.")}
    missing, unexpected = encoder.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    object.__setattr__(encoder, "vision_model", encoder)

    lcfg = LlamaConfig(vocab_size=dec.vocab, hidden_size=dec.hidden, intermediate_size=dec.intermediate,
                       num_hidden_layers=dec.layers, num_attention_heads=dec.heads,
                       num_key_value_heads=dec.kv_heads, head_dim=dec.head_dim, hidden_act="silu",
                       max_position_embeddings=131072, rms_norm_eps=dec.rms_eps, tie_word_embeddings=dec.tie,
                       attention_bias=False, mlp_bias=False,
                       rope_parameters={"rope_type": "llama3", "rope_theta": dec.rope_theta, "factor": dec.rope_factor,
                                        "low_freq_factor": dec.rope_low, "high_freq_factor": dec.rope_high,
                                        "original_max_position_embeddings": dec.rope_orig_ctx})
    lcfg._attn_implementation = "sdpa"
    # original depth on the META device (no storage for the layers the surgery removes: the 32-layer 8B model
    # would be 32 GB in fp32); storage is allocated after the surgery and filled tensor by tensor
    with torch.device("meta"):
        decoder = LlamaForCausalLM(lcfg)
    keep = zo.surrogate_layer_map(dec.layers, dec.translators_range, dec.num_translators)
    ranges = [list(r) for r in dec.translators_range]
    for range_idx, trans_range in enumerate(ranges):
        trans_range.sort()
        index_offset = sum(len(ranges[i]) for i in range(range_idx)) if range_idx > 0 else 0
        for i in range(dec.num_translators):
            lid = trans_range[i] - index_offset
            layer = decoder.model.layers[lid]
            Translator = type("Translator", (type(layer),), {})
            tr = Translator.__new__(Translator)
            tr.__dict__.update(layer.__dict__)
            decoder.model.layers[lid] = tr
            trans_range.pop(i)
        for layer_id in trans_range:
            decoder.model.layers.pop(layer_id - index_offset)
            index_offset += 1
    assert len(decoder.model.layers) == len(keep)
    decoder.to_empty(device="cpu")
    from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding
    decoder.model.rotary_emb = LlamaRotaryEmbedding(config=lcfg)      # non-persistent inv_freq buffers, rebuilt on CPU
    if dec.tie:
        decoder.tie_weights()
    own = dict(decoder.named_parameters())
    seen = set()
    with torch.no_grad():
        for k, v in params.items():
            if k.startswith("decoder."):
                own[k[len("decoder."):]].copy_(v)                       # surrogate numbering == post-surgery slots
                seen.add(k[len("decoder."):])
    left = set(own) - seen
    assert left <= {"lm_head.weight"} and (dec.tie or not left), left
    if dec.tie:
        assert decoder.lm_head.weight.data_ptr() == decoder.model.embed_tokens.weight.data_ptr()
    cfg = types.SimpleNamespace(adapter_mlp_depth=2, num_embd=dec.hidden, encoder_shave_last_n_layers=-1,
                                ignore_encoder_cls_token=False, device="cpu")
    model = mg.Zero(cfg, encoder, decoder)
    with torch.no_grad():
        for k in ("adapter.0.weight", "adapter.0.bias", "adapter.2.weight", "adapter.2.bias"):
            mod, attr = k.rsplit(".", 1)
            getattr(model.adapter[int(mod.split(".")[1])], attr).copy_(params[k].float())
        model.embd_soi.copy_(params["embd_soi"].float())
        model.embd_eoi.copy_(params["embd_eoi"].float())
    model.to(dtype)
    return model


def run(name):
    spec = FULL_CONFIGS[name]
    enc, dec = spec["enc"], spec["dec"]
    t0 = time.time()
    params = hash_params(enc, dec, spec["seed"], "cpu", dtype=torch.bfloat16, init=spec.get("init", "plain"))
    out = {"weights_checksum": checksum(params)}
    print(name, "weights", f"{sum(v.numel() for v in params.values()) / 1e9:.2f} G params", f"{time.time() - t0:.0f} s",
          flush=True)
    imgs, ids, tgts, img_id, pad_id = zo.synth_batch(enc, dec, **spec["batch"])
    model = build(enc, dec, params, torch.float32)
    del params
    gc.collect()
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        t0 = time.time()
        model.to(dtype)
        model.zero_grad(set_to_none=True)
        names = mg.set_trainable(model, spec["ft_after"])
        model.train()
        r = mg.forward_hf_tail(model, [i.to(torch.bfloat16).to(dtype) for i in imgs], ids, tgts, img_id, pad_id, 4096)
        r["loss"].backward()
        pre = tag + "."
        out[pre + "loss"] = r["loss"].detach().float().reshape(1)
        out[pre + "loss_sum"] = r["loss_sum"].float().reshape(1)
        out[pre + "num_tokens"] = r["num_tokens"].reshape(1)
        out[pre + "logits"] = shrink(r["logits"].detach().float())
        # the logits of the labelled rows (what LMHeadLossFn forms), first 256 vocabulary entries
        tgt = r["target"][0]
        rows = torch.nonzero(tgt[1:] != -100)[:, 0]
        out[pre + "label_rows"] = rows.clone()
        out[pre + "logits_label_rows"] = r["logits"].detach().float()[0, rows, :256].contiguous()
        out[pre + "inputs_embeds"] = shrink(r["inputs_embeds"].detach().float())
        out[pre + "d_inputs_embeds"] = shrink(r["inputs_embeds"].grad.detach().float())
        out[pre + "dec_hidden_last"] = shrink(r["dec_hidden"][-1].detach().float())
        pd = dict(model.named_parameters())
        for n in names:
            g = pd[n].grad.detach().float()
            on = mg.to_oracle_name(n)
            out[f"{pre}grad.{on}"] = shrink(g)
            out[f"{pre}gradnorm.{on}"] = g.norm().reshape(1)
            out[f"{pre}gradmax.{on}"] = g.abs().max().reshape(1)
        print(name, tag, "loss", float(out[pre + "loss"]), "tokens", int(out[pre + "num_tokens"]),
              f"{time.time() - t0:.0f} s", flush=True)
        del r
        gc.collect()
    out = {k: v.contiguous().clone() for k, v in out.items()}
    path = os.path.join(HERE, f"full_{name}.safetensors")
    save_file(out, path)
    print(name, "->", path, f"{os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        dist.init_process_group("gloo", rank=0, world_size=1)
    torch.set_num_threads(8)
    for name in (sys.argv[1:] or ["c3"]):
        run(name)
    dist.destroy_process_group()
