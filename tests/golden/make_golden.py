"""Generate the golden vectors that pin ``oracle/zero_oracle.py``.

Run ONCE in the build container (needs /root/reference and transformers):

    python tests/golden/make_golden.py

What runs here is the real thing, not the oracle:
  * ``transformers`` 5.15.0 ``CLIPVisionModel`` / ``LlamaForCausalLM`` built from
    LOCAL config objects (no hub access) — the third-party arithmetic the
    reference dispatches;
  * the reference's own ``Zero`` (``/root/reference/src/models/adapter.py``) and
    ``reduce_sum`` (``/root/reference/src/distributed/torchshard.py``), imported;
  * the reference's surrogate surgery: ``build_decoder.py`` cannot be imported
    (needs ``peft``), so the golden run performs the surgery with plain
    ``ModuleList`` index-assign/pop calls in the order build_decoder.py:305-330
    gives, on the real ``LlamaForCausalLM``;
  * the forward_hf tail (engine_ar_instruction_llama.py:185-313) cannot be
    imported (needs ``torchvision`` through ``utils``); it is driven here with
    the same torch calls (``pad_sequence``, ``F.cross_entropy``) in the same
    order, from synthetic token-id lists.
Weights and inputs come from ``oracle.zero_oracle.make_params`` /
``synth_batch`` (seeded), so the fixtures hold OUTPUTS only; a checksum of the
weights is stored to catch RNG drift.
"""
import os
import sys
import types

import torch
import torch.nn.functional as F
from torch.nn.utils.rnn import pad_sequence
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

from transformers import CLIPVisionConfig, CLIPVisionModel, LlamaConfig, LlamaForCausalLM  # noqa: E402
from models.adapter import Zero  # noqa: E402  (reference)
from distributed.torchshard import reduce_sum  # noqa: E402  (reference)
import torch.distributed as dist  # noqa: E402

from oracle import zero_oracle as zo  # noqa: E402
from tests.golden.configs import GOLDEN_CONFIGS, shrink  # noqa: E402

IGNORE_INDEX = -100


def build_models(enc: zo.EncoderCfg, dec: zo.DecoderCfg, params_full, attn_impl, dtype):
    ccfg = CLIPVisionConfig(hidden_size=enc.hidden, intermediate_size=enc.intermediate,
                            num_hidden_layers=enc.layers, num_attention_heads=enc.heads,
                            image_size=enc.image, patch_size=enc.patch, hidden_act="quick_gelu",
                            layer_norm_eps=enc.eps, attention_dropout=0.0)
    ccfg._attn_implementation = attn_impl
    encoder = CLIPVisionModel(ccfg)
    sd = {}
    for k, v in params_full.items():
        if k.startswith("encoder.vision_model."):
            sd[k[len("encoder.vision_model."):]] = v      # 5.15 layout has no vision_model. prefix
    missing, unexpected = encoder.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in m for m in missing), missing
    # adapter.py:94 needs encoder.vision_model.post_layernorm (4.45 layout): alias, unregistered
    object.__setattr__(encoder, "vision_model", encoder)

    lcfg = LlamaConfig(vocab_size=dec.vocab, hidden_size=dec.hidden, intermediate_size=dec.intermediate,
                       num_hidden_layers=dec.layers, num_attention_heads=dec.heads,
                       num_key_value_heads=dec.kv_heads, head_dim=dec.head_dim, hidden_act="silu",
                       max_position_embeddings=131072, rms_norm_eps=dec.rms_eps,
                       tie_word_embeddings=dec.tie, attention_bias=False, mlp_bias=False,
                       rope_parameters={"rope_type": "llama3", "rope_theta": dec.rope_theta,
                                        "factor": dec.rope_factor, "low_freq_factor": dec.rope_low,
                                        "high_freq_factor": dec.rope_high,
                                        "original_max_position_embeddings": dec.rope_orig_ctx})
    lcfg._attn_implementation = attn_impl
    decoder = LlamaForCausalLM(lcfg)
    sd = {k[len("decoder."):]: v for k, v in params_full.items() if k.startswith("decoder.")}
    if dec.tie:
        sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    missing, unexpected = decoder.load_state_dict(sd, strict=False)
    assert not unexpected and not missing, (missing, unexpected)

    # --- surrogate surgery in the call order of build_decoder.py:283-330 ---------------
    ranges = [list(r) for r in dec.translators_range]
    for range_idx, trans_range in enumerate(ranges):
        trans_range.sort()
        index_offset = 0
        if range_idx > 0:
            index_offset += sum(len(ranges[i]) for i in range(range_idx))
        for i in range(dec.num_translators):
            lid = trans_range[i] - index_offset
            layer = decoder.model.layers[lid]
            Translator = type("Translator", (type(layer),), {})        # build_decoder.py:346-353
            tr = Translator.__new__(Translator)
            tr.__dict__.update(layer.__dict__)
            decoder.model.layers[lid] = tr
            trans_range.pop(i)
        for layer_id in trans_range:
            decoder.model.layers.pop(layer_id - index_offset)           # build_decoder.py:360-362
            index_offset += 1

    cfg = types.SimpleNamespace(adapter_mlp_depth=2, num_embd=dec.hidden, encoder_shave_last_n_layers=-1,
                                ignore_encoder_cls_token=False, device="cpu")
    model = Zero(cfg, encoder, decoder)
    with torch.no_grad():
        for k in ("adapter.0.weight", "adapter.0.bias", "adapter.2.weight", "adapter.2.bias"):
            mod, attr = k.rsplit(".", 1)
            getattr(model.adapter[int(mod.split(".")[1])], attr).copy_(params_full[k])
        model.embd_soi.copy_(params_full["embd_soi"])
        model.embd_eoi.copy_(params_full["embd_eoi"])
    model.to(dtype)
    return model


def set_trainable(model, ft_after, stage="finetune-encoder", translator_ids=()):
    # configs/zero/train_llama.py:259-323, TRAINING_STAGE == "finetune-encoder"; the 5.15
    # encoder has no "vision_model" in its names, so the encoder test is made on the
    # "encoder." root instead (same set of tensors).
    names = []
    for n, p in model.named_parameters():
        p.requires_grad = False
        if "adapter" in n or "embd_soi" in n or "embd_eoi" in n:
            p.requires_grad = True
        if stage == "pretrain":
            # configs/zero/train_llama.py:281-307
            if n.startswith("encoder.") and "post_layernorm" in n:
                p.requires_grad = True
            if "decoder" in n and "layers" in n:
                parts = n.split(".")
                if int(parts[parts.index("layers") + 1]) in translator_ids:
                    p.requires_grad = True
        elif n.startswith("encoder."):
            if "post_layernorm" in n:
                p.requires_grad = True
            if "layers" in n:
                parts = n.split(".")
                if int(parts[parts.index("layers") + 1]) >= ft_after:
                    p.requires_grad = True
        if p.requires_grad:
            names.append(n)
    return names


def forward_hf_tail(model, imgs, input_ids, targets, img_id, pad_id, max_seq_len):
    """engine_ar_instruction_llama.py:47-49 and :185-313 with the same torch calls."""
    embds_imgs = model._forward(imgs)                                    # engine:49
    wte = model.decoder.model.get_input_embeddings()                     # engine:186
    input_embds, targets_ids = [], []
    for b, (input_id, target) in enumerate(zip(input_ids, targets)):
        if img_id in input_id:
            at = input_id.index(img_id)
            n_img = embds_imgs[b].shape[0]
            inp1 = input_id[:at]
            inp2 = input_id[at + 1:]
            inp2 = inp2[: max_seq_len - len(inp1) - n_img]
            e1 = wte(torch.tensor(inp1, dtype=torch.long))
            e2 = wte(torch.tensor(inp2, dtype=torch.long))
            input_embds.append(torch.cat([e1, embds_imgs[b], e2], dim=0))
            targets_ids.append(target[:at] + [IGNORE_INDEX] * n_img + target[at + 1:])
        else:
            input_embds.append(wte(torch.tensor(input_id, dtype=torch.long)))
            targets_ids.append(target)
    input_embds = pad_sequence(input_embds, batch_first=True, padding_value=0.0)
    targets_ids = pad_sequence([torch.tensor(t, dtype=torch.long) for t in targets_ids],
                               batch_first=True, padding_value=IGNORE_INDEX)
    inputs_embeds = input_embds[:, :max_seq_len, :]
    target = targets_ids[:, :max_seq_len]
    inputs_embeds.retain_grad()
    output = model.decoder(inputs_embeds=inputs_embeds, use_cache=False, output_hidden_states=True)
    logits_raw = output["logits"]
    logits = logits_raw.float()
    logits = logits[:, :-1, :].contiguous()
    tgt = target[:, 1:].contiguous()
    tgt[tgt == pad_id] = IGNORE_INDEX
    loss = F.cross_entropy(logits.view(-1, logits.shape[-1]), tgt.view(-1),
                           ignore_index=IGNORE_INDEX, reduction="none")
    loss = loss.view(logits.shape[0], -1).sum()
    num_tokens = tgt.ne(IGNORE_INDEX).sum()
    loss_sum = loss.detach().clone()
    loss = reduce_sum(loss)                                              # engine:309 (reference fn)
    num_tokens = reduce_sum(num_tokens)                                  # engine:310
    loss = loss / num_tokens
    return dict(loss=loss, loss_sum=loss_sum, num_tokens=num_tokens, logits=logits_raw,
                inputs_embeds=inputs_embeds, target=target, embds_imgs=embds_imgs,
                dec_hidden=output.hidden_states)


slice_big = shrink


def to_oracle_name(n: str) -> str:
    if n.startswith("encoder."):
        return "encoder.vision_model." + n[len("encoder."):]
    return n


def run_config(name, spec):
    enc, dec = spec["enc"], spec["dec"]
    full = zo.make_params(enc, dec, spec["seed"], surrogate=False)
    imgs, ids, tgts, img_id, pad_id = zo.synth_batch(enc, dec, **spec["batch"])
    out = {}
    chk = torch.stack([v.double().sum() for _, v in sorted(full.items())]).sum()
    out["weights_checksum"] = chk.reshape(1)
    for tag, impl, dtype in (("fp32", "sdpa", torch.float32), ("eager", "eager", torch.float32),
                             ("bf16", "sdpa", torch.bfloat16)):
        torch.manual_seed(0)
        model = build_models(enc, dec, full, impl, dtype)
        stage = spec.get("stage", "finetune-encoder")
        # translator ids by the reference's own formula (configs/zero/train_llama.py:290-301)
        tids = []
        for i, r in enumerate(dec.translators_range):
            o = 0
            if i > 0:
                o += sum(len(dec.translators_range[i]) for i in range(i)) - 1
            tids.append([n - o for n in list(r)[:dec.num_translators]])
        tids = sum(tids, [])
        names = set_trainable(model, spec["ft_after"], stage, tids)
        model.train()
        r = forward_hf_tail(model, [i.to(dtype) for i in imgs], ids, tgts, img_id, pad_id, spec["max_seq_len"])
        r["loss"].backward()
        if tag == "eager":
            out["eager.logits"] = shrink(r["logits"].detach().float())
            out["eager.loss"] = r["loss"].detach().float().reshape(1)
            continue
        pre = tag + "."
        out[pre + "loss"] = r["loss"].detach().float().reshape(1)
        out[pre + "loss_sum"] = r["loss_sum"].float().reshape(1)
        out[pre + "num_tokens"] = r["num_tokens"].reshape(1)
        out[pre + "logits"] = shrink(r["logits"].detach().float())
        out[pre + "inputs_embeds"] = shrink(r["inputs_embeds"].detach().float())
        out[pre + "d_inputs_embeds"] = shrink(r["inputs_embeds"].grad.detach().float())
        out[pre + "target"] = r["target"]
        out[pre + "embds_imgs"] = shrink(torch.stack([e.detach().float() for e in r["embds_imgs"]]))
        for i, h in enumerate(r["dec_hidden"]):
            out[f"{pre}dec_hidden.{i}"] = shrink(h.detach().float())
        pd = dict(model.named_parameters())
        for n in names:
            g = pd[n].grad
            assert g is not None, n
            on = to_oracle_name(n)
            out[f"{pre}grad.{on}"] = slice_big(g.detach().float())
            out[f"{pre}gradnorm.{on}"] = g.detach().float().norm().reshape(1)
        if tag == "fp32":
            # encoder hidden states straight from the HF module (output_hidden_states=True)
            with torch.no_grad():
                eo = model.encoder(torch.cat(imgs, 0), output_hidden_states=True)
            for i, h in enumerate(eo.hidden_states):
                out[f"fp32.enc_hidden.{i}"] = shrink(h.float())
            out["fp32.enc_last"] = shrink(eo.last_hidden_state.float())
    out = {k: v.contiguous().clone() for k, v in out.items()}
    path = os.path.join(HERE, f"{name}.safetensors")
    save_file(out, path)
    print(name, "->", path, f"{os.path.getsize(path) / 1e6:.2f} MB", "loss", float(out["fp32.loss"]),
          "bf16 loss", float(out["bf16.loss"]))


def run_graft(name, spec):
    """ref:inference/vlm_infer.py:201-270 on the real modules: Zero._forward, splice at the image token,
    pad_sequence, then transformers' own generate (greedy, KV cache)."""
    enc, dec = spec["enc"], spec["dec"]
    full = zo.make_params(enc, dec, spec["seed"], surrogate=False)
    imgs, ids, _, img_id, _ = zo.synth_batch(enc, dec, **spec["batch"])
    out = {}
    torch.manual_seed(0)
    model = build_models(enc, dec, full, "sdpa", torch.float32)
    model.eval()
    model.requires_grad_(False)
    with torch.inference_mode():
        embds_imgs = model._forward(imgs)                                  # vlm_infer.py:201
        wte = model.decoder.model.get_input_embeddings()
        input_embds = []
        for i, input_id in enumerate(ids):
            at = input_id.index(img_id)
            e1 = wte(torch.tensor(input_id[:at], dtype=torch.long))
            e2 = wte(torch.tensor(input_id[at + 1:], dtype=torch.long))
            input_embds.append(torch.cat([e1, embds_imgs[i], e2], dim=0))
        input_embds = pad_sequence(input_embds, batch_first=True, padding_value=0.0)
        mask = torch.ones(input_embds.shape[0], input_embds.shape[1])
        gc = model.decoder.generation_config
        gc.eos_token_id = None
        gc.pad_token_id = None
        toks = model.decoder.generate(inputs_embeds=input_embds, attention_mask=mask, do_sample=False,
                                      max_new_tokens=spec["max_new_tokens"], use_cache=True)
        logits = model.decoder(inputs_embeds=input_embds, use_cache=False)["logits"]
    out["fp32.gen_tokens"] = toks.clone()
    out["fp32.inputs_embeds"] = shrink(input_embds.float().clone())
    out["fp32.prompt_last_logits"] = logits[:, -1].float().clone()
    path = os.path.join(HERE, f"{name}.safetensors")
    save_file({k: v.contiguous() for k, v in out.items()}, path)
    print(name, "->", path, "tokens", toks.tolist())


def unit_vectors():
    """Small known-answer vectors straight from the transformers modules."""
    from transformers.models.llama.modeling_llama import LlamaRMSNorm, LlamaRotaryEmbedding, apply_rotary_pos_emb
    from transformers.activations import ACT2FN
    out = {}
    g = torch.Generator().manual_seed(7)
    for tag, dec in (("3b", zo.LLAMA32_3B), ("8b", zo.LLAMA31_8B)):
        lcfg = LlamaConfig(hidden_size=dec.hidden, num_attention_heads=dec.heads, head_dim=dec.head_dim,
                           num_hidden_layers=1, vocab_size=32, intermediate_size=64,
                           max_position_embeddings=131072,
                           rope_parameters={"rope_type": "llama3", "rope_theta": dec.rope_theta,
                                            "factor": dec.rope_factor, "low_freq_factor": dec.rope_low,
                                            "high_freq_factor": dec.rope_high,
                                            "original_max_position_embeddings": dec.rope_orig_ctx})
        rot = LlamaRotaryEmbedding(lcfg)
        out[f"inv_freq.{tag}"] = rot.inv_freq.clone().float()
        x = torch.zeros(1, 1, dtype=torch.bfloat16)
        pos = torch.arange(0, 4096, 61)[None]
        cos, sin = rot(x, pos)
        out[f"rope_pos.{tag}"] = pos[0].clone()
        out[f"rope_cos_bf16.{tag}"] = cos[0].float()
        out[f"rope_sin_bf16.{tag}"] = sin[0].float()
    q = torch.randn(1, 3, 7, 128, generator=g)
    k = torch.randn(1, 1, 7, 128, generator=g)
    lcfg = LlamaConfig(hidden_size=3072, num_attention_heads=24, head_dim=128, num_hidden_layers=1,
                       vocab_size=32, intermediate_size=64, max_position_embeddings=131072,
                       rope_parameters={"rope_type": "llama3", "rope_theta": 500000.0, "factor": 32.0,
                                        "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                        "original_max_position_embeddings": 8192})
    rot = LlamaRotaryEmbedding(lcfg)
    cos, sin = rot(q, torch.arange(7)[None])
    qe, ke = apply_rotary_pos_emb(q, k, cos, sin)
    out["rope.q"], out["rope.k"], out["rope.qe"], out["rope.ke"] = q, k, qe, ke
    x = torch.randn(5, 96, generator=g) * 3
    w = torch.randn(96, generator=g)
    rn = LlamaRMSNorm(96, eps=1e-5)
    with torch.no_grad():
        rn.weight.copy_(w)
    out["rms.x"], out["rms.w"] = x, w
    out["rms.y"] = rn(x).detach()
    out["rms.y_bf16"] = rn.to(torch.bfloat16)(x.to(torch.bfloat16)).detach().float()
    out["act.x"] = torch.linspace(-6, 6, 97)
    out["act.quick_gelu"] = ACT2FN["quick_gelu"](out["act.x"])
    out["act.gelu"] = torch.nn.GELU()(out["act.x"])
    out["act.silu"] = ACT2FN["silu"](out["act.x"])
    path = os.path.join(HERE, "unit.safetensors")
    save_file({k: v.contiguous() for k, v in out.items()}, path)
    print("unit ->", path)


if __name__ == "__main__":
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=0, world_size=1)   # reduce_sum queries the world size
    torch.set_num_threads(8)
    only = sys.argv[1:]                 # e.g. `make_golden.py pretrain` regenerates one fixture
    if not only:
        unit_vectors()
    for name, spec in GOLDEN_CONFIGS.items():
        if not only or name in only:
            (run_graft if spec.get("graft") else run_config)(name, spec)
    dist.destroy_process_group()
