"""Shared test plumbing: build the zero_amd model for an oracle config and load oracle params."""
import types

import torch

from oracle import zero_oracle as zo


def make_cfg(enc: zo.EncoderCfg, dec: zo.DecoderCfg, device, ft_after, max_seq_len):
    from zero_amd.configs.config_base import ConfigBase
    from zero_amd.models.clip import CLIPVisionConfig
    from zero_amd.models.llama import LlamaConfig
    cfg = ConfigBase()
    cfg.device = torch.device(device)
    cfg.encoder_config = CLIPVisionConfig(hidden_size=enc.hidden, intermediate_size=enc.intermediate,
                                          num_hidden_layers=enc.layers, num_attention_heads=enc.heads,
                                          image_size=enc.image, patch_size=enc.patch, layer_norm_eps=enc.eps)
    cfg.decoder_config = LlamaConfig(
        vocab_size=dec.vocab, hidden_size=dec.hidden, intermediate_size=dec.intermediate,
        num_hidden_layers=dec.layers, num_attention_heads=dec.heads, num_key_value_heads=dec.kv_heads,
        head_dim=dec.head_dim, rms_norm_eps=dec.rms_eps, rope_theta=dec.rope_theta,
        rope_scaling=dict(rope_type="llama3", factor=dec.rope_factor, low_freq_factor=dec.rope_low,
                          high_freq_factor=dec.rope_high, original_max_position_embeddings=dec.rope_orig_ctx),
        tie_word_embeddings=dec.tie)
    cfg.num_embd = dec.hidden
    cfg.translators_range = tuple(list(r) for r in dec.translators_range)
    cfg.num_translators = dec.num_translators
    cfg.enable_translator = True
    cfg.encoder_ft_layers_after = ft_after
    cfg.max_seq_len = max_seq_len
    cfg.llama_model_version, cfg.llama_model_scale = "3.2", "3"
    return cfg


def build_model(cfg, params_surrogate):
    """Through the reference-shaped factories: load_clip, load_hf_llama3 (incl. the surgery), Zero."""
    from zero_amd.build_decoder import load_hf_llama3
    from zero_amd.build_encoder import load_clip
    from zero_amd.models.adapter import Zero
    encoder, _, _ = load_clip(cfg)
    num_embd, decoder, _, _, _ = load_hf_llama3(cfg)
    model = Zero(cfg, encoder, decoder)
    sd = {k: v.to(torch.bfloat16) for k, v in params_surrogate.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(m == "decoder.lm_head.weight" for m in missing), missing
    return model
