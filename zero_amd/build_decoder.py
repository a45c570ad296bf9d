"""``load_hf_llama3`` / ``build_surrogate_model`` with the reference's signatures
(ref:build_decoder.py:17-146,268-362).

Returns ``(num_embd, model, tokenizer, FSDP_DECODER_LAYER, is_giant_model)``.  The decoder is
built from the public Llama-3 shape constants (or ``cfg.decoder_config``) because the gated
hub weights and the tokenizer cannot be fetched offline; ``tokenizer`` is None unless
``cfg.tokenizer`` supplies one."""
from __future__ import annotations

from copy import deepcopy

import numpy as np
import torch

from .configs.config_base import LLAMA_SHAPES
from .models.llama import LlamaConfig, LlamaDecoderLayer, LlamaForCausalLM


def print_model_numel(model, model_name="model"):
    n = sum(p.numel() for p in set(model.parameters()))
    print(f"{model_name}: {n / 1e9:.3f} B params")


def load_hf_llama3(cfg: object):
    ver, scale = cfg.llama_model_version, cfg.llama_model_scale
    cfg.decoder_hf_model_name = f"meta-llama/Llama-{ver}-{scale}B-Instruct"      # side effect, build_decoder.py:41
    lcfg = getattr(cfg, "decoder_config", None)
    if lcfg is None:
        s = dict(LLAMA_SHAPES[(ver, scale)])
        s.pop("translators_range")
        factor = s.pop("rope_factor")
        lcfg = LlamaConfig(**s)
        lcfg.rope_scaling = dict(rope_type="llama3", factor=factor, low_freq_factor=1.0, high_freq_factor=4.0,
                                 original_max_position_embeddings=8192)
    is_giant_model = lcfg.num_hidden_layers >= 80
    # (70B: the reference spreads it with device_map="balanced"; here one MI355X holds the 141 GB, and
    #  distributed/tensor_parallel.shard_llama_ splits it 8 ways for the node — SURVEY.md §8f row 1)
    if getattr(cfg, "enable_lora", False):
        raise NotImplementedError("LoRA branch is out of scope (SURVEY.md §2 row 3)")
    model = LlamaForCausalLM(lcfg, device=getattr(cfg, "device", None))
    num_embd = lcfg.hidden_size
    model = build_surrogate_model(cfg, model, num_hidden_layers=lcfg.num_hidden_layers)
    tokenizer = getattr(cfg, "tokenizer", None)
    return num_embd, model, tokenizer, LlamaDecoderLayer, is_giant_model


def build_surrogate_model(cfg, model, model_attr="model", num_hidden_layers=-1):
    """ref:build_decoder.py:268-335: per range, the first ``num_translators`` layers are
    re-classed to ``Translator`` in place and the rest popped (offsets as in the reference)."""
    model_name = getattr(cfg, "decoder_hf_model_name", "decoder")
    if cfg.enable_translator:
        print("start converting to surrogate model")
        print_model_numel(model, model_name="[orig] " + str(model_name))
        translators_range = [list(r) for r in deepcopy(cfg.translators_range)]
        for range_idx, trans_range in enumerate(translators_range):
            trans_range.sort()
            assert (np.sum(np.diff(trans_range) == 1) == len(trans_range) - 1), \
                "translators_range should be consecutive without skipping layers"
            assert np.all(np.array(trans_range) >= 0)
            assert 0 <= min(trans_range) < max(trans_range) < num_hidden_layers, \
                f"translators_range should be in the range of [0, {num_hidden_layers})"
            assert cfg.num_translators <= len(trans_range), \
                f"num_translators should be less than or equal to the number of layers in translators_range: {len(trans_range)}"
            index_offset = 0
            if range_idx > 0:
                index_offset += sum(len(translators_range[i]) for i in range(range_idx))
            for i in range(cfg.num_translators):
                translator_layer_id = trans_range[i]
                print(f"- mv layer {translator_layer_id:>3} to translator -- actual layer {translator_layer_id - index_offset:>3}")
                model = rename_layer_to_translator(model, translator_layer_id - index_offset, model_attr=model_attr)
                trans_range.pop(i)
            for layer_id in trans_range:
                print(f"- rm layer {layer_id:>3}                -- actual layer {layer_id - index_offset:>3}")
                model = remove_layer_from_model(model, layer_id - index_offset, model_attr=model_attr)
                index_offset += 1
        print_model_numel(model, model_name="[srgt] " + str(model_name))
    else:
        print_model_numel(model, model_name=str(model_name))
    return model


def rename_layer_to_translator(model: torch.nn.Module, layer_id: int, model_attr="model"):
    """Same numerics, new class name ``Translator`` sharing the layer's __dict__
    (ref:build_decoder.py:338-357)."""
    layers = getattr(model, model_attr).layers
    layer = layers[layer_id]
    Translator = type("Translator", (type(layer),), {})
    translator = Translator.__new__(Translator)
    translator.__dict__.update(layer.__dict__)
    layers[layer_id] = translator
    return model


def remove_layer_from_model(model: torch.nn.Module, layer_id: int, model_attr="model"):
    getattr(model, model_attr).layers.pop(layer_id)
    return model
