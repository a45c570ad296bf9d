"""``load_hf_llama3`` / ``build_surrogate_model`` with the reference's signatures
(ref:build_decoder.py:17-146,268-362).

Returns ``(num_embd, model, tokenizer, FSDP_DECODER_LAYER, is_giant_model)``.  The decoder is
built from the public Llama-3 shape constants (or ``cfg.decoder_config``) because the gated
hub weights and the tokenizer cannot be fetched offline; ``tokenizer`` is None unless
``cfg.tokenizer`` supplies one."""
from __future__ import annotations

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


def surgery_plan(num_hidden_layers: int, translators_range, num_translators: int):
    """What ref:build_decoder.py:283-330 does to the layer list, as two sets of ORIGINAL layer indices:
    (layers re-classed to ``Translator``, layers removed).

    Per range (sorted, consecutive, inside [0, num_hidden_layers)): the reference walks ``i`` over
    ``range(num_translators)``, re-classes entry ``i`` of the range list and deletes that entry from the list
    before taking the next one, so the layers it keeps as translators are entries 0, 2, 4, ... of the range;
    everything else in the range is popped from the model."""
    translators, removed = [], []
    last = -1
    for rng in translators_range:
        rng = sorted(int(v) for v in rng)
        if any(b - a != 1 for a, b in zip(rng, rng[1:])):
            raise AssertionError("translators_range should be consecutive without skipping layers")
        if not (0 <= rng[0] < rng[-1] < num_hidden_layers):
            raise AssertionError(f"translators_range should be in the range of [0, {num_hidden_layers})")
        if num_translators > len(rng):
            raise AssertionError("num_translators should be less than or equal to the number of layers in "
                                 f"translators_range: {len(rng)}")
        if rng[0] <= last:
            raise AssertionError("translators_range entries must be disjoint and in ascending order")
        last = rng[-1]
        picks = [2 * i for i in range(num_translators)]
        if picks and picks[-1] >= len(rng):
            raise IndexError("list index out of range")      # what the reference's shrinking-list walk raises
        kept = {rng[j] for j in picks}
        translators += sorted(kept)
        removed += [v for v in rng if v not in kept]
    return translators, removed


def build_surrogate_model(cfg, model, model_attr="model", num_hidden_layers=-1):
    """Truncate ``model`` to the surrogate (ref:build_decoder.py:268-335; 28→18, 32→19, 80→42 layers for the
    shipped ranges): translator layers keep their weights and position but read as class ``Translator``, the other
    layers of each range leave the ``ModuleList``.  ``config.num_hidden_layers`` is left alone, as there."""
    name = str(getattr(cfg, "decoder_hf_model_name", "decoder"))
    if not cfg.enable_translator:
        print_model_numel(model, model_name=name)
        return model
    print_model_numel(model, model_name="[orig] " + name)
    translators, removed = surgery_plan(num_hidden_layers, cfg.translators_range, cfg.num_translators)
    gone = 0                                   # layers already removed below the one being handled
    todo = sorted([(i, "mv") for i in translators] + [(i, "rm") for i in removed])
    for orig, what in todo:
        slot = orig - gone
        if what == "mv":
            print(f"surrogate: layer {orig} (slot {slot}) -> Translator")
            rename_layer_to_translator(model, slot, model_attr=model_attr)
        else:
            print(f"surrogate: layer {orig} (slot {slot}) removed")
            remove_layer_from_model(model, slot, model_attr=model_attr)
            gone += 1
    print_model_numel(model, model_name="[srgt] " + name)
    return model


def rename_layer_to_translator(model: torch.nn.Module, layer_id: int, model_attr="model"):
    """The layer in slot ``layer_id`` becomes an instance of a fresh subclass named ``Translator``
    (ref:build_decoder.py:338-357): same parameters, buffers and forward, only ``type(layer).__name__`` —
    what ``func_optim_filter`` and class-based wrapping policies look at — changes."""
    layers = getattr(model, model_attr).layers
    layer = layers[layer_id]
    layer.__class__ = type("Translator", (type(layer),), {})
    return model


def remove_layer_from_model(model: torch.nn.Module, layer_id: int, model_attr="model"):
    """ref:build_decoder.py:360-362."""
    layers = getattr(model, model_attr).layers
    del layers[layer_id]
    return model
