"""Subset of the reference's ConfigBase (ref:configs/config_base.py:17-230) that the hot
path reads, with the same field names and defaults, plus the public model constants the
reference fetches from the hub (not available offline)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch


@dataclass
class ConfigBase:
    # models
    decoder: str = "llama3"
    llama_model_scale: str = "3"
    llama_model_version: str = "3.2"
    encoder_model_name: str = "openai/clip-vit-large-patch14-336"
    input_img_resolution: int = 336
    input_img_patch_size: int = 14
    num_embd: int = 3072
    adapter_mlp_depth: int = 2
    enable_translator: bool = True
    translators_range: Tuple = (list(range(16, 27)),)
    num_translators: int = 1
    encoder_lr_scaler: float = 1.0
    encoder_ft_layers_after: int = 16
    encoder_shave_last_n_layers: int = -1
    ignore_encoder_cls_token: bool = False
    enable_sdpa: bool = True
    enable_lora: bool = False
    fsdp_mode: bool = False
    # training
    training_stage: str = "finetune-encoder"
    device: Optional[torch.device] = None
    dtype: str = "bfloat16"
    ptdtype: torch.dtype = torch.bfloat16
    seed: int = 1337
    batch_size: int = 8
    lr: float = 5e-5
    min_lr: float = 0.0
    wd: float = 0.0
    grad_clip: float = 1.0
    adamw_beta1: float = 0.9
    adamw_beta2: float = 0.999
    reduce_batch_loss: bool = True
    enable_balance_loss_weights: bool = False
    gradient_accumulation_steps: int = 1
    max_seq_len: int = 4096
    # prompt handling (ref:engine:75-176)
    ignore_instruction_token_ids: bool = True
    include_special_tokens_in_masking_instruction_tokens: bool = True
    enable_image_prompt: bool = True
    enable_random_image_prompt: bool = False
    split_image_tokens_in_prompt_with_newline: bool = False
    image_prompt_templates: List[str] = field(default_factory=lambda: ["Here is an image: "])
    # local model shapes (override for tests); None -> the public constants below
    encoder_config: Optional[object] = None
    decoder_config: Optional[object] = None
    hf_cache_dir: Optional[str] = None
    decoder_hf_model_name: Optional[str] = None


config = ConfigBase()

# public config.json constants, (hidden, heads, kv, ffn, layers, rope factor, tie, translators)
LLAMA_SHAPES = {
    ("3.2", "3"): dict(hidden_size=3072, num_attention_heads=24, num_key_value_heads=8, intermediate_size=8192,
                       num_hidden_layers=28, rope_factor=32.0, tie_word_embeddings=True,
                       translators_range=(list(range(16, 27)),)),
    ("3.1", "8"): dict(hidden_size=4096, num_attention_heads=32, num_key_value_heads=8, intermediate_size=14336,
                       num_hidden_layers=32, rope_factor=8.0, tie_word_embeddings=False,
                       translators_range=(list(range(17, 31)),)),
    ("3.1", "70"): dict(hidden_size=8192, num_attention_heads=64, num_key_value_heads=8, intermediate_size=28672,
                        num_hidden_layers=80, rope_factor=8.0, tie_word_embeddings=False,
                        translators_range=(list(range(40, 79)),)),
}
