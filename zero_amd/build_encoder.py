"""``load_clip`` with the reference's signature and return contract
(ref:build_encoder.py:11-29): ``(encoder, image_processor, image_input_size)``.

The reference calls ``CLIPVisionModel.from_pretrained(hub name)``; there is no hub access
here (and no network on the GPU box), so the tower is built from the public
clip-vit-large-patch14-336 shape constants (or ``cfg.encoder_config``) with random init;
weights are then loaded with ``load_state_dict`` from a checkpoint in the transformers-4.45
key layout (``vision_model.…``) when one is available."""
from __future__ import annotations

import torch

from .loader import CLIPImageProcessorHip
from .models.clip import CLIPVisionConfig, CLIPVisionModel


def load_clip(cfg: object, model_name: str = "openai/clip-vit-large-patch14-336"):
    print(f"building clip vision tower (local config, HIP kernels): {model_name}")
    ccfg = getattr(cfg, "encoder_config", None) or CLIPVisionConfig()
    device = getattr(cfg, "device", None)
    model = CLIPVisionModel(ccfg, device=device)
    if getattr(cfg, "ptdtype", torch.bfloat16) != torch.bfloat16:
        raise ValueError("zero_amd kernels compute in bf16 (ref:configs/zero/train_llama.py:80)")
    # the reference loads CLIPImageProcessor.from_pretrained (ref:build_encoder.py:24); here the same pipeline
    # with the public CLIP mean/std constants, running on the device (zero_amd/loader.py)
    processor = CLIPImageProcessorHip(ccfg.image_size, device=device)
    image_input_size = processor.size["shortest_edge"]
    return model, processor, image_input_size
