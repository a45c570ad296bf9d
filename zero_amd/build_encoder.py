"""``load_clip`` with the reference's signature and return contract
(ref:build_encoder.py:11-29): ``(encoder, image_processor, image_input_size)``.

The reference calls ``CLIPVisionModel.from_pretrained(hub name)``; there is no hub access
here (and no network on the GPU box), so the tower is built from the public
clip-vit-large-patch14-336 shape constants (or ``cfg.encoder_config``) with random init;
weights are then loaded with ``load_state_dict`` from a checkpoint in the transformers-4.45
key layout (``vision_model.…``) when one is available."""
from __future__ import annotations

import torch

from .models.clip import CLIPVisionConfig, CLIPVisionModel


class ImageProcessorStub:
    """Carries only what the reference reads from CLIPImageProcessor on this path
    (ref:build_encoder.py:27): ``size["shortest_edge"]`` and the CLIP mean/std constants."""

    def __init__(self, size: int):
        self.size = {"shortest_edge": size}
        self.crop_size = {"height": size, "width": size}
        self.image_mean = [0.48145466, 0.4578275, 0.40821073]
        self.image_std = [0.26862954, 0.26130258, 0.27577711]


def load_clip(cfg: object, model_name: str = "openai/clip-vit-large-patch14-336"):
    print(f"building clip vision tower (local config, HIP kernels): {model_name}")
    ccfg = getattr(cfg, "encoder_config", None) or CLIPVisionConfig()
    device = getattr(cfg, "device", None)
    model = CLIPVisionModel(ccfg, device=device)
    if getattr(cfg, "ptdtype", torch.bfloat16) != torch.bfloat16:
        raise ValueError("zero_amd kernels compute in bf16 (ref:configs/zero/train_llama.py:80)")
    processor = ImageProcessorStub(ccfg.image_size)
    image_input_size = processor.size["shortest_edge"]
    return model, processor, image_input_size
