"""Run-config module in the reference's plugin format (ref:configs/zero/train_llama.py): exports
``config`` and the five ``func_*`` factories that ref:train.py:63-67 consumes.  Stage:
finetune-encoder on the Llama-3.2-3B surrogate (the north-star configuration)."""
import copy

from ..config_base import LLAMA_SHAPES, config

cfg = copy.deepcopy(config)
TRAINING_STAGE = "finetune-encoder"
cfg.training_stage = TRAINING_STAGE
cfg.llama_model_scale = "3"
cfg.llama_model_version = "3.2"
cfg.enable_translator = True
cfg.translators_range = LLAMA_SHAPES[(cfg.llama_model_version, cfg.llama_model_scale)]["translators_range"]
cfg.batch_size = 8
cfg.lr = 5e-5
cfg.grad_clip = 1.0
cfg.max_seq_len = 4096
cfg.encoder_ft_layers_after = 16
config = cfg


def func_load_loader():
    raise NotImplementedError("data loading is outside the hot path (SURVEY.md §2 row 9); feed synthetic batches")


def func_load_decoder():
    from ... import build_decoder
    return build_decoder.load_hf_llama3


def func_load_encoder():
    from ... import build_encoder
    return build_encoder.load_clip


def func_load_forward():
    from ...methods import engine_ar_instruction_llama as method
    return method.forward_hf


def translator_layer_ids(cfg):
    """Indices of the translator layers AFTER the surgery (ref:configs/zero/train_llama.py:290-301,
    restated with its own offset rule: previous ranges shrink to one surviving layer each)."""
    ids = []
    for i, r in enumerate(cfg.translators_range):
        o = 0
        if i > 0:
            o += sum(len(cfg.translators_range[j]) for j in range(i)) - 1
        ids += [n - o for n in list(r)[:cfg.num_translators]]
    return ids


def func_optim_filter(cfg, model, verbose: bool = False):
    """Trainable set and AdamW groups by parameter-name substring, stages finetune-encoder and
    pretrain (ref:configs/zero/train_llama.py:246-380)."""
    decay_params, other_params, encoder_params = [], [], []
    for n, p in model.named_parameters():
        p.requires_grad = False
        if "adapter" in n:
            p.requires_grad = True
            (decay_params if p.ndim >= 2 else other_params).append(p)
        if "embd_soi" in n or "embd_eoi" in n or "special_tokens" in n:
            p.requires_grad = True
            (decay_params if p.ndim >= 2 else other_params).append(p)
        if cfg.training_stage == "finetune-encoder":
            if "vision_model" in n:
                if "post_layernorm" in n:
                    p.requires_grad = True
                    other_params.append(p)
                if "layers" in n:
                    parts = n.split(".")
                    layer_id = int(parts[parts.index("layers") + 1])
                    if layer_id >= cfg.encoder_ft_layers_after:
                        p.requires_grad = True
                        encoder_params.append(p)
        elif cfg.training_stage == "pretrain":
            # stage 1: adapter (+ soi/eoi, above), post_layernorm and the translator layer(s) of the
            # surrogate (ref:configs/zero/train_llama.py:281-307)
            if "vision_model" in n and "post_layernorm" in n:
                p.requires_grad = True
                encoder_params.append(p)
            if "decoder" in n and "layers" in n:
                parts = n.split(".")
                layer_id = int(parts[parts.index("layers") + 1])
                if layer_id in translator_layer_ids(cfg):
                    p.requires_grad = True
                    (decay_params if p.ndim >= 2 else other_params).append(p)
        elif cfg.training_stage == "finetune-decoder":
            raise NotImplementedError("stage finetune-decoder: training the whole decoder is outside the built "
                                      "paths (SURVEY.md §8f)")
        if verbose:
            print(f"p.requires_grad: {str(p.requires_grad):<5}, param: {n}, {tuple(p.shape)}")
    groups = [
        {"params": decay_params, "lr": cfg.lr, "weight_decay": cfg.wd},
        {"params": other_params, "lr": cfg.lr, "weight_decay": 0.0},
    ]
    if encoder_params:
        groups.append({"params": encoder_params, "lr": cfg.lr * cfg.encoder_lr_scaler, "weight_decay": 0.0})
    return groups
