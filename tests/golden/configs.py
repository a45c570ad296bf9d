"""Shapes of the golden fixtures (shared by make_golden.py and the tests)."""
from oracle.zero_oracle import DecoderCfg, EncoderCfg

GOLDEN_CONFIGS = {
    # reference hard-codes the encoder width 1024 (adapter.py:34), so even the tiny case keeps it
    "tiny": dict(
        enc=EncoderCfg(hidden=1024, heads=16, intermediate=512, layers=3, image=42, patch=14),
        dec=DecoderCfg(hidden=256, heads=2, kv_heads=1, head_dim=128, intermediate=512, layers=6,
                       vocab=512, tie=True, translators_range=((2, 3, 4),), num_translators=1),
        seed=1337, ft_after=1, max_seq_len=4096,
        batch=dict(B=3, seed=11, n_pre=5, n_post=24, n_resp=12, ragged=True, text_only=(1,)),
    ),
    # truncation (engine:198,239-240), two translator ranges, untied head, 3 q-heads per kv-head
    "trunc": dict(
        enc=EncoderCfg(hidden=1024, heads=16, intermediate=256, layers=2, image=28, patch=14),
        dec=DecoderCfg(hidden=384, heads=3, kv_heads=1, head_dim=128, intermediate=256, layers=9,
                       vocab=640, tie=False, translators_range=((1, 2), (5, 6, 7)), num_translators=1,
                       rope_factor=8.0),
        seed=4242, ft_after=0, max_seq_len=24,
        batch=dict(B=2, seed=5, n_pre=4, n_post=30, n_resp=20, ragged=False),
    ),
    # real-width layers: CLIP-L/14-336 width (577 tokens) and Llama-3.2-3B width, few layers
    "mid": dict(
        enc=EncoderCfg(hidden=1024, heads=16, intermediate=4096, layers=2, image=336, patch=14),
        dec=DecoderCfg(hidden=3072, heads=24, kv_heads=8, head_dim=128, intermediate=8192, layers=3,
                       vocab=4096, tie=True, translators_range=((1, 2),), num_translators=1),
        seed=2024, ft_after=1, max_seq_len=4096,
        batch=dict(B=2, seed=3, n_pre=21, n_post=40, n_resp=24, ragged=True),
    ),
    # stage "pretrain" (ref:configs/zero/train_llama.py:281-307): adapter + post_layernorm + the
    # translator decoder layers train; two translator ranges so the id offset rule is exercised
    "pretrain": dict(
        enc=EncoderCfg(hidden=1024, heads=16, intermediate=256, layers=2, image=28, patch=14),
        dec=DecoderCfg(hidden=256, heads=4, kv_heads=2, head_dim=128, intermediate=384, layers=8,
                       vocab=512, tie=True, translators_range=((1, 2), (4, 5, 6)), num_translators=1),
        seed=777, ft_after=0, max_seq_len=4096, stage="pretrain",
        batch=dict(B=2, seed=9, n_pre=5, n_post=28, n_resp=14, ragged=True),
    ),
    # zero-shot graft inference (ref:inference/vlm_infer.py): FULL decoder (no surgery), greedy decoding
    "graft": dict(
        enc=EncoderCfg(hidden=1024, heads=16, intermediate=256, layers=2, image=28, patch=14),
        dec=DecoderCfg(hidden=256, heads=4, kv_heads=2, head_dim=128, intermediate=512, layers=4,
                       vocab=512, tie=False, translators_range=(), num_translators=1),
        seed=31337, ft_after=0, max_seq_len=4096, graft=True, max_new_tokens=12,
        batch=dict(B=2, seed=17, n_pre=6, n_post=9, n_resp=1, ragged=False),
    ),
}


def shrink(t):
    """Fixture size control: big activations are stored as a fixed sub-sample
    (every 5th position, first 48 features); the tests apply the same cut."""
    if t.ndim == 3 and t.numel() > 300000:
        return t[:, ::5, :48].contiguous()
    if t.ndim == 2 and t.numel() > 65536:
        return t[:48, :96].contiguous()
    return t.contiguous()
