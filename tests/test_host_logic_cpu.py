"""Host-side logic (no GPU): the index plan of the input assembly against the oracle, the
surrogate surgery, reduce_sum semantics, the tokenisation/masking rules with a fake tokenizer."""
import numpy as np
import pytest
import torch

from oracle import zero_oracle as zo
from tests.golden.configs import GOLDEN_CONFIGS
from zero_amd.methods.engine_ar_instruction_llama import IGNORE_INDEX, build_plan, tokenize_conversations


def _emulate_gather(plan, wte, imgtab):
    codes = torch.from_numpy(plan["codes"]).long()
    out = torch.zeros(codes.numel(), wte.shape[1])
    tok = codes >= 0
    out[tok] = wte[codes[tok]]
    img = codes <= -2
    out[img] = imgtab[-2 - codes[img]]
    return out.view(plan["B"], plan["S"], -1)


@pytest.mark.parametrize("name", ["tiny", "trunc"])
def test_build_plan_matches_oracle_assembly(name):
    spec = GOLDEN_CONFIGS[name]
    enc, dec = spec["enc"], spec["dec"]
    imgs, ids, tgts, img_id, pad_id = zo.synth_batch(enc, dec, **spec["batch"])
    B, T, d = len(imgs), enc.tokens, dec.hidden
    g = torch.Generator().manual_seed(0)
    wte = torch.randn(dec.vocab, d, generator=g)
    rows = torch.randn(B * T, d, generator=g)
    soi, eoi = torch.randn(1, d, generator=g), torch.randn(1, d, generator=g)
    embds = [torch.cat([soi, rows[b * T:(b + 1) * T], eoi]) for b in range(B)]
    x_ref, y_ref = zo.assemble_inputs({"decoder.model.embed_tokens.weight": wte}, embds, ids, tgts, img_id,
                                      spec["max_seq_len"])
    plan = build_plan(ids, tgts, [T] * B, img_id, pad_id, spec["max_seq_len"])
    x = _emulate_gather(plan, wte, torch.cat([rows, soi, eoi]))
    assert torch.equal(x, x_ref)
    assert torch.equal(torch.from_numpy(plan["target"]), y_ref)
    # labels / valid rows against the oracle's loss tail conventions
    tgt = y_ref[:, 1:].clone()
    tgt[tgt == pad_id] = IGNORE_INDEX
    assert plan["n_valid"] == int(tgt.ne(IGNORE_INDEX).sum())
    S = plan["S"]
    for r, lab in zip(plan["rows"], plan["row_labels"]):
        b, s = divmod(int(r), S)
        assert y_ref[b, s + 1] == lab
    # inverse maps
    fc = plan["codes"]
    for r, pos in enumerate(plan["inv_codes"]):
        if pos >= 0:
            assert fc[pos] == -2 - r
    used = set(-2 - c for c in fc if -2 - B * T < c <= -2)
    assert used == {r for r, pos in enumerate(plan["inv_codes"]) if pos >= 0}


def test_surgery_on_modules_matches_oracle_map():
    import types
    from zero_amd.build_decoder import build_surrogate_model

    class L(torch.nn.Module):
        def __init__(self, i):
            super().__init__()
            self.idx = i
    from zero_amd.build_decoder import surgery_plan
    # nt = 2: the reference's shrinking-list walk (build_decoder.py:305-316) keeps entries 0 and 2 of each range
    assert surgery_plan(12, ([2, 3, 4, 5], [8, 9, 10]), 2) == ([2, 4, 8, 10], [3, 5, 9])
    for n, ranges, nt in ((28, (list(range(16, 27)),), 1), (12, ([2, 3, 4], [7, 8, 9]), 1), (32, (list(range(17, 31)),), 1),
                          (80, (list(range(40, 79)),), 1), (12, ([2, 3, 4, 5], [8, 9, 10]), 2)):
        model = types.SimpleNamespace(model=types.SimpleNamespace(layers=torch.nn.ModuleList([L(i) for i in range(n)])))
        cfg = types.SimpleNamespace(enable_translator=True, translators_range=ranges, num_translators=nt,
                                    decoder_hf_model_name="t")
        model.parameters = lambda: []
        build_surrogate_model(cfg, model, num_hidden_layers=n)
        kept = [l.idx for l in model.model.layers]
        assert kept == zo.surrogate_layer_map(n, ranges, nt)
        tr, _ = surgery_plan(n, ranges, nt)
        for l in model.model.layers:
            assert (type(l).__name__ == "Translator") == (l.idx in tr) and isinstance(l, L)


def test_reduce_sum_identity_backward_single_rank():
    from zero_amd.distributed.torchshard import reduce_sum, reduce_sum_pair
    x = torch.tensor(3.0, requires_grad=True)
    y = reduce_sum(x * 2)
    (y / 4).backward()
    assert y.item() == 6.0 and x.grad.item() == 0.5
    a, b = reduce_sum_pair(torch.tensor(1.5), torch.tensor(7))
    assert a.item() == 1.5 and b.item() == 7


class FakeTokenizer:
    """Word-level stand-in with the three calls the engine makes."""
    SPECIAL = ["<|begin_of_text|>", "<|end_of_text|>", "<|eot_id|>", "<|start_header_id|>", "<|end_header_id|>",
               "<|reserved_special_token_0|>"]

    def __init__(self):
        self.vocab = {s: 1000 + i for i, s in enumerate(self.SPECIAL)}
        self.chat_template = None

    def convert_tokens_to_ids(self, t):
        return self.vocab[t]

    def _enc(self, text):
        out = []
        for w in text.replace("<|reserved_special_token_0|>", " <|reserved_special_token_0|> ").split():
            out.append(self.vocab.setdefault(w, len(self.vocab) + 10) if w not in self.vocab else self.vocab[w])
        return out

    def apply_chat_template(self, msgs):
        m = msgs[0]
        return ([self.vocab["<|start_header_id|>"]] + self._enc(m["role"]) + [self.vocab["<|end_header_id|>"]]
                + self._enc(m["content"]) + [self.vocab["<|eot_id|>"]])


def test_tokenize_masking_rules():
    from zero_amd.configs.config_base import ConfigBase
    cfg = ConfigBase()
    tok = FakeTokenizer()
    conv = [[{"role": "user", "content": "<image>\nWhat is shown here"}, {"role": "assistant", "content": "a shark"}]]
    ids, tgts, img_id, pad_id = tokenize_conversations(cfg, conv, tok)
    i, t = ids[0], tgts[0]
    assert len(i) == len(t) and i[0] == tok.vocab["<|begin_of_text|>"] and t[0] == IGNORE_INDEX
    assert i.count(img_id) == 1
    # user turn: only the three special tokens are unmasked (engine:144-150)
    sh, eh, eot = (tok.vocab[k] for k in ("<|start_header_id|>", "<|end_header_id|>", "<|eot_id|>"))
    first_eot = i.index(eot)
    user_t = t[1:first_eot + 1]
    assert [x for x in user_t if x != IGNORE_INDEX] == [sh, eh, eot]
    # assistant turn: header role masked, the rest is the target (engine:168-173)
    ast_i, ast_t = i[first_eot + 1:], t[first_eot + 1:]
    assert ast_t[0] == sh and ast_t[1] == IGNORE_INDEX and ast_t[2:] == ast_i[2:]
    cfg.include_special_tokens_in_masking_instruction_tokens = False
    ids2, tgts2, _, _ = tokenize_conversations(cfg, conv, tok)
    assert tgts2[0][:first_eot + 1] == [IGNORE_INDEX] * (first_eot + 1)
    assert tgts2[0][first_eot + 1:first_eot + 4] == [IGNORE_INDEX] * 3


def test_flat_reducer_layout_cpu():
    """Packed q|k|v stay adjacent in the flat buffers; slots receive and accumulate gradients."""
    from zero_amd.distributed.ddp import FlatGradReducer
    from zero_amd.functional import adjacent
    from zero_amd.models.layers import Linear, pack_params

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q, self.k, self.v = Linear(8, 8), Linear(8, 8), Linear(8, 8)
            pack_params([self.q.weight, self.k.weight, self.v.weight])
            self.o = Linear(8, 16)
    m = M()
    ref = {n: p.detach().clone() for n, p in m.named_parameters()}
    r = FlatGradReducer(m, bucket_bytes=64)
    try:
        assert adjacent(m.q.weight, m.k.weight, m.v.weight)
        assert adjacent(m.q.weight._zk_grad, m.k.weight._zk_grad, m.v.weight._zk_grad)
        for n, p in m.named_parameters():
            assert torch.equal(p, ref[n]) and p._zk_grad.numel() == p.numel()
        assert len(r.buckets) > 1 and sum(len(b["params"]) for b in r.buckets) == len(r.params)
        assert r.buckets[0]["start"] == 0 and r.buckets[-1]["end"] == r.numel
    finally:
        r.close()


def test_nan_watch_raises_like_the_reference_on_cpu():
    """engine:315-316: a NaN loss is a ValueError (on the CPU path immediately; on the GPU the flag is
    read back asynchronously, see _NanWatch)."""
    from zero_amd.methods.engine_ar_instruction_llama import _NanWatch
    w = _NanWatch()
    w.check_and_arm(torch.tensor(1.0), torch.device("cpu"))
    with pytest.raises(ValueError, match="NaN"):
        w.check_and_arm(torch.tensor(float("nan")), torch.device("cpu"))
