"""zk_attn_decode_fused_static — the one entry point that allocates nothing (its partial results live in a device
array inside libzero_hip.so) — against zk_attn_decode_fused (caller workspace), THROUGH THE C ABI (ctypes on
zero_amd._lib.lib, no ops.* wrapper), bit for bit: both run the same two kernels.  Two calls in a row, the second
with a smaller input, so stale contents of the shared array from the first call are in place when the second runs.
Collected last (file name) so that a failure here cannot hide other tests under `pytest -x`."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _case(B, Hq, Hkv, hd, rows, pos, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    W = (Hq + 2 * Hkv) * hd
    qkv = torch.randn(B, W, generator=g).to(torch.bfloat16).to(DEV)
    kc = torch.randn(B, rows, Hkv * hd, generator=g).to(torch.bfloat16).to(DEV)
    vc = torch.randn(B, rows, Hkv * hd, generator=g).to(torch.bfloat16).to(DEV)
    ang = torch.rand(rows, hd // 2, generator=g) * 6.28
    return qkv, kc, vc, ang.cos().to(DEV).contiguous(), ang.sin().to(DEV).contiguous()


def _run_both(B, Hq, Hkv, hd, rows, pos, seed):
    from zero_amd._lib import lib
    qkv, kc, vc, cos_t, sin_t = _case(B, Hq, Hkv, hd, rows, pos, seed)
    stream = torch.cuda.current_stream().cuda_stream
    scale = hd ** -0.5
    # existing entry point: caller-owned workspace
    kc_a, vc_a = kc.clone(), vc.clone()
    out_a = torch.full((B, Hq * hd), float("nan"), dtype=torch.bfloat16, device=DEV)
    need = B * Hq * (pos // 64 + 1) * (hd + 2)
    ws = torch.empty(need, dtype=torch.float32, device=DEV)
    rc = lib.zk_attn_decode_fused(qkv.data_ptr(), cos_t[pos].data_ptr(), sin_t[pos].data_ptr(), kc_a.data_ptr(),
                                  vc_a.data_ptr(), out_a.data_ptr(), ws.data_ptr(), ws.numel(), B, Hq, Hkv, hd,
                                  qkv.stride(0), out_a.stride(0), rows, pos, scale, stream)
    assert rc == 0
    # the entry point under test: no workspace argument at all
    kc_b, vc_b = kc.clone(), vc.clone()
    out_b = torch.full((B, Hq * hd), float("nan"), dtype=torch.bfloat16, device=DEV)
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    rc = lib.zk_attn_decode_fused_static(qkv.data_ptr(), cos_t[pos].data_ptr(), sin_t[pos].data_ptr(), kc_b.data_ptr(),
                                         vc_b.data_ptr(), out_b.data_ptr(), B, Hq, Hkv, hd, qkv.stride(0),
                                         out_b.stride(0), rows, pos, scale, stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert not torch.isnan(out_b.float()).any()
    assert torch.equal(out_a, out_b)
    assert torch.equal(kc_a, kc_b) and torch.equal(vc_a, vc_b)
    return out_b


def test_static_scratch_entry_matches_workspace_entry_twice_second_smaller():
    # call 1: 24 q heads / 8 kv heads x 128 (Llama-3.2-3B), 2 sequences, 17 chunks of cached keys
    _run_both(2, 24, 8, 128, 1100, 1027, seed=5)
    # call 2, smaller in every dimension (fewer heads, head_dim 64, one chunk): the array still holds call 1's partials
    _run_both(1, 6, 6, 64, 70, 40, seed=6)
    # and a third at the first shape again (a larger call after a smaller one)
    _run_both(2, 24, 8, 128, 1100, 1027, seed=7)


def test_static_scratch_entry_rejects_oversize():
    from zero_amd._lib import lib
    B, Hq, Hkv, hd = 4, 64, 8, 128
    pos = 70000                       # 4 * 64 * 1094 chunks * 130 floats = 36 M floats > the 2 M-float array
    rows = pos + 1
    qkv, kc, vc, cos_t, sin_t = _case(B, Hq, Hkv, hd, 8, 0, seed=8)      # tiny buffers: nothing may be launched
    out = torch.zeros((B, Hq * hd), dtype=torch.bfloat16, device=DEV)
    rc = lib.zk_attn_decode_fused_static(qkv.data_ptr(), cos_t[0].data_ptr(), sin_t[0].data_ptr(), kc.data_ptr(),
                                         vc.data_ptr(), out.data_ptr(), B, Hq, Hkv, hd, qkv.stride(0), out.stride(0),
                                         rows, pos, hd ** -0.5, torch.cuda.current_stream().cuda_stream)
    assert rc == -7
    torch.cuda.synchronize()
    assert torch.count_nonzero(out) == 0


def test_static_scratch_entry_reports_bad_arguments_before_the_size_limit():
    """An invalid AND oversized call is an invalid call: -22 (null pointer, Hq not a multiple of Hkv, short row
    stride), not -7; nothing is launched either way."""
    from zero_amd._lib import lib
    B, Hq, Hkv, hd = 4, 64, 8, 128
    pos = 70000
    qkv, kc, vc, cos_t, sin_t = _case(B, Hq, Hkv, hd, 8, 0, seed=9)
    out = torch.zeros((B, Hq * hd), dtype=torch.bfloat16, device=DEV)
    st = torch.cuda.current_stream().cuda_stream

    def call(qkv_p=qkv.data_ptr(), hkv=Hkv, ldqkv=qkv.stride(0)):
        return lib.zk_attn_decode_fused_static(qkv_p, cos_t[0].data_ptr(), sin_t[0].data_ptr(), kc.data_ptr(),
                                               vc.data_ptr(), out.data_ptr(), B, Hq, hkv, hd, ldqkv, out.stride(0),
                                               pos + 1, pos, hd ** -0.5, st)
    assert call() == -7
    assert call(qkv_p=None) == -22
    assert call(hkv=7) == -22
    assert call(ldqkv=8) == -22
    torch.cuda.synchronize()
    assert torch.count_nonzero(out) == 0
