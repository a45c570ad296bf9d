"""Per-kernel parity: each C-ABI entry point against a plain PyTorch fp32 reference of the
same op, on the same bf16-rounded inputs.  Tolerances are stated per test: outputs are
bf16 (eps = 2^-8 = 3.9e-3), accumulation is fp32."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from zero_amd import ops

DEV = "cuda"


def relerr(a, b):
    a = a.float(); b = b.float()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(DEV)


# ------------------------------------------------------------------ GEMM
@pytest.fixture(params=[-1, 0, 1, 2, 3, 4], ids=["auto", "t128", "t256x128", "t256", "t128x96", "t384x192"])
def gemm_tile(request):
    """Every GEMM test runs under the cost model and under each forced block tile."""
    ops.set_gemm_tile(request.param)
    yield request.param
    ops.set_gemm_tile(-1)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (577 * 3, 1024, 640), (130, 72, 40), (64, 3072, 1024),
                                   (800, 400, 128), (1000, 776, 264), (390, 200, 328)])
def test_gemm_nt(M, N, K, gemm_tile):
    a, b = rnd(M, K, seed=1), rnd(N, K, seed=2)
    c = ops.gemm(a, b)
    ref = a.float() @ b.float().t()
    assert relerr(c, ref) < 6e-3


def test_gemm_layout_exact_integers(gemm_tile):
    # A = identity-like probe with an ASYMMETRIC B catches any row/col swap or k permutation
    M = N = K = 128
    a = torch.eye(M, K, device=DEV).to(torch.bfloat16)
    b = (torch.arange(N * K, device=DEV).reshape(N, K) % 251 - 125).to(torch.bfloat16)
    c = ops.gemm(a, b)
    assert torch.equal(c.float(), b.float().t())
    a2 = (torch.arange(M * K, device=DEV).reshape(M, K) % 13 - 6).to(torch.bfloat16)
    b2 = (torch.arange(N * K, device=DEV).reshape(N, K) % 7 - 3).to(torch.bfloat16)
    ref = a2.float() @ b2.float().t()
    got = ops.gemm(a2, b2).float()
    assert torch.equal(got, ref.to(torch.bfloat16).float())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 264, 136), (1154, 1024, 4096)])
def test_gemm_transb_dgrad(M, N, K, gemm_tile):
    # dx[M,N] = dy[M,K] @ W[K,N]  (W stored [out=K, in=N]: n contiguous)
    dy, w = rnd(M, K, seed=3), rnd(K, N, seed=4)
    c = ops.gemm(dy, w, trans_b=True)
    assert relerr(c, dy.float() @ w.float()) < 6e-3


@pytest.mark.parametrize("T,No,Ki", [(128, 128, 128), (577 * 2, 264, 136), (1000, 1024, 1024)])
def test_gemm_wgrad(T, No, Ki, gemm_tile):
    # dW[No,Ki] = dY[T,No]^T @ X[T,Ki]
    dy, x = rnd(T, No, seed=5), rnd(T, Ki, seed=6)
    c = ops.gemm(dy, x, trans_a=True, trans_b=True)
    assert relerr(c, dy.float().t() @ x.float()) < 6e-3
    c2 = ops.gemm(dy, x, trans_a=True, trans_b=True, out=c.clone(), accumulate=True)
    assert relerr(c2, 2 * (dy.float().t() @ x.float())) < 1e-2


def test_gemm_transa_only(gemm_tile):
    a, b = rnd(200, 136, seed=7), rnd(72, 200, seed=8)   # A stored [K=200][M=136], B [N=72][K=200]
    c = ops.gemm(a, b, trans_a=True)
    assert relerr(c, a.float().t() @ b.float().t()) < 6e-3


def test_gemm_strided_views(gemm_tile):
    buf = rnd(300, 512, seed=9)
    a = buf[:, 128:384]                       # [300,256] view, lda 512
    w = rnd(264, 256, seed=10)
    outbuf = torch.zeros(300, 1024, dtype=torch.bfloat16, device=DEV)
    out = outbuf[:, 512:776]
    ops.gemm(a, w, out=out)
    assert relerr(out, a.float() @ w.float().t()) < 6e-3
    assert outbuf[:, :512].abs().max() == 0 and outbuf[:, 776:].abs().max() == 0


def test_gemm_epilogues(gemm_tile):
    M, N, K = 300, 264, 136
    a, w, bias, res = rnd(M, K, seed=11), rnd(N, K, seed=12, scale=0.1), rnd(N, seed=13), rnd(M, N, seed=14)
    pre = a.float() @ w.float().t() + bias.float()
    assert relerr(ops.gemm(a, w, bias=bias), pre) < 6e-3
    pre_b = pre.to(torch.bfloat16).float()
    aux = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    got = ops.gemm(a, w, bias=bias, epilogue=ops.EPI_QUICK_GELU, aux_out=aux)
    assert relerr(aux, pre) < 6e-3
    assert relerr(got, pre_b * torch.sigmoid(1.702 * pre_b)) < 8e-3
    got = ops.gemm(a, w, bias=bias, epilogue=ops.EPI_GELU_ERF)
    assert relerr(got, F.gelu(pre_b)) < 8e-3
    got = ops.gemm(a, w, bias=bias, residual=res)
    assert relerr(got, pre_b + res.float()) < 6e-3
    alpha_dev = torch.tensor([0.25], device=DEV)
    got = ops.gemm(a, w, alpha=2.0, alpha_dev=alpha_dev)
    assert relerr(got, 0.5 * (a.float() @ w.float().t())) < 6e-3
    # backward-fused epilogues: dy W  *  act'(pre)
    h = rnd(M, N, seed=15)
    dyy, w2 = rnd(M, K, seed=16), rnd(K, N, seed=17, scale=0.1)
    base = dyy.float() @ w2.float()
    hf = h.float().requires_grad_(True)
    (hf * torch.sigmoid(1.702 * hf)).sum().backward()
    got = ops.gemm(dyy, w2, trans_b=True, epilogue=ops.EPI_MUL_DQUICK_GELU, aux_in=h)
    assert relerr(got, base * hf.grad) < 8e-3
    hf = h.float().requires_grad_(True)
    F.gelu(hf).sum().backward()
    got = ops.gemm(dyy, w2, trans_b=True, epilogue=ops.EPI_MUL_DGELU_ERF, aux_in=h)
    assert relerr(got, base * hf.grad) < 8e-3


@pytest.mark.parametrize("M,F,K", [(300, 96, 136), (512, 256, 192), (130, 32, 64)])
def test_gemm_fused_swiglu_matches_unfused(M, F, K, gemm_tile):
    """ZK_EPI_SWIGLU / ZK_EPI_SWIGLU_BWD against the separate kernels: bit-identical (same roundings)."""
    x, gw, uw = rnd(M, K, seed=41), rnd(F, K, seed=42, scale=0.2), rnd(F, K, seed=43, scale=0.2)
    wgu = torch.cat((gw, uw), 0)
    gu_ref = ops.gemm(x, wgu)
    h_ref = ops.swiglu_fwd(gu_ref, F)
    wi = ops.interleave_gate_up(gw, uw)
    gu_i, h = ops.gemm_swiglu(x, wi)
    blk = gu_i.view(M, F // 32, 2, 32)
    assert torch.equal(blk[:, :, 0].reshape(M, F), gu_ref[:, :F]) and torch.equal(blk[:, :, 1].reshape(M, F), gu_ref[:, F:])
    assert torch.equal(h, h_ref)
    # no-grad forward: the gate|up copy is neither allocated nor written (C = NULL), h is the same
    none, h2 = ops.gemm_swiglu(x, wi, keep_gu=False)
    assert none is None and torch.equal(h2, h_ref)
    # backward: dh = dy @ Wd (Wd [D, F]) then SwiGLU'
    D = 72
    dy, wd = rnd(M, D, seed=44), rnd(D, F, seed=45, scale=0.2)
    dgu_ref = ops.swiglu_bwd(ops.gemm(dy, wd, trans_b=True), gu_ref, F)
    dgu_i = ops.gemm_swiglu_bwd(dy, wd.t().contiguous(), gu_i)
    blk = dgu_i.view(M, F // 32, 2, 32)
    assert torch.equal(blk[:, :, 0].reshape(M, F), dgu_ref[:, :F]) and torch.equal(blk[:, :, 1].reshape(M, F), dgu_ref[:, F:])
    # the dgrad that follows contracts over the interleaved order on both operands
    dx_ref = ops.gemm(dgu_ref, wgu, trans_b=True)
    dx = ops.gemm(dgu_i, wi.t().contiguous())
    assert relerr(dx, dx_ref.float()) < 6e-3


# (M, N, K): > 256 tiles so a persistent workgroup walks several tiles; ragged and exact edges; K >= 256
PRING_SHAPES = [(4616, 4096, 320), (6144, 5120, 256), (4616, 3072, 1024), (5000, 4104, 264), (12288, 6144, 512)]


@pytest.mark.parametrize("M,N,K", PRING_SHAPES)
@pytest.mark.parametrize("tile", [2, 4], ids=["t256", "t384x192"])
def test_gemm_persistent_ring(M, N, K, tile):
    """Persistent slot-ring kernel (several tiles per workgroup, DMA stream across tile boundaries)
    against the fp32 product, and BIT-equal to the one-workgroup-per-tile ring kernel."""
    a, b = rnd(M, K, seed=51), rnd(N, K, seed=52, scale=0.5)
    bias, res = rnd(N, seed=53), rnd(M, N, seed=54)
    ops.set_gemm_tile(tile)
    try:
        outs = {}
        for pr in (51, 50):
            ops.set_gemm_tile(pr)
            c0 = ops.gemm(a, b)
            c1 = ops.gemm(a, b, bias=bias, residual=res)
            aux = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            c2 = ops.gemm(a, b, bias=bias, epilogue=ops.EPI_QUICK_GELU, aux_out=aux)
            c3 = ops.gemm(a, b, epilogue=ops.EPI_MUL_DGELU_ERF, aux_in=res, alpha=0.5)
            c4 = ops.gemm(a, b, out=res.clone(), accumulate=True)
            aux5 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            c5 = ops.gemm(a, b, bias=bias, epilogue=ops.EPI_GELU_ERF, aux_out=aux5)      # the adapter's epilogue pair
            outs[pr] = (c0, c1, aux, c2, c3, c4, aux5, c5)
        ref = a.float() @ b.float().t()
        assert relerr(outs[51][0], ref) < 6e-3
        for x, y in zip(outs[51], outs[50]):
            assert torch.equal(x, y)
    finally:
        ops.set_gemm_tile(51)
        ops.set_gemm_tile(-1)


@pytest.mark.parametrize("M,N,K", [(6144, 4096, 512), (1536, 2048, 320), (6144, 14336, 256), (4616, 4096, 1024),
                                   (5000, 4104, 256)])
def test_gemm_persistent_ring_192x256(M, N, K):
    """The 192x256 tile of the persistent ring (two K-tiles of 7 x 64-row slots: a wave's rows span slots):
    exact and ragged shapes (K a multiple of 64), BIT-equal to the 256x256 ring for every epilogue class, repeatable."""
    a, b = rnd(M, K, seed=57), rnd(N, K, seed=58, scale=0.5)
    bias, res = rnd(N, seed=59), rnd(M, N, seed=60)
    swiglu = N % 64 == 0
    F = N // 2
    wd_t, gu = rnd(F, K, seed=64, scale=0.2), rnd(M, 2 * F, seed=65)
    try:
        outs = {}
        for tile in (5, 2):
            ops.set_gemm_tile(tile)
            c0 = ops.gemm(a, b)
            assert ops.lib.zk_gemm_last_plan() // 100 == 30 + tile       # persistent ring on the requested tile
            c1 = ops.gemm(a, b, bias=bias, residual=res)
            aux = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            c2 = ops.gemm(a, b, bias=bias, epilogue=ops.EPI_QUICK_GELU, aux_out=aux)
            c3 = ops.gemm(a, b, out=res.clone(), accumulate=True)
            gu_i, h = ops.gemm_swiglu(a, b) if swiglu else (c0, c0)
            dgu = ops.gemm_swiglu_bwd(a, wd_t, gu) if swiglu else c0
            outs[tile] = (c0, c1, aux, c2, c3, gu_i, h, dgu)
        assert relerr(outs[5][0], a.float() @ b.float().t()) < 6e-3
        for x, y in zip(outs[5], outs[2]):
            assert torch.equal(x, y)
        ops.set_gemm_tile(5)
        for _ in range(5):
            assert torch.equal(ops.gemm(a, b), outs[5][0])
        ops.set_gemm_tile(5)
        ragged = ops.gemm(a[: M - 8], b)                                  # ragged rows fall out of the DMA descriptors
        assert ops.lib.zk_gemm_last_plan() // 100 == 35 and torch.equal(ragged, outs[2][0][: M - 8])
        odd_k = ops.gemm(a[:, : K - 8], b[:, : K - 8])                    # K not a multiple of 64: 256x256 ring
        assert ops.lib.zk_gemm_last_plan() // 100 == 32
    finally:
        ops.set_gemm_tile(-1)


@pytest.mark.parametrize("tile", [2, 4, 5], ids=["t256", "t384x192", "t192x256"])
def test_gemm_ring_epilogue_stays_inside_its_operands(tile):
    """The ring's epilogue reaches C / residual / the pre-activation copy through buffer descriptors that end at row M,
    with masked columns sent out of range: an output that is a strided window of a larger canary buffer (ragged M and
    N, row stride > N) must leave every element outside the window untouched, and equal the one-tile-per-workgroup
    kernel inside it."""
    M, N, K, LD, PAD = 4600, 4104, 320, 4104 + 72, 16
    a, b = rnd(M, K, seed=151), rnd(N, K, seed=152, scale=0.5)
    bias, res_full = rnd(N, seed=153), rnd(M + 2 * PAD, LD, seed=154)
    res = res_full[PAD:PAD + M, 8:8 + N]

    def window():
        buf = torch.full((M + 2 * PAD, LD), 7.0, dtype=torch.bfloat16, device=DEV)
        return buf, buf[PAD:PAD + M, 8:8 + N]

    def untouched(buf):
        probe = buf.clone()
        probe[PAD:PAD + M, 8:8 + N] = 7.0
        return bool((probe == 7.0).all())

    try:
        ops.set_gemm_tile(tile)
        got = {}
        for pr in (51, 50):
            ops.set_gemm_tile(pr)
            outs = []
            for kw in (dict(), dict(bias=bias, residual=res), dict(bias=bias, epilogue=ops.EPI_QUICK_GELU, aux=True),
                       dict(epilogue=ops.EPI_MUL_DGELU_ERF, aux_in=res, alpha=0.5)):
                kw = dict(kw)
                buf, win = window()
                abuf = None
                if kw.pop("aux", False):
                    abuf, awin = window()
                    kw["aux_out"] = awin
                ops.gemm(a, b, out=win, **kw)
                if pr == 51:
                    assert ops.lib.zk_gemm_last_plan() // 1000 == 3, "persistent ring expected"
                assert untouched(buf) and (abuf is None or untouched(abuf))
                outs.append(win.clone())
                if abuf is not None:
                    outs.append(awin.clone())
            got[pr] = outs
        for x, y in zip(got[51], got[50]):
            assert torch.equal(x, y)
        assert relerr(got[51][0], a.float() @ b.float().t()) < 6e-3
    finally:
        ops.set_gemm_tile(51)
        ops.set_gemm_tile(-1)


def test_gemm_ring_swiglu_epilogues_stay_inside_their_operands():
    """Same for the SwiGLU pair: ragged M, gate|up / h / d(gate|up) as windows of canary buffers."""
    M, F, K, D = 4600, 2048, 320, 512
    x, wi = rnd(M, K, seed=161), rnd(2 * F, K, seed=162, scale=0.2)
    gu, h = ops.gemm_swiglu(x, wi)
    assert ops.lib.zk_gemm_last_plan() // 1000 == 3
    dy, wdt = rnd(M, D, seed=163), rnd(F, D, seed=164, scale=0.2)
    dgu = ops.gemm_swiglu_bwd(dy, wdt, gu)
    assert ops.lib.zk_gemm_last_plan() // 1000 == 3
    PAD = 16
    big_gu = torch.full((M + 2 * PAD, 2 * F + 64), 7.0, dtype=torch.bfloat16, device=DEV)
    big_h = torch.full((M + 2 * PAD, F + 64), 7.0, dtype=torch.bfloat16, device=DEV)
    big_d = torch.full((M + 2 * PAD, 2 * F + 64), 7.0, dtype=torch.bfloat16, device=DEV)
    wgu, wh, wd = big_gu[PAD:PAD + M, :2 * F], big_h[PAD:PAD + M, :F], big_d[PAD:PAD + M, :2 * F]
    ws = ops._workspace(x.device)
    rc = ops._gemm_call(ops._p(x), ops._p(wi), ops._p(wgu), M, 2 * F, K, x.stride(0), wi.stride(0), wgu.stride(0), 0, 0,
                        ops.EPI_SWIGLU, None, None, 0, ops._p(wh), None, wh.stride(0), 1.0, None, 0, ws.data_ptr(),
                        ops.WORKSPACE_BYTES)
    assert rc == 0 and ops.lib.zk_gemm_last_plan() // 1000 == 3
    rc = ops._gemm_call(ops._p(dy), ops._p(wdt), ops._p(wd), M, F, D, dy.stride(0), wdt.stride(0), wd.stride(0), 0, 0,
                        ops.EPI_SWIGLU_BWD, None, None, 0, None, ops._p(wgu), wgu.stride(0), 1.0, None, 0, ws.data_ptr(),
                        ops.WORKSPACE_BYTES)
    assert rc == 0 and ops.lib.zk_gemm_last_plan() // 1000 == 3
    assert torch.equal(wgu, gu) and torch.equal(wh, h) and torch.equal(wd, dgu)
    for big, win_cols in ((big_gu, 2 * F), (big_h, F), (big_d, 2 * F)):
        probe = big.clone()
        probe[PAD:PAD + M, :win_cols] = 7.0
        assert bool((probe == 7.0).all())


def test_gemm_persistent_ring_repeatable():
    """Same launch 20 times: a read that overtakes its DMA (or a slot refilled too early) shows up as
    run-to-run differences; also against the fp32 product at the decoder's gate|up shape."""
    M, N, K = 6144, 16384, 3072
    a, b = rnd(M, K, seed=55), rnd(N, K, seed=56, scale=0.05)
    first = ops.gemm(a, b)
    ref = (a.float() @ b.float().t())
    assert relerr(first, ref) < 6e-3
    del ref
    for _ in range(20):
        assert torch.equal(ops.gemm(a, b), first)


@pytest.mark.parametrize("M,F,K", [(4616, 2048, 320), (6144, 8192, 256)])
def test_gemm_persistent_ring_swiglu(M, F, K):
    """Fused SwiGLU epilogues on the persistent kernel: bit-identical to the separate kernels."""
    x, gw, uw = rnd(M, K, seed=61), rnd(F, K, seed=62, scale=0.2), rnd(F, K, seed=63, scale=0.2)
    wgu = torch.cat((gw, uw), 0)
    gu_ref = ops.gemm(x, wgu)
    h_ref = ops.swiglu_fwd(gu_ref, F)
    wi = ops.interleave_gate_up(gw, uw)
    gu_i, h = ops.gemm_swiglu(x, wi)
    blk = gu_i.view(M, F // 32, 2, 32)
    assert torch.equal(blk[:, :, 0].reshape(M, F), gu_ref[:, :F]) and torch.equal(blk[:, :, 1].reshape(M, F), gu_ref[:, F:])
    assert torch.equal(h, h_ref)
    D = 512
    dy, wd = rnd(M, D, seed=64), rnd(D, F, seed=65, scale=0.2)
    dgu_ref = ops.swiglu_bwd(ops.gemm(dy, wd, trans_b=True), gu_ref, F)
    dgu_i = ops.gemm_swiglu_bwd(dy, wd.t().contiguous(), gu_i)
    blk = dgu_i.view(M, F // 32, 2, 32)
    assert torch.equal(blk[:, :, 0].reshape(M, F), dgu_ref[:, :F]) and torch.equal(blk[:, :, 1].reshape(M, F), dgu_ref[:, F:])


@pytest.mark.parametrize("M", [1, 2, 3, 4])
@pytest.mark.parametrize("N,K", [(1280, 8192), (8192, 1024), (8192, 3584), (4099 * 8, 264), (72, 40)])
def test_gemv_decode_rows(M, N, K):
    """M <= 4 rows take the weight-streaming kernel (ops.gemm dispatch); with and without residual,
    strided x rows, against fp32."""
    buf = rnd(M, K + 64, seed=81)
    x = buf[:, 32:32 + K] if K % 8 == 0 and M > 1 else rnd(M, K, seed=81)
    w, res = rnd(N, K, seed=82, scale=0.3), rnd(M, N, seed=83)
    ref = x.float() @ w.float().t()
    y = ops.gemm(x, w)
    assert relerr(y, ref) < 6e-3
    y2 = ops.gemm(x, w, residual=res)
    assert relerr(y2, ref.to(torch.bfloat16).float() + res.float()) < 6e-3
    old, ops.GEMV_MAX_ROWS = ops.GEMV_MAX_ROWS, 0          # the MFMA path on the same rows agrees
    try:
        assert relerr(ops.gemm(x.contiguous(), w), ref) < 6e-3
    finally:
        ops.GEMV_MAX_ROWS = old


@pytest.mark.parametrize("M", [1, 3, 4])
def test_gemv_fused_operands_bit_identical(M):
    """RMSNorm / SwiGLU made inside the decode product == the separate kernels followed by the plain product."""
    d, N, F = 8192, 1280, 3584
    x, lnw, w = rnd(M, d, seed=91), rnd(d, seed=92), rnd(N, d, seed=93, scale=0.05)
    h, _ = ops.rmsnorm_fwd(x, lnw, 1e-5)
    assert torch.equal(ops.gemv_fused(x, w, xmode=1, ln_w=lnw, eps=1e-5), ops.gemm(h, w))
    gu, wd, res = rnd(M, 2 * F, seed=94), rnd(d, F, seed=95, scale=0.05), rnd(M, d, seed=96)
    hm = ops.swiglu_fwd(gu, F)
    assert torch.equal(ops.gemv_fused(gu, wd, xmode=2, residual=res), ops.gemm(hm, wd, residual=res))
    x2, lnw2, w2 = rnd(M, 264, seed=97), rnd(264, seed=98), rnd(72, 264, seed=99)       # ragged K / N
    h2, _ = ops.rmsnorm_fwd(x2, lnw2, 1e-5)
    assert torch.equal(ops.gemv_fused(x2, w2, xmode=1, ln_w=lnw2, eps=1e-5), ops.gemm(h2, w2))


@pytest.mark.parametrize("splits", [100, 102, 105, 116])
def test_gemm_split_k(splits):
    """Small output + long contraction: fp32 slabs per split + finishing pass (wgrad / lm_head dgrad)."""
    ops.set_gemm_tile(splits)
    try:
        dy, x = rnd(4616, 264, seed=5), rnd(4616, 136, seed=6)            # wgrad, ragged K tail
        ref = dy.float().t() @ x.float()
        c = ops.gemm(dy, x, trans_a=True, trans_b=True)
        assert relerr(c, ref) < 6e-3
        c2 = ops.gemm(dy, x, trans_a=True, trans_b=True, out=c.clone(), accumulate=True)
        assert relerr(c2, 2 * ref) < 1e-2
        a, w = rnd(300, 8200, seed=7), rnd(8200, 520, seed=8, scale=0.05)   # dgrad over a long K
        al = torch.tensor([0.5], device=DEV)
        out = torch.zeros(300, 1024, dtype=torch.bfloat16, device=DEV)
        ops.gemm(a, w, trans_b=True, alpha_dev=al, out=out[:, 8:528])
        assert relerr(out[:, 8:528], 0.5 * (a.float() @ w.float())) < 6e-3
        assert out[:, :8].abs().max() == 0 and out[:, 528:].abs().max() == 0
        b2 = rnd(520, 8200, seed=9, scale=0.05)                              # NT over a long K
        assert relerr(ops.gemm(a, b2), a.float() @ b2.float().t()) < 6e-3
    finally:
        ops.set_gemm_tile(100)


def test_gemm_rejects_bad_shapes():
    a, b = rnd(16, 20, seed=1), rnd(16, 20, seed=2)   # K = 20 is not a multiple of 8
    with pytest.raises(RuntimeError):
        ops.gemm(a, b)
    with pytest.raises(ValueError):
        ops.gemm(a.cpu(), b.cpu())


def test_rmsnorm_dweight():
    """dW of LlamaRMSNorm (tf:llama:62-67: w * bf16(x * rstd)) against autograd: partial sums [nparts, D],
    overwritten (stale contents must not leak), bit-reproducible."""
    for rows, D, nparts in ((700, 256, 44), (1536, 3072, 96), (37, 4096, 1), (37, 4096, 7)):
        x, dy, w = rnd(rows, D, seed=71), rnd(rows, D, seed=72), rnd(D, seed=73)
        _, rstd = ops.rmsnorm_fwd(x, w, 1e-5)
        parts = torch.full((nparts, D), float("nan"), dtype=torch.float32, device=DEV)
        ops.rmsnorm_dweight_(dy, x, rstd, parts)
        xf = x.float()
        xh = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)).to(torch.bfloat16).float()
        ref = (dy.float() * xh).sum(0)
        assert relerr(parts.sum(0), ref) < 2e-3
        again = torch.zeros_like(parts)
        ops.rmsnorm_dweight_(dy, x, rstd, again)
        assert torch.equal(parts, again)


# ------------------------------------------------------------------ attention
def ref_attention(q, k, v, causal, scale):
    # q [B,Hq,S,hd] fp32 (requires_grad), k/v [B,Hkv,S,hd]
    B, Hq, S, hd = q.shape
    rep = Hq // k.shape[1]
    kk = k.repeat_interleave(rep, dim=1)
    vv = v.repeat_interleave(rep, dim=1)
    s = (q @ kk.transpose(-1, -2)) * scale
    if causal:
        mask = torch.ones(S, S, dtype=torch.bool, device=q.device).tril()
        s = s.masked_fill(~mask, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return p @ vv, torch.logsumexp(s, dim=-1)


@pytest.mark.parametrize("B,S,Hq,Hkv,hd,causal", [
    (2, 577, 16, 16, 64, False),      # CLIP-L/14-336 geometry
    (1, 64, 2, 2, 64, False),
    (2, 200, 6, 2, 128, True),        # ragged S, GQA rep 3
    (1, 768, 24, 8, 128, True),       # Llama-3.2-3B geometry at S = 768
    (1, 130, 4, 1, 128, False),
    (1, 100, 2, 2, 64, True),
    (48, 130, 16, 16, 64, False),     # 1536 key blocks: the 4-wave dK/dV block (two query halves per wave)
    (64, 200, 12, 4, 64, True),       # the same, causal, GQA rep 3
])
def test_attention_fwd_bwd(B, S, Hq, Hkv, hd, causal):
    W = (Hq + 2 * Hkv) * hd
    qkv = rnd(B * S, W, seed=21)
    q2, k2, v2 = qkv[:, :Hq * hd], qkv[:, Hq * hd:(Hq + Hkv) * hd], qkv[:, (Hq + Hkv) * hd:]
    scale = hd ** -0.5
    o, lse = ops.attn_fwd(q2, k2, v2, B, S, Hq, Hkv, hd, scale, causal)
    qf = q2.float().reshape(B, S, Hq, hd).transpose(1, 2).detach().requires_grad_(True)
    kf = k2.float().reshape(B, S, Hkv, hd).transpose(1, 2).detach().requires_grad_(True)
    vf = v2.float().reshape(B, S, Hkv, hd).transpose(1, 2).detach().requires_grad_(True)
    ro, rlse = ref_attention(qf, kf, vf, causal, scale)
    ro2 = ro.transpose(1, 2).reshape(B * S, Hq * hd)
    assert relerr(o, ro2) < 8e-3, "o"
    assert (lse - rlse).abs().max().item() < 2e-3, "lse"
    d_o = rnd(B * S, Hq * hd, seed=22)
    ro2.backward(d_o.float())
    dqkv = torch.full((B * S, W), float("nan"), dtype=torch.bfloat16, device=DEV)
    dq, dk, dv = dqkv[:, :Hq * hd], dqkv[:, Hq * hd:(Hq + Hkv) * hd], dqkv[:, (Hq + Hkv) * hd:]
    ops.attn_bwd(q2, k2, v2, o, d_o, lse, dq, dk, dv, B, S, Hq, Hkv, hd, scale, causal)
    assert relerr(dq, qf.grad.transpose(1, 2).reshape(B * S, -1)) < 1.5e-2, "dq"
    assert relerr(dk, kf.grad.transpose(1, 2).reshape(B * S, -1)) < 1.5e-2, "dk"
    assert relerr(dv, vf.grad.transpose(1, 2).reshape(B * S, -1)) < 1.5e-2, "dv"


def test_attention_online_softmax_rescale_branch():
    # a spike late in the key sequence forces the running max to jump at a later tile
    B, S, H, hd = 1, 256, 1, 64
    q = rnd(S, hd, seed=31); k = rnd(S, hd, seed=32); v = rnd(S, hd, seed=33)
    k[200] = (q[5].float() * 4).to(torch.bfloat16)
    o, lse = ops.attn_fwd(q, k, v, B, S, H, H, hd, hd ** -0.5, False)
    ro, rlse = ref_attention(q.float()[None, None], k.float()[None, None], v.float()[None, None], False, hd ** -0.5)
    assert relerr(o, ro[0, 0]) < 8e-3
    assert (lse[0, 0] - rlse[0, 0]).abs().max().item() < 5e-3


@pytest.mark.parametrize("B,S,Hq,Hkv,hd,causal", [
    (2, 577, 16, 16, 64, False),      # CLIP geometry: 2.25 query blocks of 256, last key tile crosses S
    (1, 768, 24, 8, 128, True),       # Llama-3.2-3B geometry: 3 query blocks, diagonal tile = each wave's last
    (2, 200, 6, 2, 128, True),        # ragged S inside one block, GQA rep 3, waves without rows
    (1, 64, 2, 2, 64, False),         # a single key tile: first step is also the last
    (1, 1, 4, 2, 128, False),
    (3, 1300, 4, 2, 128, True),       # 21 key tiles: the steady-state loop
    (2, 1100, 4, 4, 64, False),
])
def test_attention_fwd_w64_variant(B, S, Hq, Hkv, hd, causal):
    """zk_attn_fwd_w64 (one wave per SIMD, 64 rows per wave, sub-tile pipelined softmax, thresholded rescale)
    against the fp32 reference at the tolerance of zk_attn_fwd, and against zk_attn_fwd itself."""
    W = (Hq + 2 * Hkv) * hd
    qkv = rnd(B * S, W, seed=23)
    q2, k2, v2 = qkv[:, :Hq * hd], qkv[:, Hq * hd:(Hq + Hkv) * hd], qkv[:, (Hq + Hkv) * hd:]
    scale = hd ** -0.5
    o, lse = ops.attn_fwd(q2, k2, v2, B, S, Hq, Hkv, hd, scale, causal, variant="w64")
    o0, lse0 = ops.attn_fwd(q2, k2, v2, B, S, Hq, Hkv, hd, scale, causal)
    qf = q2.float().reshape(B, S, Hq, hd).transpose(1, 2)
    kf = k2.float().reshape(B, S, Hkv, hd).transpose(1, 2)
    vf = v2.float().reshape(B, S, Hkv, hd).transpose(1, 2)
    ro, rlse = ref_attention(qf, kf, vf, causal, scale)
    ro2 = ro.transpose(1, 2).reshape(B * S, Hq * hd)
    assert relerr(o, ro2) < 8e-3, "o"
    assert (lse - rlse).abs().max().item() < 2e-3, "lse"
    assert relerr(o, o0.float()) < 1e-2 and (lse - lse0).abs().max().item() < 2e-3     # two bf16 roundings apart


def test_attention_fwd_w64_rescale_threshold_branch():
    """The w64 kernel keeps a row's reference maximum until a tile raises it by more than 2^8; bounded random data
    never takes that branch (nor the 'stale reference' path with p > 1), so both are forced: rows whose late keys
    score 3, 7 and 40 units (log2 domain) above everything before, on the first and on later tiles, causal and not."""
    B, H, hd = 1, 2, 128
    for S, causal in ((512, False), (448, True)):
        q = rnd(S, H * hd, seed=41) * 0.5
        k = rnd(S, H * hd, seed=42) * 0.5
        v = rnd(S, H * hd, seed=43)
        # query rows 300.. get keys aligned with them: growth of the running maximum at key tiles 2, 3 and 4
        for row, key, gain in ((300, 150, 3.0), (301, 210, 7.0), (302, 290, 40.0), (400, 100, 40.0), (401, 399, 12.0)):
            qr = q[row, :hd].float()
            k[key, :hd] = (qr / qr.norm() * gain * 0.6931 * hd ** 0.5 / qr.norm()).to(torch.bfloat16)
        scale = hd ** -0.5
        o, lse = ops.attn_fwd(q, k, v, B, S, H, H, hd, scale, causal, variant="w64")
        qf = q.float().reshape(B, S, H, hd).transpose(1, 2)
        kf = k.float().reshape(B, S, H, hd).transpose(1, 2)
        vf = v.float().reshape(B, S, H, hd).transpose(1, 2)
        ro, rlse = ref_attention(qf, kf, vf, causal, scale)
        assert torch.isfinite(o.float()).all()
        assert relerr(o, ro.transpose(1, 2).reshape(B * S, H * hd)) < 8e-3
        assert (lse - rlse).abs().max().item() < 5e-3


@pytest.mark.parametrize("B,S,Hq,Hkv,hd,causal", [
    (2, 200, 6, 2, 128, True),        # ragged S, GQA rep 3
    (1, 768, 24, 8, 128, True),       # Llama-3.2-3B geometry at S = 768
    (2, 130, 4, 2, 64, False),
    (1, 1, 4, 2, 128, False),
])
def test_attention_bwd_rotary_epilogue_is_rope_inverse(B, S, Hq, Hkv, hd, causal):
    """zk_attn_bwd_rope leaves dq / dk exactly as zk_attn_bwd followed by zk_rope(inverse) does (bit for bit), dv
    as zk_attn_bwd does."""
    W = (Hq + 2 * Hkv) * hd
    qkv = rnd(B * S, W, seed=23)
    q2, k2, v2 = qkv[:, :Hq * hd], qkv[:, Hq * hd:(Hq + Hkv) * hd], qkv[:, (Hq + Hkv) * hd:]
    scale = hd ** -0.5
    o, lse = ops.attn_fwd(q2, k2, v2, B, S, Hq, Hkv, hd, scale, causal)
    d_o = rnd(B * S, Hq * hd, seed=24)
    inv = 1.0 / (500000.0 ** (torch.arange(0, hd, 2).float() / hd))
    fr = torch.outer(torch.arange(S + 3).float(), inv)      # a table longer than S is fine
    cos_t, sin_t = fr.cos().to(DEV).contiguous(), fr.sin().to(DEV).contiguous()

    def run(fused):
        d = torch.full((B * S, W), float("nan"), dtype=torch.bfloat16, device=DEV)
        ops.attn_bwd(q2, k2, v2, o, d_o, lse, d[:, :Hq * hd], d[:, Hq * hd:(Hq + Hkv) * hd], d[:, (Hq + Hkv) * hd:],
                     B, S, Hq, Hkv, hd, scale, causal, rope=(cos_t, sin_t) if fused else None)
        if not fused:
            ops.rope_(d, cos_t, sin_t, S, Hq + Hkv, hd, inverse=True)
        return d

    plain, fused = run(False), run(True)
    assert not torch.isnan(fused.float()).any()
    assert torch.equal(plain, fused)
    # a table shorter than S, or one table without the other, is refused before anything is launched
    d = torch.empty((B * S, W), dtype=torch.bfloat16, device=DEV)
    args = (q2, k2, v2, o, d_o, lse, d[:, :Hq * hd], d[:, Hq * hd:(Hq + Hkv) * hd], d[:, (Hq + Hkv) * hd:], B, S, Hq, Hkv,
            hd, scale, causal)
    if S > 1:
        with pytest.raises(ValueError):
            ops.attn_bwd(*args, rope=(cos_t[: S - 1], sin_t[: S - 1]))
    rc = ops.lib.zk_attn_bwd_rope(q2.data_ptr(), k2.data_ptr(), v2.data_ptr(), o.data_ptr(), d_o.data_ptr(), lse.data_ptr(),
                                  torch.empty_like(lse).data_ptr(), d.data_ptr(), d.data_ptr(), d.data_ptr(), B, S, Hq, Hkv,
                                  hd, W, W, W, Hq * hd, W, W, W, scale, int(causal), cos_t.data_ptr(), None, None)
    assert rc == -22      # ZK_EINVAL (-EINVAL)


# ------------------------------------------------------------------ norms
@pytest.mark.parametrize("rows,D", [(577 * 2, 1024), (37, 256), (100, 3072), (8, 4096), (5, 8192)])
def test_layernorm_fwd(rows, D):
    x, g, b = rnd(rows, D, seed=41, scale=2.0), rnd(D, seed=42), rnd(D, seed=43)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
    ref = F.layer_norm(x.float(), (D,), g.float(), b.float(), 1e-5)
    assert relerr(y, ref) < 6e-3
    assert relerr(mean, x.float().mean(-1)) < 1e-4 or (mean - x.float().mean(-1)).abs().max() < 1e-5


@pytest.mark.parametrize("rows,D,nparts", [(577 * 2, 1024, 73), (577 * 8, 1024, 289), (577 * 64, 1024, 512), (37, 256, 3),
                                           (5, 1024, 1), (50, 3072, 4), (50, 3072, 13)])
def test_layernorm_bwd(rows, D, nparts):
    """dx and the affine gradients as partial sums [nparts, D] (overwritten: stale NaNs must not leak; the parts
    added in order give the gradient; two runs are bit-identical — no atomics)."""
    x, g, b = rnd(rows, D, seed=44, scale=2.0), rnd(D, seed=45), rnd(D, seed=46)
    dy, dres = rnd(rows, D, seed=47), rnd(rows, D, seed=48)
    y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-5)
    dgamma = torch.full((nparts, D), float("nan"), device=DEV); dbeta = torch.full((nparts, D), float("nan"), device=DEV)
    dx = ops.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, dgamma=dgamma, dbeta=dbeta)
    xf = x.float().requires_grad_(True); gf = g.float().requires_grad_(True); bf = b.float().requires_grad_(True)
    F.layer_norm(xf, (D,), gf, bf, 1e-5).backward(dy.float())
    assert relerr(dx, xf.grad + dres.float()) < 8e-3
    assert relerr(dgamma.sum(0), gf.grad) < 1e-3
    assert relerr(dbeta.sum(0), bf.grad) < 1e-3
    dg2, db2 = torch.zeros_like(dgamma), torch.zeros_like(dbeta)
    dx_again = ops.layernorm_bwd(dy, x, g, mean, rstd, dres=dres, dgamma=dg2, dbeta=db2)
    assert torch.equal(dx, dx_again) and torch.equal(dgamma, dg2) and torch.equal(dbeta, db2)
    dx2 = ops.layernorm_bwd(dy, x, g, mean, rstd)
    assert relerr(dx2, xf.grad) < 8e-3
    # the ordered part sum + bf16 conversion the layer nodes use (zk_f32_to_bf16_multi)
    # (a fixed reduction tree over the parts: 16 interleaved groups, then the groups in order)
    slot = torch.zeros(D, dtype=torch.bfloat16, device=DEV)
    ops.f32_to_bf16_multi_([(dgamma, slot, False)])
    tree = torch.stack([dgamma[g_::16].sum(0) if g_ < nparts else torch.zeros(D, device=DEV) for g_ in range(16)]).sum(0)
    assert relerr(slot, tree) < 5e-3
    slot2 = torch.ones(D, dtype=torch.bfloat16, device=DEV)
    ops.f32_to_bf16_multi_([(dgamma, slot2, False)])
    assert torch.equal(slot, slot2)


@pytest.mark.parametrize("rows,D", [(768 * 2, 3072), (37, 256), (9, 4096), (5, 8192)])
def test_rmsnorm(rows, D):
    x, w = rnd(rows, D, seed=51, scale=3.0), rnd(D, seed=52)
    y, rstd = ops.rmsnorm_fwd(x, w, 1e-5)
    xf = x.float().requires_grad_(True)
    nrm = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)
    ref = w.float() * nrm
    assert relerr(y, ref) < 8e-3
    # exact cast-point check (tf:llama:67): w * bf16(x * rstd)
    exact = (w.float() * nrm.detach().to(torch.bfloat16).float()).to(torch.bfloat16)
    assert (y.float() - exact.float()).abs().max().item() <= 2 * 2 ** -8 * exact.float().abs().max().item()
    dy, dres = rnd(rows, D, seed=53), rnd(rows, D, seed=54)
    ref.backward(dy.float())
    dx = ops.rmsnorm_bwd(dy, x, w, rstd, dres=dres)
    assert relerr(dx, xf.grad + dres.float()) < 8e-3


# ------------------------------------------------------------------ rope / swiglu / act
def test_rope_matches_rotate_half_and_inverts():
    S, H, hd, Hk = 50, 3, 128, 1
    rows = 2 * S
    W = (H + 2 * Hk) * hd
    qkv = rnd(rows, W, seed=61)
    inv = 1.0 / (500000.0 ** (torch.arange(0, hd, 2).float() / hd))
    fr = torch.outer(torch.arange(S).float(), inv)
    cos_t = fr.cos().to(torch.bfloat16).float().to(DEV).contiguous()
    sin_t = fr.sin().to(torch.bfloat16).float().to(DEV).contiguous()
    orig = qkv.clone()
    ops.rope_(qkv, cos_t, sin_t, S, H + Hk, hd)
    x = orig[:, :(H + Hk) * hd].float().reshape(rows, H + Hk, hd)
    c = torch.cat([cos_t, cos_t], -1).repeat(2, 1)[:, None, :]
    s = torch.cat([sin_t, sin_t], -1).repeat(2, 1)[:, None, :]
    rot = torch.cat([-x[..., hd // 2:], x[..., :hd // 2]], -1)
    ref = (x * c + rot * s).reshape(rows, -1)
    assert relerr(qkv[:, :(H + Hk) * hd], ref) < 6e-3
    assert torch.equal(qkv[:, (H + Hk) * hd:], orig[:, (H + Hk) * hd:])      # v untouched
    ops.rope_(qkv, cos_t, sin_t, S, H + Hk, hd, inverse=True)
    # the bf16 table is not exactly orthonormal: cos^2+sin^2 = 1 +- 2^-8
    assert relerr(qkv, orig) < 2e-2


def test_swiglu_fwd_bwd():
    rows, Fd = 300, 512
    gu, dh = rnd(rows, 2 * Fd, seed=62, scale=2.0), rnd(rows, Fd, seed=63)
    h = ops.swiglu_fwd(gu, Fd)
    gf = gu.float().requires_grad_(True)
    ref = F.silu(gf[:, :Fd]) * gf[:, Fd:]
    assert relerr(h, ref) < 8e-3
    ref.backward(dh.float())
    dgu = ops.swiglu_bwd(dh, gu, Fd)
    assert relerr(dgu, gf.grad) < 8e-3


@pytest.mark.parametrize("kind", ["quick", "erf"])
def test_act_bwd(kind):
    pre, dy = rnd(100, 256, seed=64, scale=2.0), rnd(100, 256, seed=65)
    pf = pre.float().requires_grad_(True)
    (pf * torch.sigmoid(1.702 * pf) if kind == "quick" else F.gelu(pf)).backward(dy.float())
    got = ops.act_bwd(dy, pre, ops.EPI_QUICK_GELU if kind == "quick" else ops.EPI_GELU_ERF)
    assert relerr(got, pf.grad) < 8e-3


def test_colsum():
    dy = rnd(1154, 1032, seed=66)
    out = torch.full((18, 1032), float("nan"), device=DEV)          # partial sums, overwritten
    ops.colsum_(dy, out)
    assert relerr(out.sum(0), dy.float().sum(0)) < 1e-4
    again = torch.zeros_like(out)
    ops.colsum_(dy, again)
    assert torch.equal(out, again)                                  # no atomics: bit-reproducible
    view = dy[:, 8:520]
    out2 = torch.full((1, 512), float("nan"), device=DEV)
    ops.colsum_(view, out2)
    assert relerr(out2[0], view.float().sum(0)) < 1e-4
    # column slices of one partial-sum slab go through the converter with their part stride (packed q|k|v biases)
    slot = torch.ones(512, dtype=torch.bfloat16, device=DEV)
    ops.f32_to_bf16_multi_([(out[:, 8:520], slot, True)])
    assert relerr(slot, out[:, 8:520].sum(0) + 1) < 5e-3
    slot2 = torch.ones(512, dtype=torch.bfloat16, device=DEV)
    ops.f32_to_bf16_multi_([(out[:, 8:520], slot2, True)])
    assert torch.equal(slot, slot2)


# ------------------------------------------------------------------ patch embedding / gather / CE / optimizer
def test_patch_embed_front_end():
    B, H, P, D = 3, 42, 14, 1024
    px = rnd(B, 3, H, H, seed=71)
    w = rnd(D, 3, P, P, seed=72, scale=0.05)
    cls, pos = rnd(D, seed=73), rnd(10, D, seed=74)
    Kpad = 640
    cols = ops.patch_im2col(px, P, Kpad)
    wp = torch.zeros(D, Kpad, dtype=torch.bfloat16, device=DEV)
    wp[:, :588] = w.reshape(D, -1)
    patch = ops.gemm(cols, wp)
    out = ops.embed_finish(patch, cls, pos, B, 10, D)
    conv = F.conv2d(px.float(), w.float(), stride=P).flatten(2).transpose(1, 2)
    ref = torch.cat([cls.float().expand(B, 1, D), conv], 1) + pos.float()[None]
    assert relerr(out, ref) < 8e-3
    unf = F.unfold(px.float(), kernel_size=P, stride=P).transpose(1, 2).reshape(-1, 588)
    assert torch.equal(cols[:, :588].float(), unf) and cols[:, 588:].abs().max() == 0


def test_gather_rows_splice_and_accumulate():
    D = 256
    wte, img = rnd(50, D, seed=75), rnd(12, D, seed=76)
    codes = torch.tensor([3, 49, -1, -2, -13, 0, -1], dtype=torch.int32, device=DEV)
    out = ops.gather_rows(wte, img, codes, D)
    ref = torch.stack([wte[3], wte[49], torch.zeros(D, device=DEV, dtype=torch.bfloat16), img[0], img[11], wte[0],
                       torch.zeros(D, device=DEV, dtype=torch.bfloat16)])
    assert torch.equal(out, ref)
    out2 = ops.gather_rows(wte, img, codes, D, out=out.clone(), accumulate=True)
    assert relerr(out2, 2 * ref.float()) < 6e-3


@pytest.mark.parametrize("rows,V", [(37, 512), (16, 128256)])
def test_cross_entropy(rows, V):
    logits = rnd(rows, V, seed=77, scale=3.0)
    g = torch.Generator().manual_seed(78)
    tgt = torch.randint(0, V, (rows,), generator=g)
    tgt[::5] = -100
    tgt_d = tgt.to(torch.int32).to(DEV)
    acc = torch.zeros(2, device=DEV)
    dl = torch.empty_like(logits)
    ops.cross_entropy(logits, tgt_d, acc, dl)
    lf = logits.float().requires_grad_(True)
    ref = F.cross_entropy(lf, tgt.to(DEV), ignore_index=-100, reduction="sum")
    ref.backward()
    assert abs(acc[0].item() - ref.item()) < 1e-4 * abs(ref.item())
    assert acc[1].item() == (tgt != -100).sum().item()
    assert relerr(dl, lf.grad) < 8e-3
    # in place
    # in place (LMHeadLossFn's mode): same gradient AND the same loss — the target logit must be read
    # before any wave overwrites it with softmax - onehot; repeated so a timing-dependent race shows
    for _ in range(4):
        l2 = logits.clone()
        acc2 = torch.zeros(2, device=DEV)
        ops.cross_entropy(l2, tgt_d, acc2, l2)
        assert torch.equal(l2, dl)
        assert abs(acc2[0].item() - acc[0].item()) <= 1e-5 * abs(acc[0].item())
        assert acc2[1].item() == acc[1].item()


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_cross_entropy_nan_row_reaches_the_loss(bad):
    """engine:315 raises on a NaN loss: a row with a NaN / inf logit must make loss_sum NaN (the ordered row sum
    once clamped it away through fmaxf and counted the row as a zero-loss row)."""
    rows, V = 24, 1024
    logits = rnd(rows, V, seed=79, scale=2.0)
    logits[7, 100] = bad
    tgt = torch.arange(rows, dtype=torch.int32, device=DEV)
    tgt[3] = -100
    acc = torch.zeros(2, device=DEV)
    ops.cross_entropy(logits, tgt, acc, None)
    assert torch.isnan(acc[0]).item()
    assert acc[1].item() == rows - 1                           # the NaN row is still a counted row
    logits[7, 100] = 0.0                                       # and an ignored row with a NaN stays ignored
    logits[3, 5] = bad
    acc = torch.zeros(2, device=DEV)
    ops.cross_entropy(logits, tgt, acc, None)
    assert torch.isfinite(acc[0]).item() and acc[1].item() == rows - 1


def test_sumsq_adamw_f32tobf16():
    n = 100003
    p, g = rnd(n, seed=81), rnd(n, seed=82, scale=0.1)
    m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    ss = torch.zeros(1, device=DEV)
    ops.sumsq_(g, ss)
    assert abs(ss.item() - (g.float() ** 2).sum().item()) < 1e-4 * ss.item()
    pref = p.float().clone().requires_grad_(True)
    opt = torch.optim.AdamW([pref], lr=1e-2, betas=(0.9, 0.999), weight_decay=0.01)
    p_cur = p.clone()
    for step in (1, 2):
        pref.grad = g.float().clone()
        torch.nn.utils.clip_grad_norm_([pref], 1.0)
        opt.step()
        ops.adamw_step_(p_cur, g, m, v, lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=step,
                        max_norm=1.0, sumsq=ss)
        if step == 1:
            # bf16 parameter storage: one rounding of the updated value
            assert (p_cur.float() - pref.detach()).abs().max().item() <= 2 ** -8 * pref.abs().max().item()
    src = torch.randn(1000, device=DEV)
    dst = rnd(1000, seed=83)
    d0 = dst.clone()
    ops.f32_to_bf16_(src, dst, scale=0.5, accumulate=True)
    assert relerr(dst, d0.float() + 0.5 * src) < 6e-3


@pytest.mark.parametrize("R,C", [(1024, 4096), (3072, 1024), (136, 72), (8, 8), (4096, 1000)])
def test_transpose(R, C):
    """zk_transpose_bf16: bit-exact, contiguous and row-strided sources (a packed q|k|v view)."""
    w = rnd(R, C, seed=71)
    assert torch.equal(ops.transpose(w), w.t().contiguous())
    big = rnd(R, C + 24, seed=72)
    view = big[:, 8:8 + C]
    assert torch.equal(ops.transpose(view), view.t().contiguous())

