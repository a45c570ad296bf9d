"""Host logic of zk_gemm_bf16 without a GPU: zk_gemm_plan_query runs the call's whole decision path (argument
validation, tile quantisation model, split-K model, row split, ring operand limits) and launches nothing.  The codes
are zk_gemm_last_plan's: kind * 1000 + tile * 100 + lean * 10 + epilogue class."""
import ctypes

import pytest

from zero_amd import _lib
from zero_amd.ops import (EPI_NONE, EPI_QUICK_GELU, EPI_MUL_DQUICK_GELU, EPI_SWIGLU, EPI_SWIGLU_BWD, EPI_GELU_ERF,
                          WORKSPACE_BYTES)

lib = _lib.lib


def query(M, N, K, *, lda=None, ldb=None, ldc=None, ta=0, tb=0, epi=EPI_NONE, bias=0, residual=0, ldr=None, aux_out=0,
          aux_in=0, ldaux=0, accumulate=0, ws=WORKSPACE_BYTES, plan=None):
    lda = lda if lda is not None else (M if ta else K)
    ldb = ldb if ldb is not None else (N if tb else K)
    ldc = ldc if ldc is not None else N
    ldr = ldr if ldr is not None else (N if residual else 0)
    return lib.zk_gemm_plan_query(M, N, K, lda, ldb, ldc, ta, tb, epi, bias, residual, ldr, aux_out, aux_in, ldaux,
                                  accumulate, ws, ctypes.byref(plan) if plan is not None else None)


def test_c3_decoder_products_run_on_the_persistent_ring_classes():
    M, d, f, qkv = 6144, 3072, 8192, 5120
    assert query(M, d, 2 * f) == 3410                                   # d_gu: 384x192 (256 tiles = one round), plain
    assert query(M, d, f, residual=1) == 3416                           # down + residual: its own class
    assert query(M, d, d, residual=1) == 3416 and query(M, d, d) == 3410
    assert query(M, qkv, d) == 3210                                     # qkv: 256x256
    assert query(M, 2 * f, d, epi=EPI_SWIGLU, aux_out=1, ldaux=f) == 3212             # SwiGLU forward needs 64 columns per wave
    assert query(M, f, d, epi=EPI_SWIGLU_BWD, aux_in=1, ldaux=2 * f, ldc=2 * f) == 3215
    assert query(1024, 128256, d) == 3210                               # lm_head on the labelled rows


def test_encoder_products_and_split_k():
    M = 8 * 577
    assert query(M, 1024, 4096, bias=1, residual=1) == 300              # too few 256x256 tiles: two-stage 128x96
    assert query(M, 4096, 1024, bias=1, epi=EPI_QUICK_GELU, aux_out=1) == 3511        # fc1: 192x256 ring, GELU pair class
    assert query(M, 4096, 1024, epi=EPI_MUL_DQUICK_GELU, aux_in=1, ldaux=4096) == 3511
    assert query(M, 3072, 1024, bias=1) == 3210
    assert query(M, 3072, 3072, bias=1, epi=EPI_GELU_ERF, aux_out=1) // 10 == 321 and query(M, 3072, 3072, bias=1, epi=EPI_GELU_ERF, aux_out=1) % 10 == 4
    wg = query(1024, 4096, M, ta=1, tb=1)                               # wgrad: split K over the 256 CUs
    assert wg // 1000 == 4 and 2 <= wg - 4000 <= 4
    assert query(1024, 4096, M, ta=1, tb=1, ws=0) // 1000 == 0          # no workspace: never split
    lm = query(1024, 3072, 128256)                                      # lm_head dgrad: long K, small output
    assert lm // 1000 == 4 and lm - 4000 >= 5
    assert query(1024, 4096, M, ta=1, tb=1, accumulate=1) // 1000 == 4  # (the finishing kernel applies C += ...)


def test_c2_row_split_reports_the_ring_part():
    # 36928 x 1024 x 4096: 580 tiles of 256x256 = 2.27 rounds; the built-in plan runs two whole rounds on the ring and
    # the remaining rows as a product of their own; the code is the ring part's
    assert query(36928, 1024, 4096, bias=1, residual=1) == 3216


def test_ring_falls_back_when_an_operand_exceeds_its_32_bit_descriptor():
    M, N, K = 6144, 3072, 8192
    assert query(M, N, K, residual=1) == 3416
    big = (1 << 31) // (2 * M) + 64                                     # M * ld * 2 bytes beyond 2 GiB
    assert query(M, N, K, residual=1, ldr=big) // 1000 != 3             # residual stride too large: not the persistent ring
    assert query(M, N, K, ldc=big) // 1000 != 3
    assert query(M, N, K, accumulate=1) // 1000 != 3                    # C += ... is not a ring class
    p = _lib.GemmPlan.default(); p.tile = 5
    assert query(M, 4096, 512, plan=p) == 3510
    assert query(M, 4096, 512, ldc=big, plan=p) // 100 != 35            # forced 192x256 must not fail: another tile runs
    assert query(M, 4096, 520, plan=p) // 100 != 35                     # K not a multiple of 64


def test_invalid_arguments_are_reported_not_launched():
    assert query(6144, 3076, 512) == -22                                # N not a multiple of 8
    assert query(6144, 3072, 516) == -22                                # k-contiguous operands: K multiple of 8
    assert query(6144, 3072, 512, lda=515) == -22
    assert query(6144, 8192, 512, epi=EPI_SWIGLU, aux_out=0) == -22     # SwiGLU needs its h output
    assert query(6144, 8200, 512, epi=EPI_SWIGLU, aux_out=1, ldaux=4100) == -22       # N multiple of 64
    assert query(6144, 8192, 512, epi=EPI_SWIGLU_BWD, aux_in=1, ldaux=8192, ldc=16384) == -22   # ldaux >= 2N
    p = _lib.GemmPlan.default(); p.tile = 9
    assert query(6144, 3072, 512, plan=p) == -22
    p = _lib.GemmPlan.default(); p.ablation = 1
    assert query(6144, 3072, 512, plan=p) == -22                        # wrong-result timing modes: ZK_ABLATION builds only
