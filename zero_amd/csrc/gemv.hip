// Skinny products for decoding: y[m][n] = sum_k x[m][k] W[n][k] (+ residual[m][n]), M <= 4 rows.
//
// Replaces F.linear (tf:llama:174-176,230-241,280,480) on the single-token steps of greedy decoding
// (ref:inference/vlm_infer.py:259-270): every weight byte is read once per token, so this is HBM-bound
// byte work, not MFMA work — the 128/256-row GEMM tiles would stream the weights through a handful of
// CUs.  Four output rows per 256-thread block; the four waves split K in 512-element chunks (a lane
// owns 8 consecutive k: one 16-byte load per weight row and chunk, 16 independent loads in flight per
// lane at K = 8192), the lane's x values stay in registers, partial sums are reduced across the wave
// with DPP adds and across the waves through LDS.  fp32 accumulate, bf16 in/out.  The x operand can
// be produced on the fly (RMSNorm of the hidden row, SwiGLU of the gate|up row): two fewer launches
// per decoder-layer half.
#include "common.h"
#include "../../include/zero_hip.h"

namespace {

constexpr int R = 4;      // output rows (weight rows) per block
constexpr int NC = 4;     // k-chunks (8 elements) per thread and group: one group covers K = 8192

// XMODE: what the x operand is.  0 = x as stored; 1 = RMSNorm of x (tf:llama:62-67: w * bf16(x * rstd),
// rounded to bf16 like the separate kernel's output; the block's 256 threads partition K exactly like
// rmsnorm_fwd_block_kernel, so rstd is bit-identical); 2 = SwiGLU of the packed gate|up row
// (x[k] = bf16(bf16(silu(g[k])) * u[k]), g = xin[k], u = xin[K + k]; tf:llama:175).
// The weight loads of the first group are issued BEFORE the row statistics, so the two HBM/L2 round
// trips overlap (with them after the reduction's barrier the fused kernel was slower than two launches).
template <int M, int XMODE>
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* x, const bf16_t* W, bf16_t* y, const bf16_t* res,
                                                   const bf16_t* ln_w, float eps, int N, int K, int ldx, int ldw,
                                                   int ldy, int ldr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * R;
  const int k0 = wave * 512 + lane * 8;
  __shared__ float red[4][M * R];
  const u32x4 zero = {0u, 0u, 0u, 0u};
  float acc[M][R];
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < R; ++r) acc[m][r] = 0.f;
  u32x4 wv[NC][R];
  auto load_w = [&](int kb) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int k = kb + k0 + 2048 * c;
#pragma unroll
      for (int r = 0; r < R; ++r)
        wv[c][r] = (k < K && n0 + r < N) ? *reinterpret_cast<const u32x4*>(W + (size_t)(n0 + r) * ldw + k) : zero;
    }
  };
  load_w(0);
  float rstd[M];
  if (XMODE == 1) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      float ss = 0.f;
      for (int k = k0; k < K; k += 2048) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + (size_t)m * ldx + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = lo_bf(v[e]), b = hi_bf(v[e]);
          ss += a * a;
          ss += b * b;
        }
      }
      ss = wave_sum(ss);
      if (lane == 0) red[wave][m] = ss;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < M; ++m) rstd[m] = rsqrtf((red[0][m] + red[1][m] + red[2][m] + red[3][m]) / (float)K + eps);
    __syncthreads();     // red is reused below
  }
  for (int kb = 0;;) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int k = kb + k0 + 2048 * c;
      if (k < K) {
        u32x4 lw = zero;
        if (XMODE == 1) lw = *reinterpret_cast<const u32x4*>(ln_w + k);
#pragma unroll
        for (int m = 0; m < M; ++m) {
          u32x4 xv = *reinterpret_cast<const u32x4*>(x + (size_t)m * ldx + k);
          if (XMODE == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              xv[e] = pack2bf(lo_bf(lw[e]) * bf2f(f2bf(lo_bf(xv[e]) * rstd[m])),
                              hi_bf(lw[e]) * bf2f(f2bf(hi_bf(xv[e]) * rstd[m])));
          } else if (XMODE == 2) {
            const u32x4 uv = *reinterpret_cast<const u32x4*>(x + (size_t)m * ldx + K + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float g0 = lo_bf(xv[e]), g1 = hi_bf(xv[e]);
              const float s0 = bf2f(f2bf(silu_fast(g0))), s1 = bf2f(f2bf(silu_fast(g1)));
              xv[e] = pack2bf(s0 * lo_bf(uv[e]), s1 * hi_bf(uv[e]));
            }
          }
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[m][r] += lo_bf(xv[e]) * lo_bf(wv[c][r][e]) + hi_bf(xv[e]) * hi_bf(wv[c][r][e]);
        }
      }
    }
    kb += 2048 * NC;
    if (kb >= K) break;
    load_w(kb);
  }
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float s = wave_sum(acc[m][r]);
      if (lane == 0) red[wave][m * R + r] = s;
    }
  __syncthreads();
  if (tid < M * R) {
    const int m = tid / R, n = n0 + tid % R;
    if (n < N) {
      float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
      // like the GEMM epilogue: the branch output is rounded to bf16 before the residual add
      if (res) v = bf2f(f2bf(v)) + bf2f(res[(size_t)m * ldr + n]);
      y[(size_t)m * ldy + n] = f2bf(v);
    }
  }
}

// Short contractions (K <= 2048: the o-projection of a tensor-parallel shard, K = 1024): with K split over
// the four waves half of them would idle and a thread would have 4 loads in flight.  Here a WAVE owns
// RW output rows and its lanes split K, two 512-element chunks per step: 8 weight loads in flight per
// lane, no cross-wave reduction, 16 rows per block.
template <int M>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const bf16_t* x, const bf16_t* W, bf16_t* y, const bf16_t* res,
                                                        int N, int K, int ldx, int ldw, int ldy, int ldr) {
  constexpr int RW = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 4 + wave) * RW;
  if (n0 >= N) return;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  float acc[M][RW];
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[m][r] = 0.f;
  for (int kb = lane * 8; kb < K; kb += 1024) {
    u32x4 wv[2][RW], xv[2][M];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int k = kb + 512 * c;
#pragma unroll
      for (int r = 0; r < RW; ++r)
        wv[c][r] = (k < K && n0 + r < N) ? *reinterpret_cast<const u32x4*>(W + (size_t)(n0 + r) * ldw + k) : zero;
#pragma unroll
      for (int m = 0; m < M; ++m) xv[c][m] = k < K ? *reinterpret_cast<const u32x4*>(x + (size_t)m * ldx + k) : zero;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int m = 0; m < M; ++m)
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[m][r] += lo_bf(xv[c][m][e]) * lo_bf(wv[c][r][e]) + hi_bf(xv[c][m][e]) * hi_bf(wv[c][r][e]);
  }
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      float v = wave_sum(acc[m][r]);
      if (lane == 0 && n0 + r < N) {
        if (res) v = bf2f(f2bf(v)) + bf2f(res[(size_t)m * ldr + n0 + r]);
        y[(size_t)m * ldy + n0 + r] = f2bf(v);
      }
    }
}

template <int XMODE>
int launch_gemv(const void* x, const void* W, void* y, int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw,
                int64_t ldy, const void* residual, int64_t ldr, const void* ln_w, float eps, hipStream_t stream) {
  dim3 grid((unsigned)((N + R - 1) / R)), block(256);
  if (XMODE == 0 && K <= 2048) {
    dim3 g2((unsigned)((N + 15) / 16));
#define CALLR(M_) hipLaunchKernelGGL((gemv_rows_kernel<M_>), g2, block, 0, stream, (const bf16_t*)x, (const bf16_t*)W, \
    (bf16_t*)y, (const bf16_t*)residual, (int)N, (int)K, (int)ldx, (int)ldw, (int)ldy, (int)ldr)
    switch (M) {
      case 1: CALLR(1); break;
      case 2: CALLR(2); break;
      case 3: CALLR(3); break;
      default: CALLR(4); break;
    }
#undef CALLR
    ZK_CHECK_LAUNCH();
    return ZK_OK;
  }
#define CALL(M_) hipLaunchKernelGGL((gemv_kernel<M_, XMODE>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)W, \
    (bf16_t*)y, (const bf16_t*)residual, (const bf16_t*)ln_w, eps, (int)N, (int)K, (int)ldx, (int)ldw, (int)ldy, (int)ldr)
  switch (M) {
    case 1: CALL(1); break;
    case 2: CALL(2); break;
    case 3: CALL(3); break;
    default: CALL(4); break;
  }
#undef CALL
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

}  // namespace

extern "C" int zk_gemv_fused_bf16(const void* x, const void* W, void* y, int64_t M, int64_t N, int64_t K, int64_t ldx,
                                  int64_t ldw, int64_t ldy, const void* residual, int64_t ldr, int xmode,
                                  const void* ln_w, float eps, hipStream_t stream) {
  if (!x || !W || !y || M <= 0 || M > 4 || N <= 0 || K <= 0 || (K & 7) || (ldx & 7) || (ldw & 7))
    return ZK_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W)) & 15) return ZK_EINVAL;
  if (N > 0x7FFFFFF0LL || xmode < 0 || xmode > 2 || (xmode == 1 && !ln_w) || (xmode == 2 && ldx < 2 * K)) return ZK_EINVAL;
  if (xmode == 1) return launch_gemv<1>(x, W, y, M, N, K, ldx, ldw, ldy, residual, ldr, ln_w, eps, stream);
  if (xmode == 2) return launch_gemv<2>(x, W, y, M, N, K, ldx, ldw, ldy, residual, ldr, nullptr, 0.f, stream);
  return launch_gemv<0>(x, W, y, M, N, K, ldx, ldw, ldy, residual, ldr, nullptr, 0.f, stream);
}

extern "C" int zk_gemv_bf16(const void* x, const void* W, void* y, int64_t M, int64_t N, int64_t K, int64_t ldx,
                            int64_t ldw, int64_t ldy, const void* residual, int64_t ldr, hipStream_t stream) {
  return zk_gemv_fused_bf16(x, W, y, M, N, K, ldx, ldw, ldy, residual, ldr, 0, nullptr, 0.f, stream);
}
