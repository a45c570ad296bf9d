// Skinny products for decoding: y[m][n] = sum_k x[m][k] W[n][k] (+ residual[m][n]), M <= 4 rows.
//
// Replaces F.linear (tf:llama:174-176,230-241,280,480) on the single-token steps of greedy decoding
// (ref:inference/vlm_infer.py:259-270): every weight byte is read once per token, so this is HBM-bound
// byte work, not MFMA work — the 128/256-row GEMM tiles would stream the weights through a handful of
// CUs.  Four output rows per 256-thread block; the four waves split K in 512-element chunks (a lane
// owns 8 consecutive k: one 16-byte load per weight row and chunk, 16 independent loads in flight per
// lane at K = 8192), the lane's x values stay in registers, partial sums are reduced across the wave
// with DPP adds and across the waves through LDS.  fp32 accumulate, bf16 in/out.
#include "common.h"
#include "../../include/zero_hip.h"

namespace {

constexpr int R = 4;   // output rows (weight rows) per block

template <int M>
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* x, const bf16_t* W, bf16_t* y, const bf16_t* res,
                                                   int N, int K, int ldx, int ldw, int ldy, int ldr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * R;
  float acc[M][R];
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int r = 0; r < R; ++r) acc[m][r] = 0.f;
  const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll 4
  for (int k = wave * 512 + lane * 8; k < K; k += 2048) {
    u32x4 wv[R], xv[M];
#pragma unroll
    for (int r = 0; r < R; ++r)
      wv[r] = n0 + r < N ? *reinterpret_cast<const u32x4*>(W + (size_t)(n0 + r) * ldw + k) : zero;
#pragma unroll
    for (int m = 0; m < M; ++m) xv[m] = *reinterpret_cast<const u32x4*>(x + (size_t)m * ldx + k);
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[m][r] += lo_bf(xv[m][e]) * lo_bf(wv[r][e]) + hi_bf(xv[m][e]) * hi_bf(wv[r][e]);
  }
  __shared__ float red[4][M * R];
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

}  // namespace

extern "C" int zk_gemv_bf16(const void* x, const void* W, void* y, int64_t M, int64_t N, int64_t K, int64_t ldx,
                            int64_t ldw, int64_t ldy, const void* residual, int64_t ldr, hipStream_t stream) {
  if (!x || !W || !y || M <= 0 || M > 4 || N <= 0 || K <= 0 || (K & 7) || (ldx & 7) || (ldw & 7)) return ZK_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W)) & 15) return ZK_EINVAL;
  if (N > 0x7FFFFFF0LL || K > 0x7FFFFFF0LL) return ZK_EINVAL;
  dim3 grid((unsigned)((N + R - 1) / R)), block(256);
#define CALL(M_) hipLaunchKernelGGL((gemv_kernel<M_>), grid, block, 0, stream, (const bf16_t*)x, (const bf16_t*)W, \
    (bf16_t*)y, (const bf16_t*)residual, (int)N, (int)K, (int)ldx, (int)ldw, (int)ldy, (int)ldr)
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
