// LayerNorm / RMSNorm forward + backward for gfx950.  HBM-bound: one wave64 per row, the
// row is read ONCE with 16-byte loads and kept in registers for the two-pass statistics.
// Algorithmic bytes: fwd 4 B/elem (read x, write y), bwd 6-8 B/elem (read dy, x[, dres], write dx).
#include "common.h"
#include "../../include/zero_hip.h"

namespace {

template <int MAXV>
__device__ __forceinline__ void load_row(u32x4 (&v)[MAXV], const bf16_t* p, int nv, int lane) {
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int idx = lane + 64 * i;
    v[i] = (idx < nv) ? *reinterpret_cast<const u32x4*>(p + (size_t)idx * 8) : u32x4{0u, 0u, 0u, 0u};
  }
}
__device__ __forceinline__ void unpack8(const u32x4& w, float (&f)[8]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[2 * j] = lo_bf(w[j]); f[2 * j + 1] = hi_bf(w[j]); }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  u32x4 w;
#pragma unroll
  for (int j = 0; j < 4; ++j) w[j] = pack2bf(f[2 * j], f[2 * j + 1]);
  return w;
}

template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* x, const bf16_t* gamma, const bf16_t* beta,
                                                            bf16_t* y, float* mean_o, float* rstd_o, int rows, int D,
                                                            float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = D >> 3;
  u32x4 v[MAXV];
  load_row<MAXV>(v, x + (size_t)row * D, nv, lane);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) { float f[8]; unpack8(v[i], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += f[j]; }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (lane + 64 * i < nv) { float f[8]; unpack8(v[i], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) { float d = f[j] - mean; q += d * d; } }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) { if (mean_o) mean_o[row] = mean; if (rstd_o) rstd_o[row] = rstd; }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int idx = lane + 64 * i;
    if (idx < nv) {
      float f[8], gm[8], bt[8];
      unpack8(v[i], f);
      unpack8(*reinterpret_cast<const u32x4*>(gamma + (size_t)idx * 8), gm);
      unpack8(*reinterpret_cast<const u32x4*>(beta + (size_t)idx * 8), bt);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = (f[j] - mean) * rstd * gm[j] + bt[j];
      *reinterpret_cast<u32x4*>(y + (size_t)row * D + (size_t)idx * 8) = pack8(f);
    }
  }
}

// dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
// dgamma_parts[block] = sum over this block's rows of dy * xhat ; dbeta_parts[block] = sum dy.
// One workgroup of NW waves per part; a wave owns rows part*NW + wave (+ nparts*NW ...) and, when PF, requests
// the NEXT row's x / dy / dres before it reduces the current one (a wave walks only a few rows: without the
// prefetch every row pays a full dependent HBM round trip, 0.21 of peak in round 1).  The per-block sums are
// combined through LDS in wave order and STORED (no atomics): the caller adds the parts in index order, so the
// affine gradients are bit-reproducible.
template <int MAXV, int NW, bool PF>
__global__ __launch_bounds__(64 * NW) void layernorm_bwd_kernel(const bf16_t* dy, const bf16_t* x, const bf16_t* gamma,
                                                                const float* mean_i, const float* rstd_i,
                                                                const bf16_t* dres, bf16_t* dx, float* dgamma,
                                                                float* dbeta, int rows, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int stride = gridDim.x * NW;
  const int nv = D >> 3;
  float ag[MAXV][8], ab[MAXV][8];
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; }
  float gm[MAXV][8];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int idx = lane + 64 * i;
    if (idx < nv) unpack8(*reinterpret_cast<const u32x4*>(gamma + (size_t)idx * 8), gm[i]);
    else
#pragma unroll
      for (int j = 0; j < 8; ++j) gm[i][j] = 0.f;
  }
  int row = blockIdx.x * NW + wave;
  u32x4 vx[MAXV], vd[MAXV], vr[MAXV];
  float mean = 0.f, rstd = 0.f;
  if (row < rows) {
    load_row<MAXV>(vx, x + (size_t)row * D, nv, lane);
    load_row<MAXV>(vd, dy + (size_t)row * D, nv, lane);
    if (dres && dx) load_row<MAXV>(vr, dres + (size_t)row * D, nv, lane);
    mean = mean_i[row]; rstd = rstd_i[row];
  }
  for (; row < rows; row += stride) {
    u32x4 nx[MAXV], nd[MAXV], nr[MAXV];
    float nmean = 0.f, nrstd = 0.f;
    const int nrow = row + stride;
    if (PF && nrow < rows) {
      load_row<MAXV>(nx, x + (size_t)nrow * D, nv, lane);
      load_row<MAXV>(nd, dy + (size_t)nrow * D, nv, lane);
      if (dres && dx) load_row<MAXV>(nr, dres + (size_t)nrow * D, nv, lane);
      nmean = mean_i[nrow]; nrstd = rstd_i[nrow];
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (lane + 64 * i < nv) {
        float fx[8], fd[8];
        unpack8(vx[i], fx); unpack8(vd[i], fd);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float xh = (fx[j] - mean) * rstd;
          float g = fd[j] * gm[i][j];
          s1 += g; s2 += g * xh;
          ag[i][j] += fd[j] * xh; ab[i][j] += fd[j];
        }
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
    if (dx) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        int idx = lane + 64 * i;
        if (idx < nv) {
          float fx[8], fd[8], o[8], r[8];
          unpack8(vx[i], fx); unpack8(vd[i], fd);
          if (dres) unpack8(vr[i], r);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float xh = (fx[j] - mean) * rstd;
            o[j] = rstd * (fd[j] * gm[i][j] - s1 - xh * s2) + (dres ? r[j] : 0.f);
          }
          *reinterpret_cast<u32x4*>(dx + (size_t)row * D + (size_t)idx * 8) = pack8(o);
        }
      }
    }
    if (PF) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i) { vx[i] = nx[i]; vd[i] = nd[i]; vr[i] = nr[i]; }
      mean = nmean; rstd = nrstd;
    } else if (nrow < rows) {
      load_row<MAXV>(vx, x + (size_t)nrow * D, nv, lane);
      load_row<MAXV>(vd, dy + (size_t)nrow * D, nv, lane);
      if (dres && dx) load_row<MAXV>(vr, dres + (size_t)nrow * D, nv, lane);
      mean = mean_i[nrow]; rstd = rstd_i[nrow];
    }
  }
  if (dgamma) {
    __shared__ float red[2][MAXV * 512];
    for (int w = 0; w < NW; ++w) {
      if (wave == w) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
          int idx = lane + 64 * i;
          if (idx < nv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (w == 0) { red[0][idx * 8 + j] = ag[i][j]; red[1][idx * 8 + j] = ab[i][j]; }
              else { red[0][idx * 8 + j] += ag[i][j]; red[1][idx * 8 + j] += ab[i][j]; }
            }
          }
        }
      }
      __syncthreads();
    }
    for (int c = threadIdx.x; c < D; c += 64 * NW) {
      dgamma[(size_t)blockIdx.x * D + c] = red[0][c];
      dbeta[(size_t)blockIdx.x * D + c] = red[1][c];
    }
  }
}

template <int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* x, const bf16_t* w, bf16_t* y, float* rstd_o,
                                                          int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = D >> 3;
  u32x4 v[MAXV];
  load_row<MAXV>(v, x + (size_t)row * D, nv, lane);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) { float f[8]; unpack8(v[i], f);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += f[j] * f[j]; }
  const float rstd = rsqrtf(wave_sum(s) / (float)D + eps);
  if (lane == 0 && rstd_o) rstd_o[row] = rstd;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int idx = lane + 64 * i;
    if (idx < nv) {
      float f[8], ww[8];
      unpack8(v[i], f);
      unpack8(*reinterpret_cast<const u32x4*>(w + (size_t)idx * 8), ww);
      // tf:llama:67 — the normalised value is cast to bf16 BEFORE the weight multiply
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = ww[j] * bf2f(f2bf(f[j] * rstd));
      *reinterpret_cast<u32x4*>(y + (size_t)row * D + (size_t)idx * 8) = pack8(f);
    }
  }
}

template <int MAXV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MAXV <= 6 ? 4 : (MAXV <= 8 ? 2 : 1)))) void rmsnorm_bwd_kernel(const bf16_t* dy, const bf16_t* x, const bf16_t* w,
                                                          const float* rstd_i, const bf16_t* dres, bf16_t* dx,
                                                          int rows, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = D >> 3;
  u32x4 vx[MAXV], vd[MAXV];
  load_row<MAXV>(vx, x + (size_t)row * D, nv, lane);
  load_row<MAXV>(vd, dy + (size_t)row * D, nv, lane);
  // the residual gradient is requested with the other rows, not after the row reduction (one HBM
  // round trip per row instead of two)
  u32x4 vr[MAXV];
  if (dres) load_row<MAXV>(vr, dres + (size_t)row * D, nv, lane);
  const float rstd = rstd_i[row];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int idx = lane + 64 * i;
    if (idx < nv) {
      float fx[8], fd[8], ww[8];
      unpack8(vx[i], fx); unpack8(vd[i], fd);
      unpack8(*reinterpret_cast<const u32x4*>(w + (size_t)idx * 8), ww);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += fd[j] * ww[j] * fx[j] * rstd;
    }
  }
  s = wave_sum(s) / (float)D;
  // Only the packed rows stay live across the reduction: the empty asm makes them opaque, so the
  // compiler unpacks them again below instead of keeping 16 floats per 16-byte chunk in registers
  // (it did: 256 VGPRs + AGPRs, ONE wave per SIMD, 3.2 TB/s; now 4 waves per SIMD).
#pragma unroll
  for (int i = 0; i < MAXV; ++i) asm volatile("" : "+v"(vx[i]), "+v"(vd[i]));
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int idx = lane + 64 * i;
    if (idx < nv) {
      float fx[8], fd[8], ww[8], r[8], o[8];
      unpack8(vx[i], fx); unpack8(vd[i], fd);
      unpack8(*reinterpret_cast<const u32x4*>(w + (size_t)idx * 8), ww);
      if (dres) unpack8(vr[i], r);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = rstd * (fd[j] * ww[j] - fx[j] * rstd * s) + (dres ? r[j] : 0.f);
      *reinterpret_cast<u32x4*>(dx + (size_t)row * D + (size_t)idx * 8) = pack8(o);
    }
  }
}

// dw[c] += sum_rows dy[r][c] * bf16(x[r][c] * rstd[r])   (the weight multiplies the bf16-rounded
// normalised value, tf:llama:67).  Only trainable decoder layers need it (stage "pretrain": the
// translator layer), so it is its own small kernel rather than a branch of rmsnorm_bwd: one block per
// CU loops over rows with per-lane column sums in registers, block reduction through LDS, then the block's
// sums are STORED as part blockIdx.x of dw[nparts][D] (as layernorm_bwd does; the caller adds the parts in order).
template <int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_dweight_kernel(const bf16_t* dy, const bf16_t* x, const float* rstd_i,
                                                              float* dw, int rows, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = D >> 3;
  float ag[MAXV][8];
#pragma unroll
  for (int i = 0; i < MAXV; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) ag[i][j] = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    u32x4 vx[MAXV], vd[MAXV];
    load_row<MAXV>(vx, x + (size_t)row * D, nv, lane);
    load_row<MAXV>(vd, dy + (size_t)row * D, nv, lane);
    const float rstd = rstd_i[row];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (lane + 64 * i < nv) {
        float fx[8], fd[8];
        unpack8(vx[i], fx); unpack8(vd[i], fd);
#pragma unroll
        for (int j = 0; j < 8; ++j) ag[i][j] += fd[j] * bf2f(f2bf(fx[j] * rstd));
      }
    }
  }
  __shared__ float red[MAXV * 512];
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (w == 0) red[idx * 8 + j] = ag[i][j];
            else red[idx * 8 + j] += ag[i][j];
          }
        }
      }
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < D; c += 256) dw[(size_t)blockIdx.x * D + c] = red[c];     // this block's part
}

// A handful of rows (single-token decoding steps): one 256-thread block per row, same arithmetic.
__global__ __launch_bounds__(256) void rmsnorm_fwd_block_kernel(const bf16_t* x, const bf16_t* w, bf16_t* y,
                                                                float* rstd_o, int D, float eps) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x, nv = D >> 3;
  const bf16_t* xr = x + (size_t)row * D;
  float s = 0.f;
  for (int i = tid; i < nv; i += 256) {
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(xr + (size_t)i * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += f[j] * f[j];
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)D + eps);
  if (tid == 0 && rstd_o) rstd_o[row] = rstd;
  for (int i = tid; i < nv; i += 256) {
    float f[8], ww[8];
    unpack8(*reinterpret_cast<const u32x4*>(xr + (size_t)i * 8), f);
    unpack8(*reinterpret_cast<const u32x4*>(w + (size_t)i * 8), ww);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = ww[j] * bf2f(f2bf(f[j] * rstd));
    *reinterpret_cast<u32x4*>(y + (size_t)row * D + (size_t)i * 8) = pack8(f);
  }
}

inline bool bad_dim(int64_t rows, int64_t D) { return rows <= 0 || D <= 0 || (D & 7) || D > 8192 || rows > 0x7FFFFFFF; }

}  // namespace

#define DISPATCH_MAXV(D, CALL)                   \
  if ((D) <= 1024) { CALL(2); }                  \
  else if ((D) <= 2048) { CALL(4); }             \
  else if ((D) <= 3072) { CALL(6); }             \
  else if ((D) <= 4096) { CALL(8); }             \
  else { CALL(16); }

extern "C" int zk_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd,
                                int64_t rows, int64_t D, float eps, hipStream_t stream) {
  if (!x || !gamma || !beta || !y || bad_dim(rows, D)) return ZK_EINVAL;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define CALL(V) hipLaunchKernelGGL((layernorm_fwd_kernel<V>), grid, block, 0, stream, (const bf16_t*)x, \
    (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)y, mean, rstd, (int)rows, (int)D, eps)
  DISPATCH_MAXV(D, CALL)
#undef CALL
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

extern "C" int zk_layernorm_bwd(const void* dy, const void* x, const void* gamma, const float* mean, const float* rstd,
                                const void* dres, void* dx, float* dgamma, float* dbeta, int64_t nparts, int64_t rows,
                                int64_t D, hipStream_t stream) {
  if (!dy || !x || !gamma || !mean || !rstd || bad_dim(rows, D) || D > 4096) return ZK_EINVAL;
  if ((dgamma == nullptr) != (dbeta == nullptr)) return ZK_EINVAL;
  if (dgamma ? (nparts < 1 || nparts > ZK_NORM_MAX_PARTS) : false) return ZK_EINVAL;
  if (!dgamma) {                     // no affine gradients wanted: any grid will do
    nparts = (rows + 7) / 8;
    if (nparts > 512) nparts = 512;
  }
  dim3 grid((unsigned)nparts);
#define CALL(V, NW, PF) hipLaunchKernelGGL((layernorm_bwd_kernel<V, NW, PF>), grid, dim3(64 * NW), 0, stream, (const bf16_t*)dy, \
    (const bf16_t*)x, (const bf16_t*)gamma, mean, rstd, (const bf16_t*)dres, (bf16_t*)dx, dgamma, dbeta, (int)rows, (int)D)
  if (D <= 1024) { CALL(2, 8, true); } else { CALL(8, 4, false); }
#undef CALL
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

extern "C" int zk_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t rows, int64_t D, float eps,
                              hipStream_t stream) {
  if (!x || !w || !y || bad_dim(rows, D)) return ZK_EINVAL;
  if (rows <= 16) {   // decoding: a block per row instead of a single wave per row
    hipLaunchKernelGGL(rmsnorm_fwd_block_kernel, dim3((unsigned)rows), dim3(256), 0, stream, (const bf16_t*)x,
                       (const bf16_t*)w, (bf16_t*)y, rstd, (int)D, eps);
    ZK_CHECK_LAUNCH();
    return ZK_OK;
  }
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define CALL(V) hipLaunchKernelGGL((rmsnorm_fwd_kernel<V>), grid, block, 0, stream, (const bf16_t*)x, \
    (const bf16_t*)w, (bf16_t*)y, rstd, (int)rows, (int)D, eps)
  DISPATCH_MAXV(D, CALL)
#undef CALL
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

extern "C" int zk_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres,
                              void* dx, int64_t rows, int64_t D, hipStream_t stream) {
  if (!dy || !x || !w || !rstd || !dx || bad_dim(rows, D)) return ZK_EINVAL;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define CALL(V) hipLaunchKernelGGL((rmsnorm_bwd_kernel<V>), grid, block, 0, stream, (const bf16_t*)dy, \
    (const bf16_t*)x, (const bf16_t*)w, rstd, (const bf16_t*)dres, (bf16_t*)dx, (int)rows, (int)D)
  DISPATCH_MAXV(D, CALL)
#undef CALL
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

extern "C" int zk_rmsnorm_dweight(const void* dy, const void* x, const float* rstd, float* dweight, int64_t nparts,
                                  int64_t rows, int64_t D, hipStream_t stream) {
  if (!dy || !x || !rstd || !dweight || bad_dim(rows, D) || nparts < 1 || nparts > ZK_NORM_MAX_PARTS) return ZK_EINVAL;
  dim3 grid((unsigned)nparts), block(256);
#define CALL(V) hipLaunchKernelGGL((rmsnorm_dweight_kernel<V>), grid, block, 0, stream, (const bf16_t*)dy, \
    (const bf16_t*)x, rstd, dweight, (int)rows, (int)D)
  DISPATCH_MAXV(D, CALL)
#undef CALL
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}
