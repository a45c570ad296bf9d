// Bandwidth-bound kernels of the path: RoPE, SwiGLU, activation backward, bias-gradient
// column sums, the patch-embedding front end, row gather/splice, cross entropy on bf16
// logits, grad-norm + AdamW.  All use 16-byte (8 x bf16) coalesced accesses.
#include "common.h"
#include "../../include/zero_hip.h"

namespace {

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
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) t = fmaxf(t, red[i]);
  return t;
}

// ---------------------------------------------------------------- RoPE
// One thread rotates 8 (x1, x2) pairs: x1 = cols [c, c+8), x2 = cols [c + hd/2, ...).
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* x, const float* cos_t, const float* sin_t, int64_t total,
                                                   int S, int heads, int hd, int ld, int inverse) {
  const int half = hd >> 1;
  const int cpr = half >> 3;  // 16-byte chunks per half head
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % cpr);
    int64_t t = i / cpr;
    int h = (int)(t % heads);
    int64_t row = t / heads;
    int pos = (int)(row % S);
    bf16_t* p = x + row * ld + (int64_t)h * hd + c * 8;
    float a[8], b[8], cs[8], sn[8];
    unpack8(*reinterpret_cast<const u32x4*>(p), a);
    unpack8(*reinterpret_cast<const u32x4*>(p + half), b);
    const float* cp = cos_t + (int64_t)pos * half + c * 8;
    const float* sp = sin_t + (int64_t)pos * half + c * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { cs[j] = cp[j]; sn[j] = inverse ? -sp[j] : sp[j]; }
    float o1[8], o2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) rope_pair(a[j], b[j], cs[j], sn[j], o1[j], o2[j]);   // x1*cos - x2*sin, x2*cos + x1*sin
    *reinterpret_cast<u32x4*>(p) = pack8(o1);
    *reinterpret_cast<u32x4*>(p + half) = pack8(o2);
  }
}

// ---------------------------------------------------------------- SwiGLU
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* gu, bf16_t* h, int64_t total, int F) {
  const int cpr = F >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t row = i / cpr;
    int c = (int)(i % cpr);
    const bf16_t* g = gu + row * 2 * F + c * 8;
    float a[8], b[8], o[8];
    unpack8(*reinterpret_cast<const u32x4*>(g), a);
    unpack8(*reinterpret_cast<const u32x4*>(g + F), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float s = silu_fast(a[j]);
      o[j] = bf2f(f2bf(s)) * b[j];            // tf:llama:175: act_fn(gate) is a bf16 tensor
    }
    *reinterpret_cast<u32x4*>(h + row * F + c * 8) = pack8(o);
  }
}
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* dh, const bf16_t* gu, bf16_t* dgu, int64_t total,
                                                         int F) {
  const int cpr = F >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t row = i / cpr;
    int c = (int)(i % cpr);
    const bf16_t* g = gu + row * 2 * F + c * 8;
    float a[8], b[8], d[8], og[8], ou[8];
    unpack8(*reinterpret_cast<const u32x4*>(g), a);
    unpack8(*reinterpret_cast<const u32x4*>(g + F), b);
    unpack8(*reinterpret_cast<const u32x4*>(dh + row * F + c * 8), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float sg = sigmoid_fast(a[j]);
      float silu = a[j] * sg;
      og[j] = d[j] * b[j] * sg * (1.0f + a[j] * (1.0f - sg));
      ou[j] = d[j] * silu;
    }
    *reinterpret_cast<u32x4*>(dgu + row * 2 * F + c * 8) = pack8(og);
    *reinterpret_cast<u32x4*>(dgu + row * 2 * F + F + c * 8) = pack8(ou);
  }
}

// ---------------------------------------------------------------- activation backward
__global__ __launch_bounds__(256) void act_bwd_kernel(const bf16_t* dy, const bf16_t* pre, bf16_t* dx, int64_t nvec,
                                                      int kind) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    float d[8], x[8], o[8];
    unpack8(*reinterpret_cast<const u32x4*>(dy + i * 8), d);
    unpack8(*reinterpret_cast<const u32x4*>(pre + i * 8), x);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float g;
      if (kind == ZK_EPI_QUICK_GELU) {
        float s = sigmoid_fast(1.702f * x[j]);
        g = s * (1.0f + 1.702f * x[j] * (1.0f - s));
      } else {
        float cdf = 0.5f * (1.0f + erff(x[j] * 0.70710678118654752f));
        g = cdf + x[j] * 0.3989422804014327f * __expf(-0.5f * x[j] * x[j]);
      }
      o[j] = d[j] * g;
    }
    *reinterpret_cast<u32x4*>(dx + i * 8) = pack8(o);
  }
}

// ---------------------------------------------------------------- column sum (bias grads)
// block = 32 column-chunks (256 cols) x 8 row lanes; grid.y = nparts splits the rows.  Every block WRITES its
// partial sums to out[part][col] (no atomics: the result does not depend on arrival order); the caller adds the
// parts in index order (zk_f32_to_bf16_multi does, while converting).
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* dy, float* out, int rows, int N, int ld) {
  __shared__ float red[8][256 + 1];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int col = blockIdx.x * 256 + cx * 8;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (col < N) {
    for (int r = blockIdx.y * 8 + ry; r < rows; r += gridDim.y * 8) {
      float f[8];
      unpack8(*reinterpret_cast<const u32x4*>(dy + (size_t)r * ld + col), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += f[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[ry][cx * 8 + j] = acc[j];
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) t += red[r][threadIdx.x];
  const int oc = blockIdx.x * 256 + threadIdx.x;
  if (oc < N) out[(size_t)blockIdx.y * N + oc] = t;
}

// ---------------------------------------------------------------- patch embedding front end
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_t* px, bf16_t* cols, int64_t total, int H, int W, int P,
                                                     int Kpad) {
  const int G = W / P, Gy = H / P;
  const int cpr = Kpad >> 3;
  const int PP = P * P;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c8 = (int)(i % cpr);
    int64_t prow = i / cpr;           // b*Gy*G + gy*G + gx
    int gx = (int)(prow % G);
    int gy = (int)((prow / G) % Gy);
    int64_t b = prow / ((int64_t)G * Gy);
    bf16_t o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int k = c8 * 8 + j;
      bf16_t v = 0;
      if (k < 3 * PP) {
        int c = k / PP, rem = k - c * PP;
        int ii = rem / P, jj = rem - ii * P;
        v = px[((b * 3 + c) * H + (gy * P + ii)) * (int64_t)W + gx * P + jj];
      }
      o[j] = v;
    }
    *reinterpret_cast<u32x4*>(cols + prow * Kpad + c8 * 8) = *reinterpret_cast<u32x4*>(o);
  }
}
__global__ __launch_bounds__(256) void embed_finish_kernel(const bf16_t* patch, const bf16_t* cls, const bf16_t* pos,
                                                           bf16_t* out, int64_t total, int T, int D) {
  const int cpr = D >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % cpr);
    int64_t r = i / cpr;
    int t = (int)(r % T);
    int64_t b = r / T;
    const bf16_t* src = (t == 0) ? cls + c * 8 : patch + (b * (T - 1) + (t - 1)) * D + c * 8;
    float a[8], p[8];
    unpack8(*reinterpret_cast<const u32x4*>(src), a);
    unpack8(*reinterpret_cast<const u32x4*>(pos + (int64_t)t * D + c * 8), p);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += p[j];
    *reinterpret_cast<u32x4*>(out + r * D + c * 8) = pack8(a);
  }
}

// ---------------------------------------------------------------- row gather / splice
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* t0, const bf16_t* t1, const int32_t* codes,
                                                          bf16_t* dst, int64_t total, int D, int accumulate) {
  const int cpr = D >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % cpr);
    int64_t r = i / cpr;
    int code = codes[r];
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (code >= 0) v = *reinterpret_cast<const u32x4*>(t0 + (int64_t)code * D + c * 8);
    else if (code <= -2) v = *reinterpret_cast<const u32x4*>(t1 + (int64_t)(-2 - code) * D + c * 8);
    bf16_t* d = dst + r * D + c * 8;
    if (accumulate) {
      float a[8], b[8];
      unpack8(v, a);
      unpack8(*reinterpret_cast<const u32x4*>(d), b);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += b[j];
      v = pack8(a);
    }
    *reinterpret_cast<u32x4*>(d) = v;
  }
}

// ---------------------------------------------------------------- cross entropy
// One 256-thread block per row.  Pass 1: online max / sum-exp (fp32).  Pass 2 (re-read, L2
// resident: a 128256-wide bf16 row is 250 KiB): write softmax - onehot as bf16.
__global__ __launch_bounds__(256) void cross_entropy_kernel(const bf16_t* logits, const int32_t* targets,
                                                            float* loss_sum, float* row_loss, bf16_t* dlogits, int V,
                                                            int ld) {
  __shared__ float red[8];
  const int row = blockIdx.x;
  const int tgt = targets[row];
  const bf16_t* x = logits + (size_t)row * ld;
  bf16_t* dx = dlogits ? dlogits + (size_t)row * ld : nullptr;
  const int nv = V >> 3;
  if (tgt == -100) {
    if (row_loss && threadIdx.x == 0) row_loss[row] = -1.0f;       // marks "ignored" for the ordered sum
    if (dx) for (int i = threadIdx.x; i < nv; i += 256) *reinterpret_cast<u32x4*>(dx + (size_t)i * 8) = u32x4{0u, 0u, 0u, 0u};
    return;
  }
  // dlogits may alias logits: the target logit is read HERE, before any wave can have reached pass 2
  // (block_sum has no trailing barrier, so a read after pass 1 would race with other waves' stores).
  const float xt = bf2f(x[tgt]);
  float m = -INFINITY, s = 0.f;
  for (int i = threadIdx.x; i < nv; i += 256) {
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(x + (size_t)i * 8), f);
    float lm = f[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) lm = fmaxf(lm, f[j]);
    float nm = fmaxf(m, lm);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += __expf(f[j] - nm);
    s = s * __expf(m - nm) + acc;
    m = nm;
  }
  const float gm = block_max(m, red);
  s = (m == -INFINITY) ? 0.f : s * __expf(m - gm);
  const float gs = block_sum(s, red);
  if (threadIdx.x == 0) {
    const float l = logf(gs) + gm - xt;      // >= 0 up to rounding; clamped so that -1 can mark ignored rows.
    // NOT fmaxf: it returns the non-NaN operand, and a NaN / inf row must reach the loss (ref engine:315 raises on it)
    if (row_loss) row_loss[row] = (l < 0.f) ? 0.f : l;
    else { atomicAdd(loss_sum, l); atomicAdd(loss_sum + 1, 1.0f); }
  }
  if (dx) {
    const float inv = 1.0f / gs;
    for (int i = threadIdx.x; i < nv; i += 256) {
      float f[8];
      unpack8(*reinterpret_cast<const u32x4*>(x + (size_t)i * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f[j] = __expf(f[j] - gm) * inv;
        if (i * 8 + j == tgt) f[j] -= 1.0f;
      }
      *reinterpret_cast<u32x4*>(dx + (size_t)i * 8) = pack8(f);
    }
  }
}

// Ordered sum of the per-row losses: one block, thread t adds rows t, t + 256, ... in order, then a fixed tree.
__global__ __launch_bounds__(256) void ce_finish_kernel(const float* row_loss, float* loss_sum, int rows) {
  __shared__ float sl[256], sc[256];
  float l = 0.f, c = 0.f;
  for (int r = threadIdx.x; r < rows; r += 256) {
    const float v = row_loss[r];
    if (v != -1.0f) { l += v; c += 1.0f; }     // -1 = ignored row; a NaN row compares unequal and is added
  }
  sl[threadIdx.x] = l; sc[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sc[threadIdx.x] += sc[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { loss_sum[0] += sl[0]; loss_sum[1] += sc[0]; }
}

// ---------------------------------------------------------------- grad norm + AdamW
__global__ __launch_bounds__(256) void sumsq_kernel(const bf16_t* g, float* out, float* parts, int64_t n, float scale) {
  __shared__ float red[8];
  float acc = 0.f;
  const int64_t nv = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(g + i * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) { float t = f[j] * scale; acc += t * t; }
  }
  if (blockIdx.x == 0)
    for (int64_t i = nv * 8 + threadIdx.x; i < n; i += 256) { float t = bf2f(g[i]) * scale; acc += t * t; }
  float t = block_sum(acc, red);
  if (threadIdx.x == 0) {
    if (parts) parts[blockIdx.x] = t;        // ordered finish below: the clip coefficient is reproducible
    else atomicAdd(out, t);
  }
}
__global__ __launch_bounds__(256) void sumsq_finish_kernel(const float* parts, float* out, int n) {
  __shared__ float sl[256];
  float l = 0.f;
  for (int r = threadIdx.x; r < n; r += 256) l += parts[r];
  sl[threadIdx.x] = l;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sl[threadIdx.x] += sl[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] += sl[0];
}

__global__ __launch_bounds__(256) void adamw_kernel(bf16_t* p, const bf16_t* g, float* m, float* v, int64_t n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2s,
                                                    float max_norm, const float* sumsq, float gscale) {
  float coef = gscale;
  if (max_norm > 0.f && sumsq) {
    float c = max_norm / (sqrtf(*sumsq) + 1e-6f);
    coef *= fminf(c, 1.0f);
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = bf2f(g[i]) * coef;
    float pi = bf2f(p[i]);
    pi *= (1.0f - lr * wd);
    float mi = b1 * m[i] + (1.0f - b1) * gi;
    float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    float denom = sqrtf(vi) / bc2s + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = f2bf(pi);
  }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* src, bf16_t* dst, int64_t n, float scale,
                                                          int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = src[i] * scale;
    if (accumulate) v += bf2f(dst[i]);
    dst[i] = f2bf(v);
  }
}

// Up to 16 independent (fp32 source, bf16 destination, length, accumulate) segments in ONE launch: the
// vector gradients of a layer (biases, norm affines) each live in their own slot of the flat gradient
// buffer; blockIdx.y picks the segment, the table travels in the kernel arguments.
struct CvtSegs {
  const float* src[16];
  bf16_t* dst[16];
  int n[16];
  int accumulate[16];
  int nparts[16];       // src holds nparts rows of n values, `stride` elements apart: added in index order
  long long stride[16];
};
// One block = 64 columns x 16 part groups: thread (c, g) adds parts g, g + 16, ... of its column (4 loads in
// flight), the 16 group sums meet in LDS and are added in group order: a FIXED tree, so the result is
// bit-reproducible, without one thread walking hundreds of parts serially.
__global__ __launch_bounds__(1024) void f32_to_bf16_multi_kernel(CvtSegs t, float scale) {
  __shared__ float red[16][64];
  const int s = blockIdx.y;
  const float* src = t.src[s];
  bf16_t* dst = t.dst[s];
  const int n = t.n[s], acc = t.accumulate[s], np = t.nparts[s];
  const long long st = t.stride[s];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  for (int c0 = blockIdx.x * 64; c0 < n; c0 += gridDim.x * 64) {
    const int i = c0 + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < n) {
      int p = g;
      for (; p + 48 < np; p += 64) {
        a0 += src[(size_t)p * st + i];
        a1 += src[(size_t)(p + 16) * st + i];
        a2 += src[(size_t)(p + 32) * st + i];
        a3 += src[(size_t)(p + 48) * st + i];
      }
      for (; p < np; p += 16) a0 += src[(size_t)p * st + i];
    }
    red[g][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && i < n) {
      float v = red[0][c];
#pragma unroll
      for (int k = 1; k < 16; ++k) v += red[k][c];
      v *= scale;
      if (acc) v += bf2f(dst[i]);
      dst[i] = f2bf(v);
    }
    __syncthreads();
  }
}

inline unsigned grid_for(int64_t work_items, int64_t cap = 256 * 16) {
  int64_t b = (work_items + 255) / 256;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

}  // namespace

extern "C" int zk_abi_version(void) { return ZK_ABI_VERSION; }

extern "C" int zk_rope(void* x, const float* cos_t, const float* sin_t, int64_t rows, int64_t S, int64_t heads,
                       int64_t head_dim, int64_t ld, int inverse, hipStream_t stream) {
  if (!x || !cos_t || !sin_t || rows <= 0 || S <= 0 || heads <= 0 || (head_dim & 15) || (ld & 7)) return ZK_EINVAL;
  int64_t total = rows * heads * (head_dim / 16);
  hipLaunchKernelGGL(rope_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (bf16_t*)x, cos_t, sin_t, total, (int)S,
                     (int)heads, (int)head_dim, (int)ld, inverse);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

extern "C" int zk_swiglu_fwd(const void* gu, void* h, int64_t rows, int64_t F, hipStream_t stream) {
  if (!gu || !h || rows <= 0 || F <= 0 || (F & 7)) return ZK_EINVAL;
  int64_t total = rows * (F / 8);
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16_t*)gu, (bf16_t*)h,
                     total, (int)F);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}
extern "C" int zk_swiglu_bwd(const void* dh, const void* gu, void* dgu, int64_t rows, int64_t F, hipStream_t stream) {
  if (!dh || !gu || !dgu || rows <= 0 || F <= 0 || (F & 7)) return ZK_EINVAL;
  int64_t total = rows * (F / 8);
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16_t*)dh,
                     (const bf16_t*)gu, (bf16_t*)dgu, total, (int)F);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

extern "C" int zk_act_bwd(const void* dy, const void* pre, void* dx, int64_t n, int kind, hipStream_t stream) {
  if (!dy || !pre || !dx || n <= 0 || (n & 7)) return ZK_EINVAL;
  if (kind != ZK_EPI_QUICK_GELU && kind != ZK_EPI_GELU_ERF) return ZK_EINVAL;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16_t*)dy,
                     (const bf16_t*)pre, (bf16_t*)dx, n / 8, kind);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

extern "C" int zk_colsum(const void* dy, float* out, int64_t rows, int64_t N, int64_t ld, int64_t nparts,
                         hipStream_t stream) {
  if (!dy || !out || rows <= 0 || N <= 0 || (N & 7) || (ld & 7) || nparts <= 0 || nparts > 65535) return ZK_EINVAL;
  dim3 grid((unsigned)((N + 255) / 256), (unsigned)nparts);
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, stream, (const bf16_t*)dy, out, (int)rows, (int)N, (int)ld);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

extern "C" int zk_patch_im2col(const void* pixels, void* cols, int64_t B, int64_t H, int64_t W, int64_t P, int64_t Kpad,
                               hipStream_t stream) {
  if (!pixels || !cols || B <= 0 || P <= 0 || H % P || W % P || (Kpad & 7) || Kpad < 3 * P * P) return ZK_EINVAL;
  int64_t total = B * (H / P) * (W / P) * (Kpad / 8);
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16_t*)pixels, (bf16_t*)cols,
                     total, (int)H, (int)W, (int)P, (int)Kpad);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}
extern "C" int zk_embed_finish(const void* patch, const void* cls, const void* pos, void* out, int64_t B, int64_t T,
                               int64_t D, hipStream_t stream) {
  if (!patch || !cls || !pos || !out || B <= 0 || T <= 1 || (D & 7)) return ZK_EINVAL;
  int64_t total = B * T * (D / 8);
  hipLaunchKernelGGL(embed_finish_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16_t*)patch,
                     (const bf16_t*)cls, (const bf16_t*)pos, (bf16_t*)out, total, (int)T, (int)D);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

// dst[c][r] = src[r][c]: 64 x 64 tiles through LDS (rows padded to 66 elements: the column reads of the second
// phase step 33 dwords, an odd stride, across the banks), 16-byte global loads, 8-byte global stores.  Layout
// preparation for the k-contiguous GEMM variant: the trainable weights are transposed once per optimizer step
// (ATen's strided copy moves ~0.4 TB/s here; this kernel is HBM-rate).
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* src, bf16_t* dst, int R, int C, int lds_, int ldd) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {              // 64 rows x 8 chunks of 8 elements = 512 chunks, two per thread
    const int idx = t + 256 * i, r = idx >> 3, ch = idx & 7;
    u32x4 v = u32x4{0u, 0u, 0u, 0u};
    if (r0 + r < R && c0 + ch * 8 < C) v = *reinterpret_cast<const u32x4*>(src + (size_t)(r0 + r) * lds_ + c0 + ch * 8);
    unsigned int* p = reinterpret_cast<unsigned int*>(&tile[r][ch * 8]);
    p[0] = v[0]; p[1] = v[1]; p[2] = v[2]; p[3] = v[3];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {              // 64 output rows (= source columns) x 16 chunks of 4 elements
    const int idx = t + 256 * i, c = idx >> 4, q = idx & 15;
    if (c0 + c < C && r0 + q * 4 < R) {
      u32x2 o;
      o[0] = (unsigned int)tile[q * 4][c] | ((unsigned int)tile[q * 4 + 1][c] << 16);
      o[1] = (unsigned int)tile[q * 4 + 2][c] | ((unsigned int)tile[q * 4 + 3][c] << 16);
      *reinterpret_cast<u32x2*>(dst + (size_t)(c0 + c) * ldd + r0 + q * 4) = o;
    }
  }
}

extern "C" int zk_transpose_bf16(const void* src, void* dst, int64_t R, int64_t C, int64_t ld_src, int64_t ld_dst,
                                 hipStream_t stream) {
  if (!src || !dst || R <= 0 || C <= 0 || (R & 7) || (C & 7) || (ld_src & 7) || (ld_dst & 7) || ld_src < C || ld_dst < R)
    return ZK_EINVAL;
  if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) return ZK_EINVAL;
  if ((C + 63) / 64 > 0x7FFFFFFF || (R + 63) / 64 > 65535) return ZK_EINVAL;
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64)), dim3(256), 0, stream,
                     (const bf16_t*)src, (bf16_t*)dst, (int)R, (int)C, (int)ld_src, (int)ld_dst);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

extern "C" int zk_gather_rows(const void* table0, const void* table1, const int32_t* codes, void* dst, int64_t rows,
                              int64_t D, int accumulate, hipStream_t stream) {
  if (!codes || !dst || rows <= 0 || D <= 0 || (D & 7)) return ZK_EINVAL;
  int64_t total = rows * (D / 8);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16_t*)table0,
                     (const bf16_t*)table1, codes, (bf16_t*)dst, total, (int)D, accumulate);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

extern "C" int zk_cross_entropy(const void* logits, const int32_t* targets, float* loss_sum, float* row_loss,
                                void* dlogits, int64_t rows, int64_t V, int64_t ld, hipStream_t stream) {
  if (!logits || !targets || !loss_sum || rows <= 0 || rows > 0x7FFFFFFF || V <= 0 || (V & 7) || (ld & 7)) return ZK_EINVAL;
  hipLaunchKernelGGL(cross_entropy_kernel, dim3((unsigned)rows), dim3(256), 0, stream, (const bf16_t*)logits, targets,
                     loss_sum, row_loss, (bf16_t*)dlogits, (int)V, (int)ld);
  if (row_loss) hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, stream, (const float*)row_loss, loss_sum, (int)rows);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

extern "C" int zk_sumsq(const void* g, float* out, float* parts, int64_t n, float scale, hipStream_t stream) {
  if (!g || !out || n <= 0) return ZK_EINVAL;
  const unsigned nb = grid_for(n / 8 + 1, ZK_SUMSQ_PARTS);
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, stream, (const bf16_t*)g, out, parts, n, scale);
  if (parts) hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, stream, (const float*)parts, out, (int)nb);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

extern "C" int zk_adamw_step(void* p, const void* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int64_t step, float max_norm, const float* sumsq,
                             float grad_scale, hipStream_t stream) {
  if (!p || !g || !m || !v || n <= 0 || step < 1) return ZK_EINVAL;
  double bc1 = 1.0 - pow((double)beta1, (double)step);
  double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, stream, (bf16_t*)p, (const bf16_t*)g, m, v, n, lr,
                     beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), max_norm, sumsq, grad_scale);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

extern "C" int zk_f32_to_bf16(const float* src, void* dst, int64_t n, float scale, int accumulate, hipStream_t stream) {
  if (!src || !dst || n <= 0) return ZK_EINVAL;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, stream, src, (bf16_t*)dst, n, scale,
                     accumulate);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

extern "C" int zk_f32_to_bf16_multi(const float* const* src, void* const* dst, const int64_t* n, const int* accumulate,
                                    const int* nparts, const int64_t* part_stride, int count, float scale,
                                    hipStream_t stream) {
  if (!src || !dst || !n || !accumulate || count <= 0 || count > 16) return ZK_EINVAL;
  CvtSegs t;
  int64_t longest = 0;
  for (int i = 0; i < 16; ++i) {
    const int j = i < count ? i : 0;
    if (!src[j] || !dst[j] || n[j] <= 0 || n[j] > 0x7FFFFFFF) return ZK_EINVAL;
    t.src[i] = src[j]; t.dst[i] = (bf16_t*)dst[j]; t.n[i] = (int)n[j]; t.accumulate[i] = accumulate[j];
    t.nparts[i] = nparts ? nparts[j] : 1;
    t.stride[i] = part_stride ? part_stride[j] : n[j];
    if (t.nparts[i] < 1 || t.stride[i] < n[j]) return ZK_EINVAL;
    if (n[j] > longest) longest = n[j];
  }
  hipLaunchKernelGGL(f32_to_bf16_multi_kernel, dim3((unsigned)((longest + 63) / 64 > 64 ? 64 : (longest + 63) / 64), (unsigned)count), dim3(1024), 0, stream, t, scale);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}
