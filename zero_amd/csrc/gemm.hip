// bf16 MFMA GEMM for gfx950:  C[m][n] = epilogue( alpha * sum_k A(m,k) * B(n,k) )
//
// Replaces the ATen/rocBLAS linear calls that transformers dispatches on the reference's
// hot path (tf:clip:293-296,333,346-350; tf:llama:174-176,230-241,280,480) and their
// autograd dgrad/wgrad counterparts.
//
// Operand layouts (all bf16, row-major with leading dimension in elements):
//   transA == 0 : A is [M][lda], k contiguous          (activations x in y = x W^T)
//   transA == 1 : A is [K][lda], m contiguous          (dY^T in dW = dY^T x, stored as dY [K=tokens][M=out])
//   transB == 0 : B is [N][ldb], k contiguous          (nn.Linear weight [out,in] in y = x W^T)
//   transB == 1 : B is [K][ldb], n contiguous          (W [K=out][N=in] in dx = dy W;  x in dW = dY^T x)
//
// Tiling: 128x128x64 block tile, 256 threads = 4 waves (2x2), each wave 64x64 as 4x4
// v_mfma_f32_16x16x32_bf16 tiles.  Operand tiles are staged HBM -> VGPR -> LDS
// (double-buffered, the next tile's loads are in flight during the MFMAs).  k-contiguous
// tiles are stored as [row][64k] with a 16-byte-chunk XOR swizzle and read with
// ds_read_b128; mn-contiguous tiles are stored as [k][128] and read with
// ds_read_b64_tr_b16 (the CDNA4 transposed LDS read), so the dgrad/wgrad variants need no
// transposed copies in HBM.  The MFMA is issued with the operands swapped (D = B_frag x
// A_frag) so each lane ends up with 4 consecutive n of one m: 8-byte C stores.
#include "common.h"
#include "../../include/zero_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KiB

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  bf16_t* C;
  const bf16_t* bias;      // [N] or null
  const bf16_t* residual;  // [M][ldr] or null
  bf16_t* aux_out;         // [M][ldc] pre-activation copy, or null
  const bf16_t* aux_in;    // [M][ldaux] for the *_DACT epilogues
  const float* alpha_dev;  // optional device scalar multiplied into the accumulator
  float alpha;
  int M, N, K;
  int lda, ldb, ldc, ldr, ldaux;
  int epilogue;
  int accumulate;  // C += result
  int tiles_m, tiles_n;
};

__device__ __forceinline__ int swz_t(int k) { return ((k & 3) << 2) | ((k >> 2) & 3); }

__device__ __forceinline__ float act_quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float act_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dact_quick_gelu(float x) {
  float s = 1.0f / (1.0f + __expf(-1.702f * x));
  return s * (1.0f + 1.702f * x * (1.0f - s));
}
__device__ __forceinline__ float dact_gelu_erf(float x) {
  float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// Issue the 4 x 16-byte global loads of one operand tile for this thread.
template <bool T>
__device__ __forceinline__ void tile_load(u32x4 (&r)[4], __amdgpu_buffer_rsrc_t rs, int ld, int mn0, int MN,
                                          int k0, int K, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int c = tid + 256 * i;
    unsigned int off;
    bool ok;
    if (!T) {
      int row = c >> 3, kc = c & 7;
      ok = (mn0 + row < MN) && (k0 + kc * 8 < K);
      off = (unsigned int)(row * ld + k0 + kc * 8) * 2u;
    } else {
      int kr = c >> 4, ch = c & 15;
      ok = (k0 + kr < K) && (mn0 + ch * 8 < MN);
      off = (unsigned int)((k0 + kr) * ld + ch * 8) * 2u;
    }
    r[i] = buf_load16(rs, ok ? off : 0xFFFFFFF0u);
  }
}

template <bool T>
__device__ __forceinline__ void tile_store_lds(const u32x4 (&r)[4], unsigned char* lds, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int c = tid + 256 * i;
    int addr;
    if (!T) {
      int row = c >> 3, kc = c & 7;
      addr = row * 128 + ((kc ^ (row & 7)) << 4);
    } else {
      int kr = c >> 4, ch = c & 15;
      addr = kr * 256 + ((ch ^ swz_t(kr)) << 4);
    }
    *reinterpret_cast<u32x4*>(lds + addr) = r[i];
  }
}

// One 16(mn) x 32(k) MFMA operand fragment for k-step kk from the LDS tile.
template <bool T>
__device__ __forceinline__ bf16x8 frag_load(const unsigned char* lds, int mn_base, int kk, int lane) {
  if (!T) {
    int row = mn_base + (lane & 15);
    int chunk = kk * 4 + (lane >> 4);
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((chunk ^ (row & 7)) << 4));
  } else {
    int i = lane & 15, g = lane >> 4;
    int p = i & 3, q = i >> 2;
    int mn = mn_base + 4 * p;
    int ch = mn >> 3;
    int sub = (mn & 7) * 2;
    int k0r = kk * 32 + 8 * g + q;
    int k1r = k0r + 4;
    short4_t lo = lds_read_tr16(lds + k0r * 256 + ((ch ^ swz_t(k0r)) << 4) + sub);
    short4_t hi = lds_read_tr16(lds + k1r * 256 + ((ch ^ swz_t(k1r)) << 4) + sub);
    union { struct { short4_t a, b; } s; bf16x8 v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware, bijective block remap (blocks b and b+8 share an XCD / L2), then a banded
  // tile order (8 m-tiles per band, m fastest) so co-resident blocks share B panels in L2.
  const int nwg = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int band = bid / (8 * g.tiles_n);
  const int band_m = min(8, g.tiles_m - band * 8);
  const int within = bid - band * 8 * g.tiles_n;
  const int tm = band * 8 + within % band_m;
  const int tn = within / band_m;
  const int m0 = tm * BM, n0 = tn * BN;

  // Per-block descriptors: the base is moved to the tile origin so 32-bit offsets suffice.
  const bf16_t* Ab = TA ? g.A + m0 : g.A + (size_t)m0 * g.lda;
  const bf16_t* Bb = TB ? g.B + n0 : g.B + (size_t)n0 * g.ldb;
  __amdgpu_buffer_rsrc_t rA = make_rsrc(Ab, 0x7FFFFFF0u);
  __amdgpu_buffer_rsrc_t rB = make_rsrc(Bb, 0x7FFFFFF0u);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[4], rb[4];
  const int nk = (g.K + BK - 1) / BK;
  tile_load<TA>(ra, rA, g.lda, m0, g.M, 0, g.K, tid);
  tile_load<TB>(rb, rB, g.ldb, n0, g.N, 0, g.K, tid);
  tile_store_lds<TA>(ra, smem, tid);
  tile_store_lds<TB>(rb, smem + BM * BK * 2, tid);
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    unsigned char* cur = smem + (t & 1) * STAGE_BYTES;
    unsigned char* nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
    const bool more = (t + 1 < nk);
    if (more) {
      tile_load<TA>(ra, rA, g.lda, m0, g.M, (t + 1) * BK, g.K, tid);
      tile_load<TB>(rb, rB, g.ldb, n0, g.N, (t + 1) * BK, g.K, tid);
    }
    const unsigned char* la = cur;
    const unsigned char* lb = cur + BM * BK * 2;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = frag_load<TA>(la, wm * 64 + i * 16, kk, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = frag_load<TB>(lb, wn * 64 + j * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    if (more) {
      tile_store_lds<TA>(ra, nxt, tid);
      tile_store_lds<TB>(rb, nxt + BM * BK * 2, tid);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m][n..n+3], m = l&15, n = 4*(l>>4) + reg ----
  float alpha = g.alpha;
  if (g.alpha_dev) alpha *= *g.alpha_dev;
  const int epi = g.epilogue;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
      if (n >= g.N) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * alpha;
      if (g.bias) {
        u32x2 b = *reinterpret_cast<const u32x2*>(g.bias + n);
        v[0] += lo_bf(b[0]); v[1] += hi_bf(b[0]); v[2] += lo_bf(b[1]); v[3] += hi_bf(b[1]);
      }
      if (epi == ZK_EPI_QUICK_GELU || epi == ZK_EPI_GELU_ERF) {
        // the reference rounds the pre-activation to bf16 before the activation
        u32x2 pre;
        pre[0] = pack2bf(v[0], v[1]);
        pre[1] = pack2bf(v[2], v[3]);
        if (g.aux_out) *reinterpret_cast<u32x2*>(g.aux_out + (size_t)m * g.ldc + n) = pre;
        v[0] = lo_bf(pre[0]); v[1] = hi_bf(pre[0]); v[2] = lo_bf(pre[1]); v[3] = hi_bf(pre[1]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (epi == ZK_EPI_QUICK_GELU) ? act_quick_gelu(v[e]) : act_gelu_erf(v[e]);
      } else if (epi == ZK_EPI_MUL_DQUICK_GELU || epi == ZK_EPI_MUL_DGELU_ERF) {
        u32x2 h = *reinterpret_cast<const u32x2*>(g.aux_in + (size_t)m * g.ldaux + n);
        float x[4] = {lo_bf(h[0]), hi_bf(h[0]), lo_bf(h[1]), hi_bf(h[1])};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] *= (epi == ZK_EPI_MUL_DQUICK_GELU) ? dact_quick_gelu(x[e]) : dact_gelu_erf(x[e]);
      }
      if (g.residual) {
        // x = residual + bf16(f(x)) : the branch output is rounded before the add
        u32x2 r = *reinterpret_cast<const u32x2*>(g.residual + (size_t)m * g.ldr + n);
        v[0] = bf2f(f2bf(v[0])) + lo_bf(r[0]); v[1] = bf2f(f2bf(v[1])) + hi_bf(r[0]);
        v[2] = bf2f(f2bf(v[2])) + lo_bf(r[1]); v[3] = bf2f(f2bf(v[3])) + hi_bf(r[1]);
      }
      bf16_t* cp = g.C + (size_t)m * g.ldc + n;
      if (g.accumulate) {
        u32x2 o = *reinterpret_cast<const u32x2*>(cp);
        v[0] += lo_bf(o[0]); v[1] += hi_bf(o[0]); v[2] += lo_bf(o[1]); v[3] += hi_bf(o[1]);
      }
      u32x2 out;
      out[0] = pack2bf(v[0], v[1]);
      out[1] = pack2bf(v[2], v[3]);
      *reinterpret_cast<u32x2*>(cp) = out;
    }
  }
}

}  // namespace

extern "C" int zk_gemm_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                            int64_t ldb, int64_t ldc, int transA, int transB, int epilogue, const void* bias,
                            const void* residual, int64_t ldr, void* aux_out, const void* aux_in, int64_t ldaux,
                            float alpha, const float* alpha_dev, int accumulate, hipStream_t stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return ZK_EINVAL;
  if ((N & 7) || (ldc & 3)) return ZK_EINVAL;
  if ((!transA || !transB) && (K & 7)) return ZK_EINVAL;  // k-contiguous operands are read in 8-element chunks
  if (transA ? ((M & 7) || (lda & 7)) : (lda & 7)) return ZK_EINVAL;
  if (ldb & 7) return ZK_EINVAL;
  if (epilogue < 0 || epilogue > ZK_EPI_MUL_DGELU_ERF) return ZK_EINVAL;
  if ((epilogue == ZK_EPI_MUL_DQUICK_GELU || epilogue == ZK_EPI_MUL_DGELU_ERF) && !aux_in) return ZK_EINVAL;
  if (residual && (ldr & 3)) return ZK_EINVAL;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) return ZK_EINVAL;
  if (reinterpret_cast<uintptr_t>(C) & 7) return ZK_EINVAL;
  // 32-bit byte offsets inside one block tile
  if ((transA ? K : 128) * lda * 2 >= 0x7FFFFFF0LL || (transB ? K : 128) * ldb * 2 >= 0x7FFFFFF0LL) return ZK_EINVAL;
  GemmArgs g;
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = (bf16_t*)C;
  g.bias = (const bf16_t*)bias; g.residual = (const bf16_t*)residual;
  g.aux_out = (bf16_t*)aux_out; g.aux_in = (const bf16_t*)aux_in;
  g.alpha_dev = alpha_dev; g.alpha = alpha;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.lda = (int)lda; g.ldb = (int)ldb; g.ldc = (int)ldc; g.ldr = (int)ldr; g.ldaux = (int)ldaux;
  g.epilogue = epilogue; g.accumulate = accumulate;
  g.tiles_m = (int)((M + BM - 1) / BM);
  g.tiles_n = (int)((N + BN - 1) / BN);
  dim3 grid(g.tiles_m * g.tiles_n), block(256);
  if (!transA && !transB) hipLaunchKernelGGL((gemm_kernel<false, false>), grid, block, 0, stream, g);
  else if (!transA && transB) hipLaunchKernelGGL((gemm_kernel<false, true>), grid, block, 0, stream, g);
  else if (transA && transB) hipLaunchKernelGGL((gemm_kernel<true, true>), grid, block, 0, stream, g);
  else hipLaunchKernelGGL((gemm_kernel<true, false>), grid, block, 0, stream, g);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}
