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
// Structure: block tile BM x BN x 64 with 16x16x32 bf16 MFMAs; three instantiations
//   256x256 (8 waves, 128x64 per wave), 256x128 (8 waves, 64x64), 128x128 (4 waves, 64x64),
//   128x96 (4 waves, 64x48: N = 3072 at M = 6144 is 1536 tiles = exactly 3 rounds of 512 slots),
// plus split-K (fp32 slabs in a caller-provided workspace + a finishing pass) for small outputs
// with long contractions (weight gradients, lm_head dgrad),
// chosen per problem by a tile-quantisation cost model (256 CUs, one 8-wave block or two
// 4-wave blocks per CU).  Operand tiles go HBM/L2 -> LDS directly with
// buffer_load_dwordx4 ... lds (LDS-DMA: no VGPR staging, no ds_write; out-of-range lanes are
// zero-filled by the buffer bounds check, which is how ragged M/N/K edges are handled).  The LDS
// image is lane-linear per wave-instruction (1 KiB), so the bank-conflict swizzle is applied to
// the per-lane SOURCE address and again on the read.  Two LDS stages: the DMA of K-tile t+1 is
// in flight while tile t feeds the MFMAs; one barrier per K-tile.
// k-contiguous tiles are [row][64k] images read with ds_read_b128; mn-contiguous tiles are
// [k][BM|BN] images read with ds_read_b64_tr_b16 (CDNA4 transposed LDS read), so dgrad/wgrad
// need no transposed copies in HBM.  The MFMA is issued with swapped operands (D = B_frag x
// A_frag) so each lane ends up with 4 consecutive n of one m: 8-byte C stores.
#include "common.h"
#include "../../include/zero_hip.h"

namespace {

constexpr int BK = 64;

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
  int splits, kchunk;      // split-K: block (tile, s) covers k in [s*kchunk, min(K, (s+1)*kchunk))
  float* slab;             // [splits][M][N] fp32 partial results (split-K only)
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

// LDS-DMA of one operand tile (R rows|cols x 64 k).  Slot s (16 bytes) of the image is filled by
// thread s % NT in its (s / NT)-th instruction; a wave-instruction fills 64 consecutive slots.
//   !T : image [R][64k], 8 slots per row; slot (row, pos) holds k-chunk  pos ^ (row & 7)
//    T : image [64k][R],  R/8 slots per k-row; slot (k, pos) holds mn-chunk  pos ^ swz_t(k)
template <bool T, int R, int NT>
__device__ __forceinline__ void tile_dma(__amdgpu_buffer_rsrc_t rs, unsigned char* lds, int ld, int mn0, int MN, int k0,
                                         int K, int tid) {
  constexpr int N = R * 8 / NT;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int s = tid + NT * i;
    unsigned int off;
    bool ok;
    if (!T) {
      const int row = s >> 3, kc = (s & 7) ^ (row & 7);
      ok = (mn0 + row < MN) && (k0 + kc * 8 < K);
      off = (unsigned int)(row * ld + k0 + kc * 8) * 2u;
    } else {
      constexpr int CPR = R / 8;
      const int kr = s / CPR, ch = (s % CPR) ^ swz_t(kr);
      ok = (k0 + kr < K) && (mn0 + ch * 8 < MN);
      off = (unsigned int)((k0 + kr) * ld + ch * 8) * 2u;
    }
    // wave-uniform LDS base of this wave-instruction; the hardware adds lane * 16
    unsigned char* dst = lds + ((tid & ~63) + NT * i) * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16,
                                             ok ? off : 0xFFFFFFF0u, 0, 0, 0);
  }
}

// Register-staged alternative for the transposed ([k][R]) image: global -> VGPR (issued before the
// MFMAs of the current tile) and VGPR -> LDS (after them).  Same image as tile_dma<true>.
template <int R, int NT>
__device__ __forceinline__ void tile_load_regs(u32x4 (&r)[R * 8 / NT], __amdgpu_buffer_rsrc_t rs, int ld, int mn0,
                                               int MN, int k0, int K, int tid) {
  constexpr int N = R * 8 / NT, CPR = R / 8;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int s = tid + NT * i;
    const int kr = s / CPR, ch = s % CPR;
    const bool ok = (k0 + kr < K) && (mn0 + ch * 8 < MN);
    r[i] = buf_load16(rs, ok ? (unsigned int)((k0 + kr) * ld + ch * 8) * 2u : 0xFFFFFFF0u);
  }
}
template <int R, int NT>
__device__ __forceinline__ void tile_store_regs(const u32x4 (&r)[R * 8 / NT], unsigned char* lds, int tid) {
  constexpr int N = R * 8 / NT, CPR = R / 8;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int s = tid + NT * i;
    const int kr = s / CPR, ch = s % CPR;
    *reinterpret_cast<u32x4*>(lds + kr * (R * 2) + ((ch ^ swz_t(kr)) << 4)) = r[i];
  }
}

// One 16(mn) x 32(k) MFMA operand fragment for k-step kk from the LDS image.
template <bool T, int R>
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
    short4_t lo = lds_read_tr16(lds + k0r * (R * 2) + ((ch ^ swz_t(k0r)) << 4) + sub);
    short4_t hi = lds_read_tr16(lds + k1r * (R * 2) + ((ch ^ swz_t(k1r)) << 4) + sub);
    union { struct { short4_t a, b; } s; bf16x8 v; } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
  }
}

template <bool TA, bool TB, int BM, int BN, int WAVES_M, int WAVES_N, int STG>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_kernel(GemmArgs g) {
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int TM = BM / WAVES_M / 16, TN = BN / WAVES_N / 16;
  constexpr int A_BYTES = BM * BK * 2, STAGE_BYTES = (BM + BN) * BK * 2;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // XCD-aware, bijective block remap (blocks b and b+8 share an XCD / L2), then a banded
  // tile order (8 m-tiles per band, m fastest) so co-resident blocks share B panels in L2.
  const int nwg = g.tiles_m * g.tiles_n;
  const int split = blockIdx.x / nwg;
  int bid = blockIdx.x - split * nwg;
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

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // STG == 1: transposed operands are staged through registers (ds_write), k-contiguous ones by DMA
  constexpr bool RA = TA && (STG == 1), RB = TB && (STG == 1);
  u32x4 ra[RA ? BM * 8 / NT : 1], rb[RB ? BN * 8 / NT : 1];
  const int kb = split * g.kchunk;
  const int Ke = min(g.K, kb + g.kchunk);
  const int nk = (Ke - kb + BK - 1) / BK;
  if constexpr (RA) { tile_load_regs<BM, NT>(ra, rA, g.lda, m0, g.M, kb, Ke, tid); tile_store_regs<BM, NT>(ra, smem, tid); }
  else tile_dma<TA, BM, NT>(rA, smem, g.lda, m0, g.M, kb, Ke, tid);
  if constexpr (RB) { tile_load_regs<BN, NT>(rb, rB, g.ldb, n0, g.N, kb, Ke, tid); tile_store_regs<BN, NT>(rb, smem + A_BYTES, tid); }
  else tile_dma<TB, BN, NT>(rB, smem + A_BYTES, g.ldb, n0, g.N, kb, Ke, tid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    const unsigned char* la = smem + (t & 1) * STAGE_BYTES;
    const unsigned char* lb = la + A_BYTES;
    unsigned char* nxt = smem + ((t + 1) & 1) * STAGE_BYTES;
    const bool more = t + 1 < nk;
    if (more) {
      if constexpr (RA) tile_load_regs<BM, NT>(ra, rA, g.lda, m0, g.M, kb + (t + 1) * BK, Ke, tid);
      else tile_dma<TA, BM, NT>(rA, nxt, g.lda, m0, g.M, kb + (t + 1) * BK, Ke, tid);
      if constexpr (RB) tile_load_regs<BN, NT>(rb, rB, g.ldb, n0, g.N, kb + (t + 1) * BK, Ke, tid);
      else tile_dma<TB, BN, NT>(rB, nxt + A_BYTES, g.ldb, n0, g.N, kb + (t + 1) * BK, Ke, tid);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[TM], fb[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = frag_load<TB, BN>(lb, wn * (TN * 16) + j * 16, kk, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = frag_load<TA, BM>(la, wm * (TM * 16) + i * 16, kk, lane);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    if (more) {
      if constexpr (RA) tile_store_regs<BM, NT>(ra, nxt, tid);
      if constexpr (RB) tile_store_regs<BN, NT>(rb, nxt + A_BYTES, tid);
    }
    // the DMA of tile t+1 must have landed (every wave waits for its own pieces) before any wave
    // reads it, and every wave must be done reading tile t before it is overwritten at t+2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- split-K: plain 16-byte fp32 stores into this split's slab; zk_splitk_finish sums the slabs
  if (g.splits > 1) {
    float* slab = g.slab + (size_t)split * g.M * g.N;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * (TM * 16) + i * 16 + (lane & 15);
      if (m >= g.M) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (TN * 16) + j * 16 + (lane >> 4) * 4;
        if (n >= g.N) continue;
        *reinterpret_cast<f32x4*>(slab + (size_t)m * g.N + n) = acc[i][j];
      }
    }
    return;
  }
  // ---- epilogue: lane holds C[m][n..n+3], m = l&15, n = 4*(l>>4) + reg ----
  float alpha = g.alpha;
  if (g.alpha_dev) alpha *= *g.alpha_dev;
  const int epi = g.epilogue;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * (TM * 16) + i * 16 + (lane & 15);
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (TN * 16) + j * 16 + (lane >> 4) * 4;
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

// C = bf16(alpha * sum_s slab[s] (+ C)), 4 columns per thread
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* slab, bf16_t* C, int M, int N, int ldc,
                                                            int splits, float alpha, const float* alpha_dev,
                                                            int accumulate) {
  if (alpha_dev) alpha *= *alpha_dev;
  const int n4 = N >> 2;
  const int64_t total = (int64_t)M * n4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int m = (int)(i / n4), n = (int)(i % n4) * 4;
    f32x4 a = *reinterpret_cast<const f32x4*>(slab + (size_t)m * N + n);
    for (int s = 1; s < splits; ++s) {
      f32x4 b = *reinterpret_cast<const f32x4*>(slab + ((size_t)s * M + m) * N + n);
      a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
    }
    bf16_t* cp = C + (size_t)m * ldc + n;
    float v[4] = {a[0] * alpha, a[1] * alpha, a[2] * alpha, a[3] * alpha};
    if (accumulate) {
      u32x2 o = *reinterpret_cast<const u32x2*>(cp);
      v[0] += lo_bf(o[0]); v[1] += hi_bf(o[0]); v[2] += lo_bf(o[1]); v[3] += hi_bf(o[1]);
    }
    u32x2 out;
    out[0] = pack2bf(v[0], v[1]);
    out[1] = pack2bf(v[2], v[3]);
    *reinterpret_cast<u32x2*>(cp) = out;
  }
}

}  // namespace

namespace {
int g_force_tile = -1;   // tuning hook: -1 = cost model, 0 = 128x128, 1 = 256x128, 2 = 256x256
int g_force_splits = 0;  // tuning hook: 0 = heuristic, 1 = never split K, n > 1 = force n splits
int g_stage_mode = 1;    // tuning hook: 0 = LDS-DMA for every operand, 1 = registers for transposed operands

template <int BM, int BN, int WM_, int WN_>
void launch_cfg(const GemmArgs& g0, int transA, int transB, hipStream_t stream) {
  GemmArgs g = g0;
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  dim3 grid(g.tiles_m * g.tiles_n * g.splits), block(64 * WM_ * WN_);
  const bool reg = g_stage_mode == 1;
#define ZK_LAUNCH(TA_, TB_, STG_) hipLaunchKernelGGL((gemm_kernel<TA_, TB_, BM, BN, WM_, WN_, STG_>), grid, block, 0, stream, g)
  if (!transA && !transB) ZK_LAUNCH(false, false, 0);
  else if (transA && !transB) { if (reg) ZK_LAUNCH(true, false, 1); else ZK_LAUNCH(true, false, 0); }
  else if constexpr ((BN & (BN - 1)) == 0) {   // the transposed-B image needs a power-of-two row (swizzle)
    if (!transA) { if (reg) ZK_LAUNCH(false, true, 1); else ZK_LAUNCH(false, true, 0); }
    else { if (reg) ZK_LAUNCH(true, true, 1); else ZK_LAUNCH(true, true, 0); }
  }
#undef ZK_LAUNCH
}

// Tile choice: time ~ rounds(tiles / concurrent blocks) x tile area / relative efficiency.
int choose_tile(int64_t M, int64_t N, int transB) {
  if (g_force_tile >= 0) return (g_force_tile == 3 && transB) ? 0 : g_force_tile;
  const int bm[4] = {128, 256, 256, 128}, bn[4] = {128, 128, 256, 96};
  const double slots[4] = {512.0, 256.0, 256.0, 512.0};   // two 4-wave blocks or one 8-wave block per CU
  const double eff[4] = {1.0, 0.85, 1.02, 0.93};          // measured on MI355X (tools/bench_gemm.py)
  int best = 0;
  double best_cost = 1e300;
  for (int c = 0; c < (transB ? 3 : 4); ++c) {
    double tiles = (double)((M + bm[c] - 1) / bm[c]) * (double)((N + bn[c] - 1) / bn[c]);
    double rounds = (double)(int64_t)((tiles + slots[c] - 1) / slots[c]);
    double cost = rounds * slots[c] * bm[c] * bn[c] / eff[c];
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}
}  // namespace

extern "C" int zk_gemm_set_tile(int cfg) {
  // cfg in [-1, 2] selects the tile; cfg = 10 / 11 selects the staging mode (0 / 1)
  if (cfg == 10 || cfg == 11) { g_stage_mode = cfg - 10; return ZK_OK; }
  if (cfg >= 100 && cfg <= 116) { g_force_splits = cfg - 100; return ZK_OK; }   // 100 = heuristic, 101 = off
  if (cfg < -1 || cfg > 3) return ZK_EINVAL;
  g_force_tile = cfg;
  return ZK_OK;
}

extern "C" int zk_gemm_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                            int64_t ldb, int64_t ldc, int transA, int transB, int epilogue, const void* bias,
                            const void* residual, int64_t ldr, void* aux_out, const void* aux_in, int64_t ldaux,
                            float alpha, const float* alpha_dev, int accumulate, void* workspace,
                            int64_t workspace_bytes, hipStream_t stream) {
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
  if ((transA ? K : 256) * lda * 2 >= 0x7FFFFFF0LL || (transB ? K : 256) * ldb * 2 >= 0x7FFFFFF0LL) return ZK_EINVAL;
  GemmArgs g;
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = (bf16_t*)C;
  g.bias = (const bf16_t*)bias; g.residual = (const bf16_t*)residual;
  g.aux_out = (bf16_t*)aux_out; g.aux_in = (const bf16_t*)aux_in;
  g.alpha_dev = alpha_dev; g.alpha = alpha;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.lda = (int)lda; g.ldb = (int)ldb; g.ldc = (int)ldc; g.ldr = (int)ldr; g.ldaux = (int)ldaux;
  g.epilogue = epilogue; g.accumulate = accumulate;
  g.tiles_m = g.tiles_n = 0;
  g.splits = 1; g.kchunk = (int)((K + BK - 1) / BK * BK); g.slab = nullptr;
  // split-K for small outputs with a long contraction (wgrad, lm_head dgrad): fill the 256 CUs
  const bool plain = epilogue == ZK_EPI_NONE && !bias && !residual && !aux_out;
  if (plain && workspace && g_force_splits != 1) {
    const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
    int64_t sp = g_force_splits > 1 ? g_force_splits : (tiles <= 300 && K >= 2048 ? 1024 / tiles : 1);
    if (sp > 16) sp = 16;
    while (sp > 1 && (K / sp < 512 || sp * M * N * 4 > workspace_bytes)) --sp;
    if (sp > 1) {
      g.splits = (int)sp;
      g.kchunk = (int)(((K + sp - 1) / sp + BK - 1) / BK * BK);
      g.splits = (int)((K + g.kchunk - 1) / g.kchunk);
      g.slab = (float*)workspace;
    }
  }
  if (g.splits > 1) {
    launch_cfg<128, 128, 2, 2>(g, transA, transB, stream);
    int64_t nb = (M * (N / 4) + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)nb), dim3(256), 0, stream, g.slab, g.C, g.M, g.N, g.ldc,
                       g.splits, alpha, alpha_dev, accumulate);
    ZK_CHECK_LAUNCH();
    return ZK_OK;
  }
  switch (choose_tile(M, N, transB)) {
    case 3: launch_cfg<128, 96, 2, 2>(g, transA, transB, stream); break;
    case 2: launch_cfg<256, 256, 2, 4>(g, transA, transB, stream); break;
    case 1: launch_cfg<256, 128, 4, 2>(g, transA, transB, stream); break;
    default: launch_cfg<128, 128, 2, 2>(g, transA, transB, stream); break;
  }
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}
