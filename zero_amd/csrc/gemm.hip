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
#include <type_traits>
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
  int band;                // m-tiles per band of the tile order (persistent ring kernel)
  int phase_delay;         // experiment: start delay (in 10 ns ticks) per phase step of the persistent kernel
  int splits, kchunk;      // split-K: block (tile, s) covers k in [s*kchunk, min(K, (s+1)*kchunk))
  float* slab;             // [splits][M][N] fp32 partial results (split-K only)
};

__device__ __forceinline__ int swz_t(int k) { return ((k & 3) << 2) | ((k >> 2) & 3); }

__device__ __forceinline__ float act_quick_gelu(float x) { return x * sigmoid_fast(1.702f * x); }
__device__ __forceinline__ float act_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dact_quick_gelu(float x) {
  float s = sigmoid_fast(1.702f * x);
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
template <bool T, int R, int NT, int ONLY = -1>   // ONLY >= 0: just that one of the R * 8 / NT instructions
__device__ __forceinline__ void tile_dma(__amdgpu_buffer_rsrc_t rs, unsigned char* lds, int ld, int mn0, int MN, int k0,
                                         int K, int tid, int row0 = 0) {
  constexpr int N = R * 8 / NT;
#pragma unroll
  for (int i = (ONLY >= 0 ? ONLY : 0); i < (ONLY >= 0 ? ONLY + 1 : N); ++i) {
    const int s = tid + NT * i;
    unsigned int off;
    bool ok;
    if (!T) {
      const int row = s >> 3, kc = (s & 7) ^ (row & 7);
      ok = (mn0 + row0 + row < MN) && (k0 + kc * 8 < K);
      off = (unsigned int)((row0 + row) * ld + k0 + kc * 8) * 2u;
    } else {
      constexpr int CPR = R / 8;
      const int kr = s / CPR, ch = (s % CPR) ^ swz_t(kr);
      ok = (k0 + kr < K) && (mn0 + ch * 8 < MN);
      off = (unsigned int)((k0 + kr) * ld + ch * 8) * 2u;
    }
    // wave-uniform LDS base of this wave-instruction; the hardware adds lane * 16
    unsigned char* dst = lds + ((tid & ~63) + NT * i) * 16;
    // (transposed images are read with ds_read_b64_tr_b16: the asm form keeps the compiler from draining the DMA in
    //  front of the first such read, see lds_dma)
    lds_dma<16, T>(rs, dst, ok ? off : 0xFFFFFFF0u);
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

// Epilogue.  The MFMA leaves lane l with C[m = l&15][n = 4*(l>>4) .. +3] of each 16x16 fragment:
// stored from there a wave-instruction touches 16 rows x 32 bytes, and the write-back of a 256x256
// tile took ~25 us (measured: 0.8-2.3 TB/s chip-wide, a third of the kernel at K = 3072).  So each
// wave transposes its accumulators through a private LDS patch (32 rows x TN*16 fp32, padded): a
// lane then owns 8 consecutive n of one row, a wave-instruction writes whole 128-byte rows, and
// bias / residual / aux operands are read 16 bytes per lane.  `wl` = this wave's patch
// (kEpiPatchBytes, only used after the block's last operand read; the caller has synchronised).
constexpr int kEpiPatchBytes = 32 * (4 * 16 + 4) * 4;

template <int TM, int TN>
__device__ __forceinline__ void store_tile(const GemmArgs& g, f32x4 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                           int lane, int split, float* wl) {
  static_assert(TM % 2 == 0 && TN <= 4, "patch is 32 rows x up to 64 columns");
  constexpr int COLS = TN * 16, LD = COLS + 4, CPRW = COLS / 8, RPP = 64 / CPRW, NPASS = (32 + RPP - 1) / RPP;
  const int rrow = lane / CPRW, rch = lane % CPRW;
  const bool lane_on = lane < RPP * CPRW;
  const int n = n0 + wn * COLS + rch * 8;
  const bool n_ok = lane_on && n < g.N;
  float alpha = g.alpha;
  if (g.alpha_dev) alpha *= *g.alpha_dev;
  const int epi = g.epilogue;
  float bias[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (g.bias && n_ok && g.splits == 1) {
    u32x4 b = *reinterpret_cast<const u32x4*>(g.bias + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) { bias[2 * e] = lo_bf(b[e]); bias[2 * e + 1] = hi_bf(b[e]); }
  }
  float* slab = g.splits > 1 ? g.slab + (size_t)split * g.M * g.N : nullptr;
#pragma unroll
  for (int p = 0; p < TM / 2; ++p) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        *reinterpret_cast<f32x4*>(wl + (ii * 16 + (lane & 15)) * LD + j * 16 + (lane >> 4) * 4) = acc[2 * p + ii][j];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int row = ps * RPP + rrow;
      const int m = m0 + wm * (TM * 16) + p * 32 + row;
      if (!n_ok || row >= 32 || m >= g.M) continue;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(wl + row * LD + rch * 8);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(wl + row * LD + rch * 8 + 4);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if (slab) {   // split-K: fp32 partial sums; splitk_finish_kernel applies alpha / accumulate
        *reinterpret_cast<f32x4*>(slab + (size_t)m * g.N + n) = lo;
        *reinterpret_cast<f32x4*>(slab + (size_t)m * g.N + n + 4) = hi;
        continue;
      }
      if (epi == ZK_EPI_SWIGLU_BWD) {
        // v = dh for features n..n+7 (rounded to bf16 as the unfused path stores it); gate / up of
        // the forward sit in the interleaved layout at column (n / 32) * 64 + n % 32 (+ 32)
        const int col = ((n >> 5) << 6) + (n & 31);
        const u32x4 gq = *reinterpret_cast<const u32x4*>(g.aux_in + (size_t)m * g.ldaux + col);
        const u32x4 uq = *reinterpret_cast<const u32x4*>(g.aux_in + (size_t)m * g.ldaux + col + 32);
        u32x4 og, ou;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float r0[2], r1[2];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const float gg = hh ? hi_bf(gq[e]) : lo_bf(gq[e]);
            const float uu = hh ? hi_bf(uq[e]) : lo_bf(uq[e]);
            const float d = bf2f(f2bf(v[2 * e + hh]));
            const float sg = sigmoid_fast(gg);
            r0[hh] = d * uu * sg * (1.0f + gg * (1.0f - sg));
            r1[hh] = d * (gg * sg);
          }
          og[e] = pack2bf(r0[0], r0[1]);
          ou[e] = pack2bf(r1[0], r1[1]);
        }
        *reinterpret_cast<u32x4*>(g.C + (size_t)m * g.ldc + col) = og;
        *reinterpret_cast<u32x4*>(g.C + (size_t)m * g.ldc + col + 32) = ou;
        continue;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * alpha + bias[e];
      if (epi == ZK_EPI_QUICK_GELU || epi == ZK_EPI_GELU_ERF) {
        // the reference rounds the pre-activation to bf16 before the activation
        u32x4 pre;
#pragma unroll
        for (int e = 0; e < 4; ++e) pre[e] = pack2bf(v[2 * e], v[2 * e + 1]);
        if (g.aux_out) *reinterpret_cast<u32x4*>(g.aux_out + (size_t)m * g.ldc + n) = pre;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] = lo_bf(pre[e]); v[2 * e + 1] = hi_bf(pre[e]); }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (epi == ZK_EPI_QUICK_GELU) ? act_quick_gelu(v[e]) : act_gelu_erf(v[e]);
      } else if (epi == ZK_EPI_MUL_DQUICK_GELU || epi == ZK_EPI_MUL_DGELU_ERF) {
        const u32x4 h = *reinterpret_cast<const u32x4*>(g.aux_in + (size_t)m * g.ldaux + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = (e & 1) ? hi_bf(h[e >> 1]) : lo_bf(h[e >> 1]);
          v[e] *= (epi == ZK_EPI_MUL_DQUICK_GELU) ? dact_quick_gelu(x) : dact_gelu_erf(x);
        }
      }
      if (g.residual) {
        // x = residual + bf16(f(x)) : the branch output is rounded before the add
        const u32x4 r = *reinterpret_cast<const u32x4*>(g.residual + (size_t)m * g.ldr + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e])) + ((e & 1) ? hi_bf(r[e >> 1]) : lo_bf(r[e >> 1]));
      }
      bf16_t* cp = g.C + (size_t)m * g.ldc + n;
      if (g.accumulate) {
        const u32x4 o = *reinterpret_cast<const u32x4*>(cp);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (e & 1) ? hi_bf(o[e >> 1]) : lo_bf(o[e >> 1]);
      }
      u32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = pack2bf(v[2 * e], v[2 * e + 1]);
      if (g.C) *reinterpret_cast<u32x4*>(cp) = out;     // (null C: ZK_EPI_SWIGLU without the gate|up copy)
    }
    if constexpr (TN == 4) {
      if (epi == ZK_EPI_SWIGLU) {
        // second pass over the same patch: this wave's 64 columns are one interleave block
        // [32 gate | 32 up]; 4 lanes per row, 8 features each: h = bf16(silu(gate)) * up
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int row = ps * 16 + (lane >> 2), c = lane & 3;
          const int m = m0 + wm * (TM * 16) + p * 32 + row;
          const int nb = n0 + wn * 64;
          if (m >= g.M || nb >= g.N) continue;
          const float* pr = wl + row * LD + c * 8;
          u32x4 o4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float r[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const float gg = bf2f(f2bf(pr[2 * e + hh])), uu = bf2f(f2bf(pr[32 + 2 * e + hh]));
              const float sl = silu_fast(gg);
              r[hh] = bf2f(f2bf(sl)) * uu;      // tf:llama:175: act_fn(gate) is a bf16 tensor
            }
            o4[e] = pack2bf(r[0], r[1]);
          }
          *reinterpret_cast<u32x4*>(g.aux_out + (size_t)m * g.ldaux + (nb >> 1) + c * 8) = o4;
        }
      }
    }
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

  // (the loop's last barrier has passed: no wave reads operand tiles any more)
  store_tile<TM, TN>(g, acc, m0, n0, wm, wn, lane, split, reinterpret_cast<float*>(smem + wave * kEpiPatchBytes));
}

// ------------------------------------------------------------------------------------------
// NT ring variant: NSTAGE LDS stages, the DMA of K-tile t+2 is issued before the MFMAs of tile t
// and only tile t+1 is waited for (counted s_waitcnt vmcnt, raw s_barrier), so two tiles of HBM/L2
// latency are covered instead of one.  All LDS in ONE array (no compiler-inserted vmcnt(0)).
// ------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, int MODE = 0>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void gemm_ring_kernel(GemmArgs g) {
  constexpr int NT = 64 * WAVES_M * WAVES_N;
  constexpr int TM = BM / WAVES_M / 16, TN = BN / WAVES_N / 16;
  constexpr int A_BYTES = BM * BK * 2, STAGE_BYTES = (BM + BN) * BK * 2;
  constexpr int PER = (BM + BN) * 8 / NT;      // DMA instructions per thread per K-tile
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTAGE * STAGE_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
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
  __amdgpu_buffer_rsrc_t rA = make_rsrc(g.A + (size_t)m0 * g.lda, 0x7FFFFFF0u);
  __amdgpu_buffer_rsrc_t rB = make_rsrc(g.B + (size_t)n0 * g.ldb, 0x7FFFFFF0u);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kb = split * g.kchunk;
  const int Ke = min(g.K, kb + g.kchunk);
  const int nk = (Ke - kb + BK - 1) / BK;
  auto issue = [&](int t) {
    unsigned char* st = smem + (t % NSTAGE) * STAGE_BYTES;
    tile_dma<false, BM, NT>(rA, st, g.lda, m0, g.M, kb + t * BK, Ke, tid);
    tile_dma<false, BN, NT>(rB, st + A_BYTES, g.ldb, n0, g.N, kb + t * BK, Ke, tid);
  };
  issue(0);
  if (nk > 1) {
    issue(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  // Fragment double buffering: the ds_reads of the NEXT k-half are issued before the MFMAs of the
  // current one, so LDS latency hides behind 16 MFMAs of the same wave.
  bf16x8 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
  auto load_frags = [&](bf16x8 (&fa)[TM], bf16x8 (&fb)[TN], int t, int kk) {
    const unsigned char* la = smem + (t % NSTAGE) * STAGE_BYTES;
    const unsigned char* lb = la + A_BYTES;
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = frag_load<false, BN>(lb, wn * (TN * 16) + j * 16, kk, lane);
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = frag_load<false, BM>(la, wm * (TM * 16) + i * 16, kk, lane);
  };
  auto mfma_all = [&](bf16x8 (&fa)[TM], bf16x8 (&fb)[TN]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  load_frags(fa0, fb0, 0, 0);
  for (int t = 0; t < nk; ++t) {
    if (MODE != 2 && t + 2 < nk) issue(t + 2);
    if (MODE != 1) {
      load_frags(fa1, fb1, t, 1);
      mfma_all(fa0, fb0);
    }
    if (t + 1 < nk) {
      // tile t+1 (issued one iteration ago) must have landed; tile t+2 stays in flight
      if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (MODE != 1) load_frags(fa0, fb0, t + 1, 0);
    }
    if (MODE != 1) mfma_all(fa1, fb1);
  }
  __syncthreads();   // every wave has its last fragments in registers: LDS becomes epilogue scratch
  store_tile<TM, TN>(g, acc, m0, n0, wm, wn, lane, split, reinterpret_cast<float*>(smem + wave * kEpiPatchBytes));
}

// ------------------------------------------------------------------------------------------
// NT slot ring, the kernel for the large products.  Measured on MI355X, LDS-DMA from L2 tops out
// near 55-70 GB/s per CU, so operand bytes per flop (not the MFMA rate) bound the 128x128 and
// 256x128 tiles near 1.0-1.2 PF: a 256x256 tile needs 7.8 B/kflop (256x128: 11.7, 128x128: 15.6).
// Its 64 KiB K-tiles do not fit three times in 160 KiB of LDS, so the ring is cut in SLOTS of RS
// rows x 64 k: BM/RS slots of A then BN/RS slots of B per K-tile, NSLOT slots in LDS.
//   256x256 -> RS 256, 5 slots: A_t B_t A_t+1 B_t+1 A_t+2; when tile t is consumed its two slots
//              take B_t+2 and A_t+3: one barrier per K-tile, 64-96 KiB of DMA always queued.
//   384x192 -> RS 192, 6 slots: M = 6144, N = 3072 is exactly 256 such tiles, one per CU, at the
//              same 7.8 B/kflop (288 tiles of 256x256 would run a second, 12 %-full round).
// 8 waves.  Registers: accumulators + A fragments refilled right after the last MFMA that reads
// each (rolling), B fragments double-buffered: ds_read latency hides behind the MFMAs.  The
// steady-state K loop is straight-line code (tail steps are separate instantiations): with the
// tail cases merged in, the compiler's s_waitcnt placement assumes the shortest LDS queue and
// waits for just-issued reads.  MODE 1/2/3 are ablations (no MFMA / no DMA / no epilogue).
// ------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, int RS, int NSLOT, int MODE, bool LEAN>
__global__ __launch_bounds__(512) void gemm_ringx_kernel(GemmArgs g) {
  constexpr int NT = 512;
  static_assert(WAVES_M * WAVES_N == 8, "8 waves");
  constexpr int TM = BM / WAVES_M / 16, TN = BN / WAVES_N / 16;
  constexpr int SA = BM / RS, SB = BN / RS, SPT = SA + SB;
  static_assert(SA * RS == BM && SB * RS == BN && NSLOT >= 2 * SPT && NSLOT - 2 * SPT <= SPT, "slot geometry");
  static_assert((TM * 16) <= RS && RS % (TM * 16) == 0 && RS % (TN * 16) == 0, "a wave's rows lie in one slot");
  constexpr int SLOT = RS * BK * 2;
  constexpr int PER = RS * 8 / NT;             // DMA instructions per thread per slot
  static_assert(PER * NT == RS * 8, "slot rows");
  constexpr int AHEAD = NSLOT - 2 * SPT;       // slots that may stay in flight when tile t+1 is needed
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NSLOT * SLOT];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
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
  __amdgpu_buffer_rsrc_t rA = make_rsrc(g.A + (size_t)m0 * g.lda, 0x7FFFFFF0u);
  __amdgpu_buffer_rsrc_t rB = make_rsrc(g.B + (size_t)n0 * g.ldb, 0x7FFFFFF0u);

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (g.K + BK - 1) / BK;
  const int total = nk * SPT;                  // slot loads of the whole K loop, in ring order
  // slot load q = (K-tile q / SPT, part q % SPT); parts 0..SA-1 are A row blocks, SA.. are B's.
  // LEAN (host: M % BM == 0, N % BN == 0, K % 64 == 0 - every large product of the step): the
  // per-lane source offset is ONE register per operand for the whole kernel (instruction i of a
  // slot reads 64 rows further down: a scalar offset), the LDS address is scalar too, so a K-tile's
  // DMA costs ~3 scalar instructions per load instead of ~12 VALU ones between the MFMA bursts.
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const unsigned voff_a = (unsigned)((tid >> 3) * g.lda + (((tid & 7) ^ ((tid >> 3) & 7)) * 8)) * 2u;
  const unsigned voff_b = (unsigned)((tid >> 3) * g.ldb + (((tid & 7) ^ ((tid >> 3) & 7)) * 8)) * 2u;
  auto issue = [&](int q) {
    const int t = q / SPT, part = q - t * SPT;
    unsigned char* dst = smem + (q % NSLOT) * SLOT;
    if constexpr (LEAN) {
      const bool is_a = part < SA;
      const int ld = is_a ? g.lda : g.ldb;
      const int row0 = (is_a ? part : part - SA) * RS;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        unsigned char* d = dst + (wv * 64 + NT * i) * 16;
        const int soff = ((row0 + 64 * i) * ld + t * BK) * 2;
        if (is_a) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)d, 16, voff_a, soff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)d, 16, voff_b, soff, 0, 0);
      }
    } else if (part < SA) {
      tile_dma<false, RS, NT>(rA, dst, g.lda, m0, g.M, t * BK, g.K, tid, part * RS);
    } else {
      tile_dma<false, RS, NT>(rB, dst, g.ldb, n0, g.N, t * BK, g.K, tid, (part - SA) * RS);
    }
  };
  auto wait_slots = [&](int nslots) {          // at most nslots slot loads may remain in flight
    switch (nslots) {
      case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PER) : "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PER) : "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * PER) : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * PER) : "memory"); break;
    }
  };
  static_assert(NSLOT - SPT <= 4 && 4 * PER <= 63, "wait_slots covers the prologue");
  {
    const int first = min(NSLOT, total);
    for (int q = 0; q < first; ++q) issue(q);
    wait_slots(first - SPT);
  }
  __builtin_amdgcn_s_barrier();

  // LDS addresses of this wave's fragments inside K-tile t
  const int a_part = wm * (TM * 16) / RS, a_row = wm * (TM * 16) - a_part * RS;
  const int b_part = SA + wn * (TN * 16) / RS, b_row = wn * (TN * 16) % RS;
  auto lds_a = [&](int t) { return smem + ((t * SPT + a_part) % NSLOT) * SLOT; };
  auto lds_b = [&](int t) { return smem + ((t * SPT + b_part) % NSLOT) * SLOT; };

  bf16x8 fa[TM], fb0[TN], fb1[TN];
  // One k-half: MFMAs on (fa, fcur) while the fragments of the NEXT k-half arrive: B's into the
  // other buffer up front, each A fragment in place as soon as its last MFMA has been issued.
  auto burst = [&](bf16x8 (&fcur)[TN], bf16x8 (&fnext)[TN], const unsigned char* na, const unsigned char* nb, int nkk,
                   auto has_next) {
    constexpr bool NX = decltype(has_next)::value;
    if constexpr (NX) {
#pragma unroll
      for (int j = 0; j < TN; ++j) fnext[j] = frag_load<false, RS>(nb, b_row + j * 16, nkk, lane);
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fcur[j], fa[i], acc[i][j], 0, 0, 0);
      // (scheduling fence: keep the refill BEHIND the MFMAs that read the old value, or the
      //  compiler hoists it into fresh registers and the kernel no longer fits 256 VGPRs)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NX) fa[i] = frag_load<false, RS>(na, a_row + i * 16, nkk, lane);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  // One K-tile; STEADY = every refill of this step exists and tile t+2 exists (straight-line code)
  auto step = [&](int t, auto steady, auto has_next) {
    constexpr bool ST = decltype(steady)::value, HN = decltype(has_next)::value;
    if (MODE != 1) burst(fb0, fb1, lds_a(t), lds_b(t), 1, T_{});
    if constexpr (HN) {
      // tile t+1 must have landed and every wave must hold its last fragments of tile t before the
      // slots of tile t are refilled
      if (ST) wait_slots(AHEAD);
      else wait_slots(t + 2 < nk ? AHEAD : 0);
      __builtin_amdgcn_s_barrier();
      if (MODE != 2) {
#pragma unroll
        for (int r = 0; r < SPT; ++r) {
          const int q = t * SPT + NSLOT + r;
          if (ST || q < total) issue(q);
        }
      }
      if (MODE != 1) burst(fb1, fb0, lds_a(t + 1), lds_b(t + 1), 0, T_{});
    } else {
      if (MODE != 1) burst(fb1, fb0, nullptr, nullptr, 0, F_{});
    }
  };
  if (MODE != 1) {
#pragma unroll
    for (int j = 0; j < TN; ++j) fb0[j] = frag_load<false, RS>(lds_b(0), b_row + j * 16, 0, lane);
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = frag_load<false, RS>(lds_a(0), a_row + i * 16, 0, lane);
  }
  int t = 0;
  for (; (t + 1) * SPT + NSLOT <= total && t + 2 < nk; ++t) step(t, T_{}, T_{});
  for (; t + 1 < nk; ++t) step(t, F_{}, T_{});
  step(t, F_{}, F_{});
  if (MODE == 3) {   // experiment: no epilogue (one conditional store keeps the accumulators alive)
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sum == 123.456f) g.C[0] = f2bf(sum);
    return;
  }
  __syncthreads();   // every wave has its last fragments in registers: LDS becomes epilogue scratch
  store_tile<TM, TN>(g, acc, m0, n0, wm, wn, lane, 0, reinterpret_cast<float*>(smem + wave * kEpiPatchBytes));
}

// ------------------------------------------------------------------------------------------
// Persistent NT slot ring ("pring"): the kernel for the large NT products.
// Measured on the ringx kernel above (tools/ring_modes.py, 6144 x 16384 x 3072): of 95 us per
// 256x256 tile, 66 us are the K loop, ~9 us the hand-over between two workgroups on a CU (LDS
// release, dispatch, argument loads, the first DMA round trip with an empty ring) and 12-20 us the
// epilogue, during which the CU's matrix pipes idle and 256 CUs write in the same few microseconds.
// Here ONE workgroup per CU walks its tiles (id, id + grid, ...: the same XCD-aware order as before)
// and the DMA ring never drains: slot loads form one stream over (tile, K-tile, part), so while the
// last K-tiles of a tile are being multiplied the first K-tiles of the NEXT tile are already landing.
// The epilogue uses a small dedicated LDS patch (2 KiB per wave, XOR-swizzled, 16 rows x 32 columns
// per pass) instead of the ring's memory, and its global stores are fire-and-forget: the next
// tile's first K-tile needs no vmcnt wait (its data was waited for BEFORE the first store), so the
// stores drain behind the next tile's MFMAs.
//   256x256: RS 128, 9 slots of 16 KiB (2.25 K-tiles) + 16 KiB patch = 160 KiB
//   384x192: RS 192, 6 slots of 24 KiB (2 K-tiles)    + 16 KiB patch = 160 KiB, waves 4(M) x 2(N)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int patch_swz(int r) { return ((r >> 1) & 3) | ((r & 1) << 2); }

// Epilogue of one wave's (TM*16) x (TN*16) sub-tile through its private 16 x 32 fp32 LDS patch.
// Write side: lane (r = l & 15, g = l >> 4) owns C[r][4g..4g+3] of a 16x16 fragment; read side: lane
// (row = l >> 2, q = l & 3) gets 8 consecutive columns of one row: a wave-instruction stores 16 rows
// x 64 bytes.  Chunk c (16 bytes) of row r lives at chunk c ^ patch_swz(r): conflict-free both ways.
// EPI selects the code that is compiled in: 0 = alpha / bias; 6 = 0 + residual; 1 = the
// quick-GELU pair (forward with pre-activation copy, backward multiply); 4 = the erf-GELU pair; 2 = SwiGLU forward;
// 5 = SwiGLU backward; 3 = nothing (ablation).
//
// Every global access of the passes is a BUFFER instruction through a descriptor that ends at the operand's last row
// (M * ld * 2 bytes; 0 bytes for an operand the call does not have): rows past M, masked columns (offset 2 GiB) and
// absent operands are out of range for the hardware — loads return zeros, stores are dropped — so the TM * NJ passes
// are ONE straight line of code without a single branch around a memory instruction.  That is what lets the operands
// an epilogue READS (saved gate|up, residual, old C, saved pre-activation) be requested AUXD passes ahead: vmcnt
// counts in order, and only in branch-free code can the compiler (and the hand-placed wait below) know how many
// younger requests stand behind a load.  With the loads and stores in `if (row < M)` blocks, as they were through
// round 2, every wait the compiler placed was vmcnt(0): each pass of the residual / SwiGLU-backward epilogues drained
// the stores of the pass before it and the next tile's DMA (profiles/r03_gemm_epilogue_branchfree.txt).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t epi_rsrc(const void* p, int rows, int ld) {
  return make_rsrc(p, p ? (unsigned)rows * (unsigned)ld * 2u : 0u);
}
__device__ __forceinline__ u32x4 epi_load(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
}
__device__ __forceinline__ void epi_store(__amdgpu_buffer_rsrc_t r, unsigned off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);
}

template <int TM, int TN, int WAITN, int EPI>
__device__ __forceinline__ void store_tile_p(const GemmArgs& g, f32x4 (&acc)[TM][TN], int mw, int nw, int lane,
                                             unsigned char* patch) {
  static_assert(TN % 2 == 0, "two fragments per pass");
  constexpr int NJ = TN / 2, NPASS = TM * NJ;
  constexpr int NLD = EPI == 5 ? 2 : (EPI == 1 || EPI == 4 || EPI == 6) ? 1 : 0;       // register loads per pass
  // request distance, in passes (4 VGPRs per operand and pass): the 384x192 tile holds 144 accumulators and the next
  // tile's 72 fragment registers through the epilogue, which leaves room for 16
  constexpr int AUXD = (TM * TN > 32 && NLD == 2) ? 2 : 4;
  static_assert(WAITN + NLD * AUXD <= 63, "vmcnt field");
  constexpr unsigned OOB = 0x80000000u;                  // past every descriptor (host: operands under 2 GiB)
  // An opaque copy of the lane id: the patch addresses and row / column indices below are loop-invariant, and hoisted
  // out of the persistent tile loop they stay live across the K loop — the 384x192 tile (144 accumulators + 72 fragment
  // registers) then spills, and the spill reloads sit in the first K-step of every tile behind a vmcnt(0).  Recomputing
  // them per tile costs ~10 vector instructions.
  asm volatile("" : "+v"(lane));
  const int wr = lane & 15, wg = lane >> 4, rr = lane >> 2, rq = lane & 3;
  unsigned char* wp0 = patch + wr * 128 + ((wg ^ patch_swz(wr)) << 4);
  unsigned char* wp1 = patch + wr * 128 + (((4 + wg) ^ patch_swz(wr)) << 4);
  const unsigned char* rp0 = patch + rr * 128 + (((2 * rq) ^ patch_swz(rr)) << 4);
  const unsigned char* rp1 = patch + rr * 128 + (((2 * rq + 1) ^ patch_swz(rr)) << 4);
  const int epi = g.epilogue;
  const bool fwd_act = epi == ZK_EPI_QUICK_GELU || epi == ZK_EPI_GELU_ERF;              // classes 1 / 4: forward member
  float alpha = g.alpha;
  if (g.alpha_dev) alpha *= *g.alpha_dev;
  u32x4 biasq[NJ];                          // kept packed (4 registers per 8 columns), unpacked where it is added
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    biasq[jj] = u32x4{0u, 0u, 0u, 0u};
    const int n = nw + jj * 32 + rq * 8;
    if ((EPI < 2 || EPI == 4 || EPI == 6) && g.bias && n < g.N) biasq[jj] = *reinterpret_cast<const u32x4*>(g.bias + n);
  }
  const __amdgpu_buffer_rsrc_t rC = epi_rsrc(g.C, g.M, g.ldc);
  // the operands of the two prefetched loads (A, B) and of the second store (X), by class
  const __amdgpu_buffer_rsrc_t rA = EPI == 5   ? epi_rsrc(g.aux_in, g.M, g.ldaux)
                                    : EPI == 6 ? epi_rsrc(g.residual, g.M, g.ldr)
                                               : epi_rsrc(fwd_act ? nullptr : g.aux_in, g.M, g.ldaux);
  const __amdgpu_buffer_rsrc_t rX = EPI == 2 ? epi_rsrc(g.aux_out, g.M, g.ldaux)
                                             : epi_rsrc(fwd_act ? g.aux_out : nullptr, g.M, g.ldc);
  const int ld_a = EPI == 6 ? g.ldr : g.ldaux;
  u32x4 pa[NLD >= 1 ? AUXD : 1], pb[NLD >= 2 ? AUXD : 1];
  auto prefetch = [&](auto p_c) __attribute__((always_inline)) {
    constexpr int P = decltype(p_c)::value;
    if constexpr (NLD > 0 && P < NPASS) {
      const int m = mw + (P / NJ) * 16 + rr, n = nw + (P % NJ) * 32 + rq * 8;
      const bool ok = m < g.M && n < g.N;
      if constexpr (EPI == 5) {               // saved gate|up, interleaved in blocks of 32: [32 gate | 32 up]
        const unsigned off = ((unsigned)m * (unsigned)g.ldaux + (unsigned)(((n >> 5) << 6) + (n & 31))) * 2u;
        pa[P % AUXD] = epi_load(rA, ok ? off : OOB);
        pb[P % AUXD] = epi_load(rA, ok ? off + 64u : OOB);
      } else {
        pa[P % AUXD] = epi_load(rA, ok ? ((unsigned)m * (unsigned)ld_a + (unsigned)n) * 2u : OOB);
      }
    }
  };
  static_for<AUXD>([&](auto p_c) { prefetch(p_c); });
  static_for<TM>([&](auto i_c) {
    constexpr int i = decltype(i_c)::value;
    const int m = mw + i * 16 + rr;
    float gate[8];
    static_for<NJ>([&](auto jj_c) {
      constexpr int jj = decltype(jj_c)::value;
      constexpr int P = i * NJ + jj;
      *reinterpret_cast<f32x4*>(wp0) = acc[i][2 * jj];
      *reinterpret_cast<f32x4*>(wp1) = acc[i][2 * jj + 1];
      const f32x4 lo = *reinterpret_cast<const f32x4*>(rp0);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(rp1);
      // the next tile's second K-tile (issued before this point) has landed once at most WAITN DMA instructions are
      // left, not counting the NLD * AUXD requests issued above, which are younger: wait here, before the first
      // store enters the (in-order) vmcnt queue
      if (P == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN + NLD * AUXD) : "memory");
      if (EPI == 3) {
        if (lo[0] + hi[3] == 123.456f) g.C[0] = f2bf(lo[1]);
        return;
      }
      const int n = nw + jj * 32 + rq * 8;
      const bool ok = m < g.M && n < g.N;
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if constexpr (EPI == 5) {
        const u32x4 gq = pa[P % AUXD], uq = pb[P % AUXD];
        prefetch(std::integral_constant<int, P + AUXD>{});
        u32x4 og, ou;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float r0[2], r1[2];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const float gg = hh ? hi_bf(gq[e]) : lo_bf(gq[e]);
            const float uu = hh ? hi_bf(uq[e]) : lo_bf(uq[e]);
            const float d = bf2f(f2bf(v[2 * e + hh]));
            const float sg = sigmoid_fast(gg);
            r0[hh] = d * uu * sg * (1.0f + gg * (1.0f - sg));
            r1[hh] = d * (gg * sg);
          }
          og[e] = pack2bf(r0[0], r0[1]);
          ou[e] = pack2bf(r1[0], r1[1]);
        }
        const unsigned off = ((unsigned)m * (unsigned)g.ldc + (unsigned)(((n >> 5) << 6) + (n & 31))) * 2u;
        epi_store(rC, ok ? off : OOB, og);
        epi_store(rC, ok ? off + 64u : OOB, ou);
        return;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * alpha + ((e & 1) ? hi_bf(biasq[jj][e >> 1]) : lo_bf(biasq[jj][e >> 1]));
      const unsigned offc = ok ? ((unsigned)m * (unsigned)g.ldc + (unsigned)n) * 2u : OOB;
      if constexpr (EPI == 1 || EPI == 4) {      // 1: the quick-GELU pair, 4: the erf-GELU pair (adapter)
        const u32x4 h = pa[P % AUXD];
        prefetch(std::integral_constant<int, P + AUXD>{});
        u32x4 pre;
#pragma unroll
        for (int e = 0; e < 4; ++e) pre[e] = pack2bf(v[2 * e], v[2 * e + 1]);
        epi_store(rX, offc, pre);                // (forward with a pre-activation copy; dropped otherwise)
        if (fwd_act) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] = EPI == 1 ? act_quick_gelu(lo_bf(pre[e])) : act_gelu_erf(lo_bf(pre[e]));
            v[2 * e + 1] = EPI == 1 ? act_quick_gelu(hi_bf(pre[e])) : act_gelu_erf(hi_bf(pre[e]));
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = (e & 1) ? hi_bf(h[e >> 1]) : lo_bf(h[e >> 1]);
            v[e] *= EPI == 1 ? dact_quick_gelu(x) : dact_gelu_erf(x);
          }
        }
      }
      if constexpr (EPI == 6) {
        const u32x4 r = pa[P % AUXD];
        prefetch(std::integral_constant<int, P + AUXD>{});
        // x = residual + bf16(f(x)) : the branch output is rounded before the add
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e])) + ((e & 1) ? hi_bf(r[e >> 1]) : lo_bf(r[e >> 1]));
      }
      u32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = pack2bf(v[2 * e], v[2 * e + 1]);
      epi_store(rC, offc, out);                  // (null C: ZK_EPI_SWIGLU without the gate|up copy)
      if constexpr (TN == 4 && EPI == 2) {
        // ZK_EPI_SWIGLU: this wave's 64 columns are one interleave block [32 gate | 32 up]: jj = 0
        // delivered the lane's 8 gate features, jj = 1 the matching up features of the same row
        if (jj == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { gate[2 * e] = lo_bf(out[e]); gate[2 * e + 1] = hi_bf(out[e]); }
        } else {
          u32x4 o4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float r[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const float gg = gate[2 * e + hh], uu = hh ? hi_bf(out[e]) : lo_bf(out[e]);
              const float sl = silu_fast(gg);
              r[hh] = bf2f(f2bf(sl)) * uu;      // tf:llama:175: act_fn(gate) is a bf16 tensor
            }
            o4[e] = pack2bf(r[0], r[1]);
          }
          epi_store(rX, ok ? ((unsigned)m * (unsigned)g.ldaux + (unsigned)((nw >> 1) + rq * 8)) * 2u : OOB, o4);
        }
      }
    });
  });
}

#ifdef ZK_RING_STAMPS
// diagnostic build only (tools/ring_stamps.py): where a persistent workgroup's time goes.  100 MHz wall-clock stamps of
// lane 0 of every wave at tile start / K loop end / epilogue code end, for the first 8 tiles of the first 256 workgroups
static __device__ unsigned long long g_rstamps[256 * 8 * 8 * 4];
#define ZK_RSTAMP(ord, k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 256 && (ord) < 8) \
    g_rstamps[((blockIdx.x * 8 + (ord)) * 8 + (threadIdx.x >> 6)) * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ZK_RSTAMP(ord, k) do { } while (0)
#endif

template <int BM, int BN, int WAVES_M, int WAVES_N, int RS, int NSLOT, bool LEAN, int EPI>
__global__ __launch_bounds__(512) void gemm_pring_kernel(GemmArgs g) {
  constexpr int NT = 512;
  static_assert(WAVES_M * WAVES_N == 8, "8 waves");
  constexpr int TM = BM / WAVES_M / 16, TN = BN / WAVES_N / 16;
  constexpr int SA = BM / RS, SB = BN / RS, SPT = SA + SB;
  static_assert(SA * RS == BM && SB * RS == BN && NSLOT >= 2 * SPT && NSLOT - 2 * SPT < SPT, "slot geometry");
  // With exactly two K-tiles in the ring a K-tile never wraps: its slots are one linear [BM + BN][64] image and a
  // wave's rows may span slots (192x256: 7 slots of 64 rows per K-tile); otherwise they must lie in one slot.
  constexpr bool LINEAR = NSLOT == 2 * SPT;
  static_assert(LINEAR || (RS % (TM * 16) == 0 && RS % (TN * 16) == 0), "a wave's rows lie in one slot");
  static_assert(RS % 16 == 0, "a fragment's rows lie in one slot");
  constexpr int SLOT = RS * BK * 2;
  constexpr int PER = RS * 8 / NT;             // DMA instructions per thread per slot
  static_assert(PER * NT == RS * 8, "slot rows");
  constexpr int AHEAD = NSLOT - 2 * SPT;       // slots that stay in flight when the next K-tile is needed
  constexpr int WAITN = AHEAD * PER;
  constexpr int PATCH = 2048;
  static_assert(NSLOT * SLOT + 8 * PATCH <= 160 * 1024, "LDS budget");
  static_assert((NSLOT - SPT) * PER <= 63, "prologue wait");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NSLOT * SLOT + 8 * PATCH];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / WAVES_N, wn = wv % WAVES_N;
  const int ntiles = g.tiles_m * g.tiles_n;
  const int G = gridDim.x;
  const int n_my = (ntiles - (int)blockIdx.x + G - 1) / G;
  const int nk = (g.K + BK - 1) / BK;

  auto coords = [&](int ord, int& m0, int& n0) {   // (past the last tile: the last tile again, see below)
    int id = blockIdx.x + (ord < n_my ? ord : n_my - 1) * G;
    int q = ntiles >> 3, r = ntiles & 7, xcd = id & 7, idx = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int band = id / (g.band * g.tiles_n);
    const int band_m = min(g.band, g.tiles_m - band * g.band);
    const int within = id - band * g.band * g.tiles_n;
    m0 = (band * g.band + within % band_m) * BM;
    n0 = (within / band_m) * BN;
  };

  if (g.phase_delay > 0) {   // start stagger (host policy in zk_gemm_bf16): the XCD's workgroups in 4 phases
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int nph = g.phase_delay >> 20 ? g.phase_delay >> 20 : 4;
    const unsigned long long wait = (unsigned long long)(g.phase_delay & 0xFFFFF) * ((blockIdx.x >> 3) % nph);
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }
  // whole-matrix descriptors; the tile origin goes into the (scalar) offset (host: M*lda, N*ldb < 2^30)
  // LEAN addresses rows without a bounds test: the descriptors end at the last row, so rows past M / N (ragged
  // edge tiles) are out of range for the hardware and arrive as zeros (host: M * lda * 2 < 2^31, K % 64 == 0)
  __amdgpu_buffer_rsrc_t rA = make_rsrc(g.A, LEAN ? (unsigned)g.M * (unsigned)g.lda * 2u : 0x7FFFFFF0u);
  __amdgpu_buffer_rsrc_t rB = make_rsrc(g.B, LEAN ? (unsigned)g.N * (unsigned)g.ldb * 2u : 0x7FFFFFF0u);
  const unsigned voff_a = (unsigned)((tid >> 3) * g.lda + (((tid & 7) ^ ((tid >> 3) & 7)) * 8)) * 2u;
  const unsigned voff_b = (unsigned)((tid >> 3) * g.ldb + (((tid & 7) ^ ((tid >> 3) & 7)) * 8)) * 2u;

  // ---- issue side: a cursor over the stream of slot loads (tile, K-tile, part), all scalar and
  // branch-free.  Past the end of the stream the cursor keeps re-reading the last tile into slots
  // nobody reads again, so every step of every tile is the same straight-line code with the same
  // vmcnt arithmetic (cost: NSLOT slot loads per workgroup, drained before the workgroup exits).
  int i_tk = 0, i_pos = 0, i_m0, i_n0, nx_m0, nx_n0;
  coords(0, i_m0, i_n0);
  coords(1, nx_m0, nx_n0);
  auto issue_part = [&](auto part_c) {
    constexpr int part = decltype(part_c)::value;
    constexpr bool is_a = part < SA;
    constexpr int row0 = (is_a ? part : part - SA) * RS;
    unsigned char* dst = smem + i_pos * SLOT;
    const int ld = is_a ? g.lda : g.ldb;
    const int base_row = (is_a ? i_m0 : i_n0) + row0;
    if constexpr (LEAN) {
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        unsigned char* d = dst + (wv * 64 + NT * i) * 16;
        const int soff = ((base_row + 64 * i) * ld + i_tk * BK) * 2;
        if (is_a) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)d, 16, voff_a, soff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)d, 16, voff_b, soff, 0, 0);
      }
    } else {
      tile_dma<false, RS, NT>(is_a ? rA : rB, dst, ld, 0, is_a ? g.M : g.N, i_tk * BK, g.K, tid, base_row);
    }
    i_pos = (i_pos + 1 == NSLOT) ? 0 : i_pos + 1;
    if constexpr (part == SPT - 1) {
      const bool wrap = i_tk + 1 == nk;
      i_tk = wrap ? 0 : i_tk + 1;
      i_m0 = wrap ? nx_m0 : i_m0;
      i_n0 = wrap ? nx_n0 : i_n0;
    }
  };
  // ONE DMA instruction (piece i of a part): in the steady state the pieces of a K-tile are spread between the
  // MFMA rows of a burst instead of being issued back to back behind the barrier (a piece costs its wave 60-185
  // cycles of issue time in a burst, 25-60 in a gap between MFMAs: MI355X_MICROARCH.md, cycle constants;
  // measured here: +4-12 % on every large product, tools/bench_pring_ab.py)
  auto issue_piece = [&](auto part_c, auto i_c) {
    constexpr int part = decltype(part_c)::value, i = decltype(i_c)::value;
    constexpr bool is_a = part < SA;
    constexpr int row0 = (is_a ? part : part - SA) * RS;
    const int ld = is_a ? g.lda : g.ldb;
    if constexpr (LEAN) {
      unsigned char* d = smem + i_pos * SLOT + (wv * 64 + NT * i) * 16;
      const int soff = (((is_a ? i_m0 : i_n0) + row0 + 64 * i) * ld + i_tk * BK) * 2;
      if (is_a) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)d, 16, voff_a, soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)d, 16, voff_b, soff, 0, 0);
    } else {
      tile_dma<false, RS, NT, i>(is_a ? rA : rB, smem + i_pos * SLOT, ld, 0, is_a ? g.M : g.N, i_tk * BK, g.K, tid,
                                 (is_a ? i_m0 : i_n0) + row0);
    }
    if constexpr (i == PER - 1) {
      i_pos = (i_pos + 1 == NSLOT) ? 0 : i_pos + 1;
      if constexpr (part == SPT - 1) {
        const bool wrap = i_tk + 1 == nk;
        i_tk = wrap ? 0 : i_tk + 1;
        i_m0 = wrap ? nx_m0 : i_m0;
        i_n0 = wrap ? nx_n0 : i_n0;
      }
    }
  };
  constexpr int NP = SPT * PER;                 // pieces per K-tile and thread
  // pieces [lo, hi) of the steady-state group (parts in ring order, starting at part NSLOT % SPT)
  auto issue_pieces = [&](auto lo_c, auto hi_c) {
    constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
    static_for<HI - LO>([&](auto k_c) {
      constexpr int p = LO + decltype(k_c)::value;
      issue_piece(std::integral_constant<int, (NSLOT % SPT + p / PER) % SPT>{}, std::integral_constant<int, p % PER>{});
    });
  };
  auto issue_group = [&](auto first_c, auto count_c) {
    constexpr int FIRST = decltype(first_c)::value, COUNT = decltype(count_c)::value;
    static_for<COUNT>([&](auto k_c) { issue_part(std::integral_constant<int, (FIRST + decltype(k_c)::value) % SPT>{}); });
  };

  // prologue: fill the ring (host guarantees nk >= 4: no tile boundary in here), wait for K-tile 0
  issue_group(std::integral_constant<int, 0>{}, std::integral_constant<int, NSLOT>{});
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - SPT) * PER) : "memory");
  __builtin_amdgcn_s_barrier();

  // ---- consume side
  const int a_part = LINEAR ? 0 : wm * (TM * 16) / RS, a_row = wm * (TM * 16) - a_part * RS;
  const int b_part = LINEAR ? SA : SA + wn * (TN * 16) / RS, b_row = LINEAR ? wn * (TN * 16) : wn * (TN * 16) % RS;
  int c_pos = 0;                                // ring position of part 0 of the K-tile being multiplied
  auto slot_of = [&](int pos0, int part) {
    int p = pos0 + part;
    p = p >= NSLOT ? p - NSLOT : p;
    return smem + p * SLOT;
  };
  f32x4 acc[TM][TN];
  bf16x8 fa[TM], fb0[TN], fb1[TN];
  // One k-half: MFMAs on (fa, fcur) while the fragments of the NEXT k-half arrive: B's into the
  // other buffer up front, each A fragment in place as soon as its last MFMA has been issued.
  auto burst = [&](bf16x8 (&fcur)[TN], bf16x8 (&fnext)[TN], const unsigned char* na, const unsigned char* nb, int nkk,
                   auto spread_c) {
    constexpr bool SPREAD = decltype(spread_c)::value;   // issue this K-tile's DMA pieces between the MFMA rows
#pragma unroll
    for (int j = 0; j < TN; ++j) fnext[j] = frag_load<false, RS>(nb, b_row + j * 16, nkk, lane);
    __builtin_amdgcn_s_setprio(1);
    static_for<TM>([&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
#pragma unroll
      for (int j = 0; j < TN / 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fcur[j], fa[i], acc[i][j], 0, 0, 0);
      if constexpr (SPREAD) {      // mid-row: away from the A refill read at the end of the row
        __builtin_amdgcn_sched_barrier(0);
        issue_pieces(std::integral_constant<int, i * NP / TM>{}, std::integral_constant<int, (i + 1) * NP / TM>{});
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = TN / 2; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fcur[j], fa[i], acc[i][j], 0, 0, 0);
      // (scheduling fence: keep the refill BEHIND the MFMAs that read the old value, or the
      //  compiler hoists it into fresh registers and the kernel no longer fits 256 VGPRs)
      __builtin_amdgcn_sched_barrier(0);
      fa[i] = frag_load<false, RS>(na, a_row + i * 16, nkk, lane);
    });
    __builtin_amdgcn_s_setprio(0);
  };
  {
    const unsigned char* la = slot_of(0, a_part);
    const unsigned char* lb = slot_of(0, b_part);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb0[j] = frag_load<false, RS>(lb, b_row + j * 16, 0, lane);
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = frag_load<false, RS>(la, a_row + i * 16, 0, lane);
  }

  for (int ord = 0; ord < n_my; ++ord) {
    int m0, n0;
    coords(ord, m0, n0);
    coords(ord + 1, nx_m0, nx_n0);              // the issue cursor crosses into the next tile at t = nk - 3
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // In a continued tile K-tile 1 was waited for before the previous epilogue's stores: its first
    // step must not wait on vmcnt (the stores are still draining behind the MFMAs).
    bool skip_vm = ord > 0;
    ZK_RSTAMP(ord, 0);
    for (int t = 0; t < nk; ++t) {
      int npos = c_pos + SPT;
      npos = npos >= NSLOT ? npos - NSLOT : npos;
      burst(fb0, fb1, slot_of(c_pos, a_part), slot_of(c_pos, b_part), 1, std::false_type{});
      // the next K-tile of the stream must have landed, and every wave must hold its last
      // fragments of this one before its slots are refilled
      if (skip_vm) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(WAITN) : "memory");
      skip_vm = false;
      __builtin_amdgcn_s_barrier();
      // (for t = nk - 1 the fragments fetched below are K-tile 0 of the next tile)
      burst(fb1, fb0, slot_of(npos, a_part), slot_of(npos, b_part), 0, std::true_type{});
      c_pos = npos;
    }
    ZK_RSTAMP(ord, 1);
    store_tile_p<TM, TN, WAITN, EPI>(g, acc, m0 + wm * (TM * 16), n0 + wn * (TN * 16), lane,
                                     smem + NSLOT * SLOT + wv * PATCH);
    ZK_RSTAMP(ord, 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
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

#ifdef ZK_RING_STAMPS
extern "C" int zk_debug_read_ring_stamps(void* host_dst) {
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_rstamps), sizeof(g_rstamps)) == hipSuccess ? 0 : -5;
}
extern "C" int zk_debug_clear_ring_stamps() {
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_rstamps)) != hipSuccess) return -5;
  return hipMemset(p, 0, sizeof(g_rstamps)) == hipSuccess ? 0 : -5;
}
#endif

#ifndef ZK_KERNEL_PROBE   // (probe builds instantiate one kernel for disassembly: tools/probe_kernel.sh)
namespace {
// Per-call plan (zk_gemm_plan_t of the header, or its defaults): the library keeps NO mutable tuning state.
struct Plan {
  int tile = -1;         // -1 = cost model, 0 = 128x128, 1 = 256x128, 2 = 256x256, 3 = 128x96, 4 = 384x192
  int splits = 0;        // 0 = heuristic, 1 = never split K, n > 1 = force n splits
  int stage_mode = 0;    // 0 = LDS-DMA for every operand (default since round 3: with the DMA as inline asm the compiler no
                         // longer drains it before the transposed reads, tools/bench_stage.py), 1 = registers for transposed operands
  int ring = 1;          // 1 = 3-stage ring kernel for NT products on the 256x128 tile
  int ring_mode = 0;     // ablations (ZK_ABLATION builds only): 1 = ring without MFMAs, 2 = without DMA, 3 = no epilogue
  int lean = 1;          // 0 = never use the lean DMA path
  int pring = 1;         // 1 = persistent slot ring for the large NT products (0 = one workgroup per tile)
  int band = 0;          // m-tiles per band of the persistent kernel's tile order (0 = default)
  int phase_delay = -1;  // persistent kernel start stagger, 10 ns ticks per phase step (-1 = built-in policy)
};
int g_num_cu[16] = {0};
// zk_gemm_plan_query: the whole decision path of zk_gemm_bf16_plan runs, nothing is launched (host logic, testable
// without a GPU)
thread_local bool g_dry_run = false;
#define ZK_GO(...) do { if (!g_dry_run) hipLaunchKernelGGL(__VA_ARGS__); } while (0)
#define ZK_CHECK_LAUNCH_G() do { if (!g_dry_run) { ZK_CHECK_LAUNCH(); } } while (0)
thread_local bool g_in_row_split = false;   // the two halves of a row-split product are not split again
thread_local int g_last_plan = -1;   // kernel variant of this thread's last zk_gemm_bf16 call (zk_gemm_last_plan)

int num_cus() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (!g_num_cu[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_num_cu[dev] = n;
  }
  return g_num_cu[dev];
}

template <int BM, int BN, int WM_, int WN_>
void launch_cfg(const GemmArgs& g0, int transA, int transB, hipStream_t stream, const Plan& P) {
  GemmArgs g = g0;
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  dim3 grid(g.tiles_m * g.tiles_n * g.splits), block(64 * WM_ * WN_);
  const bool reg = P.stage_mode == 1;
#define ZK_LAUNCH(TA_, TB_, STG_) ZK_GO((gemm_kernel<TA_, TB_, BM, BN, WM_, WN_, STG_>), grid, block, 0, stream, g)
  if (!transA && !transB) ZK_LAUNCH(false, false, 0);
  else if (transA && !transB) { if (reg) ZK_LAUNCH(true, false, 1); else ZK_LAUNCH(true, false, 0); }
  else if constexpr ((BN & (BN - 1)) == 0) {   // the transposed-B image needs a power-of-two row (swizzle)
    if (!transA) { if (reg) ZK_LAUNCH(false, true, 1); else ZK_LAUNCH(false, true, 0); }
    else { if (reg) ZK_LAUNCH(true, true, 1); else ZK_LAUNCH(true, true, 0); }
  }
#undef ZK_LAUNCH
}

// Split count of a plain product on the two-stage 128x128 kernel (two 4-wave workgroups per CU = 512 slots), by
// estimated time in microseconds.  Constants fitted to tools/bench_splitk.py on MI355X (profiles/
// r02_gemm_splitk_sweep.txt): a K-tile takes a workgroup 1.2 us with a co-resident partner and 0.87 us alone on its
// CU (k-contiguous operands: x0.56), a workgroup costs 5 us to start and store, the fp32 slabs move 8 bytes per
// element and split at ~8 TB/s (they stay in the Infinity Cache) and the finishing kernel costs 4 us.
int splitk_choice(int64_t tiles, int64_t M, int64_t N, int64_t K, bool transposed, int64_t workspace_bytes) {
  const double tau2 = transposed ? 1.2 : 0.67, tau1 = transposed ? 0.87 : 0.49, fixed = 5.0;
  int best = 1;
  double best_cost = 1e300;
  for (int s_ = 1; s_ <= 16; ++s_) {
    if (s_ > 1 && (K / s_ < 512 || (int64_t)s_ * M * N * 4 > workspace_bytes)) break;
    const double kt = (double)((K + s_ - 1) / s_ + BK - 1) / BK;
    const int64_t blocks = tiles * s_, full = blocks / 512, rem = blocks % 512;
    double cost = (double)full * (kt * tau2 + fixed);
    if (rem > 0) cost += kt * ((rem > 256 || full > 0) ? tau2 : tau1) + fixed;
    if (s_ > 1) cost += (double)s_ * M * N * 8.0 / 8.0e6 + 4.0;
    if (cost < best_cost) { best_cost = cost; best = s_; }
  }
  return best;
}

// persistent ring: byte offsets from the matrix base are 32-bit, and the epilogue must be one of its classes
// Epilogue class of the persistent ring for a call (store_tile_p), or -1 if none of them covers it
int pring_class(int epilogue, bool has_residual, bool accumulate) {
  if (accumulate) return -1;                 // (C += ... only occurs on transposed products, which the ring does not take)
  if (epilogue == ZK_EPI_NONE) return has_residual ? 6 : 0;
  if (has_residual) return -1;
  if (epilogue == ZK_EPI_QUICK_GELU || epilogue == ZK_EPI_MUL_DQUICK_GELU) return 1;
  if (epilogue == ZK_EPI_SWIGLU) return 2;
  if (epilogue == ZK_EPI_SWIGLU_BWD) return 5;
  if (epilogue == ZK_EPI_GELU_ERF || epilogue == ZK_EPI_MUL_DGELU_ERF) return 4;
  return -1;
}
// The persistent ring addresses whole operands with 32-bit byte offsets: A / B through the DMA descriptors (64 rows of
// slack for ragged tiles), every epilogue operand through a descriptor that ends at its last row
bool pring_fits(int64_t M, int64_t N, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int64_t ldaux, int epilogue,
                bool has_residual, bool accumulate) {
  const int64_t lim = 0x7FFFFFF0LL;
  if (pring_class(epilogue, has_residual, accumulate) < 0) return false;
  if ((M + 64) * lda * 2 >= lim || (N + 64) * ldb * 2 >= lim || M * ldc * 2 >= lim) return false;
  if (has_residual && M * ldr * 2 >= lim) return false;
  const bool uses_aux = epilogue == ZK_EPI_SWIGLU || epilogue == ZK_EPI_SWIGLU_BWD ||
                        epilogue == ZK_EPI_MUL_DQUICK_GELU || epilogue == ZK_EPI_MUL_DGELU_ERF;
  return !(uses_aux && M * ldaux * 2 >= lim);
}

// Tile choice: time ~ rounds(tiles / concurrent blocks) x tile area / relative efficiency.
int choose_tile(int64_t M, int64_t N, int64_t K, int transA, int transB, bool need_tn4, const Plan& P,
                double* cost_out = nullptr) {
  const bool nt_ring = !transA && !transB && P.ring;
  if (P.tile >= 0) {
    if (P.tile == 3 && (transB || need_tn4)) return 0;
    if (P.tile == 4 && (!nt_ring || need_tn4)) return 2;
    if (P.tile == 5 && !(nt_ring && P.pring && (P.lean || (M % 192 == 0 && N % 256 == 0)) && K % BK == 0 && K >= 4 * BK)) return 2;
    return P.tile;
  }
  const int bm[6] = {128, 256, 256, 128, 384, 192}, bn[6] = {128, 128, 256, 96, 192, 256};
  const double slots[6] = {512.0, 256.0, 256.0, 512.0, 256.0, 256.0};   // two 4-wave blocks or one 8-wave block per CU
  // measured on MI355X (tools/bench_gemm.py); the slot-ring kernels exist for NT products only.  The persistent
  // ring's advantage grows with K (its ring fill and store burst amortise): 1.3 at K <= 2048 (M = 36928, N = 1024,
  // K = 1024: 580 tiles of 256x256 in three rounds 78 us, 2312 tiles of 128x128 in five 90 us; encoder fc1 at
  // M = 4616: 51 us against 52 us), 1.35 from K = 4096 (the N = 4096 products of the 8B surrogate: 384 tiles of
  // 256x256 in two rounds beat 1536 tiles of 128x128 in three, 178 vs 190 us at K = 4096, 630 vs 727 us at K = 14336)
  const double kf = K <= 2048 ? 0.0 : (K >= 4096 ? 1.0 : (double)(K - 2048) / 2048.0);
  const double pr = 1.3 + 0.05 * kf;
  // 192x256 (persistent ring): 9.1 B/kFLOP of operand traffic against 7.8 — it wins where it removes a partial
  // round (M = 6144, N = 4096: 512 tiles = two per CU, against 384 tiles of 256x256 = 1.5; the encoder's fc1 at
  // M = 4616: 400 tiles, 16 of them 8 rows tall, 42 us against 52 us on 128x128).  Ragged M / N ride on the lean
  // descriptors.
  const bool t5 = nt_ring && P.pring && (P.lean || (M % 192 == 0 && N % 256 == 0)) && K % BK == 0 && K >= 4 * BK;
  const double eff[6] = {1.0, 0.9, nt_ring ? pr : 1.02, 0.93, pr, pr * 0.94};
  int best = 0;
  double best_cost = 1e300;
  for (int c = 0; c < 6; ++c) {
    if ((c == 3 && transB) || (c == 4 && !nt_ring) || (c == 5 && !t5)) continue;
    if (need_tn4 && (c == 3 || c == 4)) continue;   // the SwiGLU epilogue needs 64 columns per wave
    double tiles = (double)((M + bm[c] - 1) / bm[c]) * (double)((N + bn[c] - 1) / bn[c]);
    double rounds = (double)(int64_t)((tiles + slots[c] - 1) / slots[c]);
    double cost = rounds * slots[c] * bm[c] * bn[c] / eff[c];
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  if (cost_out) *cost_out = best_cost;
  return best;
}

// Row split.  A product whose tiling leaves a nearly empty last round (C2's fc2: 580 tiles of 256x256 on 256 CUs =
// 2.27 rounds, paid as 3) is cut into a part of whole rounds and a remainder that is dispatched as a product of
// its own (for which the cost model picks its own tile, usually the small two-stage tiles that fill the chip with
// few rows).  Both run on the caller's stream, one after the other.  Estimated times in microseconds from the cost
// model's unit (tile area x rounds x slots / efficiency) at K: cost * 2K flops over ~1075 TFLOP/s at efficiency 1.0,
// plus a launch / ramp / tail allowance per launch; the split is taken when it saves 7 % or more.
// Returns the rows of the first part (a multiple of its tile height), 0 for "do not split".
int64_t row_split(int64_t M, int64_t N, int64_t K, int transB, bool need_tn4, const Plan& P) {
  if (M < 2048 || !P.ring || !P.pring || K % BK != 0 || K < 4 * BK) return 0;
  const double us_per_cost = 2.0 * (double)K / 1.075e9, launch_us = 7.0;
  double c0;
  choose_tile(M, N, K, 0, transB, need_tn4, P, &c0);
  const double t0 = c0 * us_per_cost + launch_us;
  const int bm[3] = {256, 384, 192}, bn[3] = {256, 192, 256};
  const double kf = K <= 2048 ? 0.0 : (K >= 4096 ? 1.0 : (double)(K - 2048) / 2048.0);
  const double eff[3] = {1.3 + 0.05 * kf, 1.3 + 0.05 * kf, (1.3 + 0.05 * kf) * 0.94};
  int64_t best_m1 = 0;
  double best_t = t0 * 0.93;
  for (int c = 0; c < 3; ++c) {
    if (transB) break;                                   // the persistent ring runs NT products only
    if (need_tn4 && c == 1) continue;
    if (c == 2 && !P.lean && (N % 256 != 0)) continue;
    const int64_t tn = (N + bn[c] - 1) / bn[c];
    const int64_t rounds = ((M / bm[c]) * tn) / 256;     // whole rounds available
    for (int64_t r = rounds; r >= 1 && r + 1 >= rounds; --r) {
      const int64_t m1 = (r * 256 / tn) * bm[c];
      if (m1 <= 0 || m1 >= M || M - m1 < 64) continue;
      const double used = (double)((m1 / bm[c]) * tn) / (double)(r * 256);
      if (used < 0.93) continue;                         // the "whole" rounds must be nearly full
      const double t1 = (double)r * 256.0 * bm[c] * bn[c] / eff[c] * us_per_cost + launch_us;
      double c2;
      choose_tile(M - m1, N, K, 0, transB, need_tn4, P, &c2);
      const double t = t1 + c2 * us_per_cost + launch_us;
      if (t < best_t) { best_t = t; best_m1 = m1; }
    }
  }
  return best_m1;
}
}  // namespace

extern "C" int zk_gemm_last_plan(void) { return g_last_plan; }

extern "C" int zk_gemm_plan_query(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int transA,
                                  int transB, int epilogue, int has_bias, int has_residual, int64_t ldr, int has_aux_out,
                                  int has_aux_in, int64_t ldaux, int accumulate, int64_t workspace_bytes,
                                  const zk_gemm_plan_t* plan) {
  // 16-byte aligned stand-ins that are never dereferenced: with g_dry_run set nothing is launched
  void* const p = reinterpret_cast<void*>(static_cast<uintptr_t>(0x1000));
  g_dry_run = true;
  g_last_plan = -1;
  const int rc = zk_gemm_bf16_plan(p, p, p, M, N, K, lda, ldb, ldc, transA, transB, epilogue, has_bias ? p : nullptr,
                                   has_residual ? p : nullptr, ldr, has_aux_out ? p : nullptr, has_aux_in ? p : nullptr,
                                   ldaux, 1.0f, nullptr, accumulate, workspace_bytes > 0 ? p : nullptr, workspace_bytes,
                                   plan, nullptr);
  g_dry_run = false;
  return rc == ZK_OK ? g_last_plan : rc;
}

extern "C" int zk_gemm_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                            int64_t ldb, int64_t ldc, int transA, int transB, int epilogue, const void* bias,
                            const void* residual, int64_t ldr, void* aux_out, const void* aux_in, int64_t ldaux,
                            float alpha, const float* alpha_dev, int accumulate, void* workspace,
                            int64_t workspace_bytes, hipStream_t stream) {
  return zk_gemm_bf16_plan(A, B, C, M, N, K, lda, ldb, ldc, transA, transB, epilogue, bias, residual, ldr, aux_out,
                           aux_in, ldaux, alpha, alpha_dev, accumulate, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int zk_gemm_bf16_plan(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                                 int64_t ldb, int64_t ldc, int transA, int transB, int epilogue, const void* bias,
                                 const void* residual, int64_t ldr, void* aux_out, const void* aux_in, int64_t ldaux,
                                 float alpha, const float* alpha_dev, int accumulate, void* workspace,
                                 int64_t workspace_bytes, const zk_gemm_plan_t* plan, hipStream_t stream) {
  Plan P;
  if (plan) {
    if (plan->tile < -1 || plan->tile > 5 || plan->splits < 0 || plan->splits > 16 || plan->band < 0 || plan->band > 16)
      return ZK_EINVAL;
    P.tile = plan->tile; P.splits = plan->splits;
    if (plan->stage_mode >= 0) P.stage_mode = plan->stage_mode != 0;
    if (plan->ring >= 0) P.ring = plan->ring != 0;
    if (plan->lean >= 0) P.lean = plan->lean != 0;
    if (plan->pring >= 0) P.pring = plan->pring != 0;
    P.band = plan->band; P.phase_delay = plan->phase_delay;
#ifdef ZK_ABLATION
    P.ring_mode = plan->ablation;
#else
    if (plan->ablation) return ZK_EINVAL;       // wrong-result timing modes exist in ZK_ABLATION builds only
#endif
  }
  // C may be null for ZK_EPI_SWIGLU only: h = silu(gate) * up is still written to aux_out, the interleaved gate|up
  // pre-activations (needed by the backward alone) are not — inference / no-grad forward saves their 2F bf16 per row
  if (!A || !B || (!C && epilogue != ZK_EPI_SWIGLU) || M <= 0 || N <= 0 || K <= 0) return ZK_EINVAL;
  if ((N & 7) || (ldc & 7)) return ZK_EINVAL;   // the epilogue moves 8 consecutive n (16 bytes) per lane
  if ((!transA || !transB) && (K & 7)) return ZK_EINVAL;  // k-contiguous operands are read in 8-element chunks
  if (transA ? ((M & 7) || (lda & 7)) : (lda & 7)) return ZK_EINVAL;
  if (ldb & 7) return ZK_EINVAL;
  if (epilogue < 0 || epilogue > ZK_EPI_SWIGLU_BWD) return ZK_EINVAL;
  if (epilogue == ZK_EPI_SWIGLU || epilogue == ZK_EPI_SWIGLU_BWD) {
    if (bias || residual || accumulate || alpha_dev || alpha != 1.0f || (ldaux & 7)) return ZK_EINVAL;
    if (epilogue == ZK_EPI_SWIGLU && (!aux_out || (N & 63) || ldaux < N / 2)) return ZK_EINVAL;
    if (epilogue == ZK_EPI_SWIGLU_BWD && (!aux_in || (N & 31) || ldaux < 2 * N || ldc < 2 * N)) return ZK_EINVAL;
  }
  if ((epilogue == ZK_EPI_MUL_DQUICK_GELU || epilogue == ZK_EPI_MUL_DGELU_ERF) && !aux_in) return ZK_EINVAL;
  if (residual && ((ldr & 7) || (reinterpret_cast<uintptr_t>(residual) & 15))) return ZK_EINVAL;
  if (bias && (reinterpret_cast<uintptr_t>(bias) & 15)) return ZK_EINVAL;
  if (aux_in && ((ldaux & 7) || (reinterpret_cast<uintptr_t>(aux_in) & 15))) return ZK_EINVAL;
  if (aux_out && (reinterpret_cast<uintptr_t>(aux_out) & 15)) return ZK_EINVAL;
  if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) return ZK_EINVAL;
  if (reinterpret_cast<uintptr_t>(C) & 15) return ZK_EINVAL;
  // 32-bit byte offsets inside one block tile
  if ((transA ? K : 384) * lda * 2 >= 0x7FFFFFF0LL || (transB ? K : 256) * ldb * 2 >= 0x7FFFFFF0LL) return ZK_EINVAL;
  GemmArgs g;
  g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.C = (bf16_t*)C;
  g.bias = (const bf16_t*)bias; g.residual = (const bf16_t*)residual;
  g.aux_out = (bf16_t*)aux_out; g.aux_in = (const bf16_t*)aux_in;
  g.alpha_dev = alpha_dev; g.alpha = alpha;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.lda = (int)lda; g.ldb = (int)ldb; g.ldc = (int)ldc; g.ldr = (int)ldr; g.ldaux = (int)ldaux;
  g.epilogue = epilogue; g.accumulate = accumulate;
  g.tiles_m = g.tiles_n = 0;
  g.band = 8;
  g.phase_delay = 0;
  g.splits = 1; g.kchunk = (int)((K + BK - 1) / BK * BK); g.slab = nullptr;
  // split-K for small outputs with a long contraction (wgrad, lm_head dgrad): fill the 256 CUs
  // ---- row split (built-in plan only): whole rounds on the persistent ring + the remainder as its own product
#ifndef ZK_NO_ROW_SPLIT
  if (!plan && !transA && !g_in_row_split) {
    const bool plain_sk = epilogue == ZK_EPI_NONE && !bias && !residual && !aux_out && workspace &&
                          ((M + 127) / 128) * ((N + 127) / 128) <= 1024 && K >= 1024;     // split-K territory below
    const int64_t m1 = plain_sk ? 0 : row_split(M, N, K, transB, epilogue == ZK_EPI_SWIGLU, P);
    if (m1 > 0) {
      const int64_t ld_aux_out = epilogue == ZK_EPI_SWIGLU ? ldaux : ldc;    // (the GELU pre-activation copy has C's stride)
      auto at = [](const void* p, int64_t elems) -> const void* {
        return p ? static_cast<const void*>(static_cast<const bf16_t*>(p) + elems) : nullptr;
      };
      g_in_row_split = true;
      int rc = zk_gemm_bf16_plan(A, B, C, m1, N, K, lda, ldb, ldc, transA, transB, epilogue, bias, residual, ldr, aux_out,
                                 aux_in, ldaux, alpha, alpha_dev, accumulate, workspace, workspace_bytes, nullptr, stream);
      const int main_plan = g_last_plan;
      if (rc == ZK_OK)
        rc = zk_gemm_bf16_plan(at(A, m1 * lda), B, const_cast<void*>(at(C, m1 * ldc)), M - m1, N, K, lda, ldb, ldc, transA,
                               transB, epilogue, bias, at(residual, m1 * ldr), ldr,
                               const_cast<void*>(at(aux_out, m1 * ld_aux_out)), at(aux_in, m1 * ldaux), ldaux, alpha,
                               alpha_dev, accumulate, workspace, workspace_bytes, nullptr, stream);
      g_in_row_split = false;
      g_last_plan = main_plan;           // the measurement hook reports the kernel that did most of the work
      return rc;
    }
  }
#endif
  const bool plain = epilogue == ZK_EPI_NONE && !bias && !residual && !aux_out;
  if (plain && workspace && P.splits != 1) {
    const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
    int64_t sp = P.splits > 1 ? P.splits : 1;
    if (P.splits <= 1 && tiles <= 1024 && K >= 1024) sp = splitk_choice(tiles, M, N, K, transA || transB, workspace_bytes);
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
    g_last_plan = 4000 + g.splits;
    launch_cfg<128, 128, 2, 2>(g, transA, transB, stream, P);
    int64_t nb = (M * (N / 4) + 255) / 256;
    if (nb > 2048) nb = 2048;
    ZK_GO(splitk_finish_kernel, dim3((unsigned)nb), dim3(256), 0, stream, g.slab, g.C, g.M, g.N, g.ldc,
                       g.splits, alpha, alpha_dev, accumulate);
    ZK_CHECK_LAUNCH_G();
    return ZK_OK;
  }
  int tile = choose_tile(M, N, K, transA, transB, epilogue == ZK_EPI_SWIGLU, P);
  // the 192x256 tile exists on the persistent ring only, which needs whole-matrix 32-bit byte offsets (operands
  // under 2 GiB) and one of its epilogue classes: anything else runs as the 256x256 tile, whose one-workgroup-per-
  // tile ring / two-stage kernels take every shape (a plain zk_gemm_bf16 call must not fail on a tile choice)
  const bool ring_ok = pring_fits(M, N, lda, ldb, ldc, ldr, ldaux, epilogue, residual != nullptr, accumulate != 0);
  if (tile == 5 && !ring_ok) tile = 2;
  g_last_plan = tile * 100;            // plain two-stage kernel unless a ring variant takes over below
  switch (tile) {
    case 3: launch_cfg<128, 96, 2, 2>(g, transA, transB, stream, P); break;
    case 5:
    case 4:
    case 2: {
      if (!transA && !transB && P.ring) {
        GemmArgs gr = g;
        const int bm = tile == 4 ? 384 : (tile == 5 ? 192 : 256), bn = tile == 4 ? 192 : 256;
        gr.tiles_m = (g.M + bm - 1) / bm; gr.tiles_n = (g.N + bn - 1) / bn;
        const dim3 rg(gr.tiles_m * gr.tiles_n);
        const bool lean = g.M % bm == 0 && g.N % bn == 0 && g.K % BK == 0 && P.lean;   // one-workgroup-per-tile ring
        const bool lean_p = g.K % BK == 0 && P.lean;   // persistent ring: ragged M / N fall out of the descriptors
        // persistent ring: needs >= 4 K-tiles per tile and whole-matrix 32-bit byte offsets
        const int pclass = P.ring_mode == 3 ? 3 : pring_class(epilogue, residual != nullptr, accumulate != 0);
        if (P.pring && pclass >= 0 && ring_ok && (g.K + BK - 1) / BK >= 4 && (P.ring_mode == 0 || P.ring_mode == 3)) {
          const int ncu = num_cus();
          const int ntl = gr.tiles_m * gr.tiles_n;
          const dim3 pg(ntl < ncu ? ntl : ncu);
          g_last_plan = 3000 + tile * 100 + (lean_p ? 10 : 0) + pclass;
          // an XCD holds 32 consecutive tiles of the order = band x (32 / band): its L2 then streams
          // band * BM rows of A and (32 / band) * BN rows of B per K-tile; 384x192 tiles want 4 x 8
          gr.band = P.band > 0 ? P.band : (tile == 4 ? 4 : 8);
          // Start stagger.  Identical workgroups reach their epilogues together, and 256 CUs storing
          // 128-256 KiB each within the same few microseconds run into the fabric's write rate (the
          // stores block at issue, the matrix pipes idle).  Starting the 32 workgroups of an XCD in 4
          // phases 3 us apart spreads every later tile boundary; measured (tools/bench_phase.py): fused
          // gate|up 589 -> 501 us, fused down-dgrad 316 -> 294, lm_head 710 -> 684, C2 fc1 437 -> 412;
          // with fewer than 3 tiles per workgroup the 9 us tail is not earned back.
          // Round 3 (branch-free epilogues, tools/bench_phase.py again): the light epilogues (classes 0 / 6) keep 3 us;
          // the heavy ones (activation pairs: twice the stores plus the vector work) want about 1 us per tile round,
          // between 3 and 6 (fused gate|up, 6 rounds: 468 us at 3, 454 at 6; SwiGLU backward, 3 rounds: 269 at 3)
          const int rounds = ntl / ncu;
          const int heavy_us = rounds < 3 ? 3 : (rounds > 6 ? 6 : rounds);
          gr.phase_delay = P.phase_delay >= 0 ? P.phase_delay
                           : ntl < 3 * ncu     ? 0
                           : (pclass == 0 || pclass == 6 || pclass == 3) ? 300 : heavy_us * 100;
#define ZK_PRING(EPI_) \
  do { \
    if (tile == 5) ZK_GO((gemm_pring_kernel<192, 256, 2, 4, 64, 14, true, EPI_>), pg, dim3(512), 0, stream, gr); \
    else if (tile == 4 && lean_p) ZK_GO((gemm_pring_kernel<384, 192, 4, 2, 192, 6, true, EPI_>), pg, dim3(512), 0, stream, gr); \
    else if (tile == 4) ZK_GO((gemm_pring_kernel<384, 192, 4, 2, 192, 6, false, EPI_>), pg, dim3(512), 0, stream, gr); \
    else if (lean_p) ZK_GO((gemm_pring_kernel<256, 256, 2, 4, 128, 9, true, EPI_>), pg, dim3(512), 0, stream, gr); \
    else ZK_GO((gemm_pring_kernel<256, 256, 2, 4, 128, 9, false, EPI_>), pg, dim3(512), 0, stream, gr); \
  } while (0)
          if (pclass == 0) ZK_PRING(0);
          else if (pclass == 1) ZK_PRING(1);
          else if (pclass == 2) ZK_PRING(2);
          else if (pclass == 4) ZK_PRING(4);
          else if (pclass == 5) ZK_PRING(5);
          else if (pclass == 6) ZK_PRING(6);
          else ZK_PRING(3);
#undef ZK_PRING
          break;
        }
        if (tile == 5) return ZK_EINVAL;   // unreachable: tile 5 was turned into tile 2 above unless the ring's conditions hold (forced ablation modes aside)
        g_last_plan = 2000 + tile * 100 + (lean ? 10 : 0);
#define ZK_RINGX(MODE_) \
  do { \
    if (tile == 4 && lean) ZK_GO((gemm_ringx_kernel<384, 192, 2, 4, 192, 6, MODE_, true>), rg, dim3(512), 0, stream, gr); \
    else if (tile == 4) ZK_GO((gemm_ringx_kernel<384, 192, 2, 4, 192, 6, MODE_, false>), rg, dim3(512), 0, stream, gr); \
    else if (lean) ZK_GO((gemm_ringx_kernel<256, 256, 2, 4, 256, 5, MODE_, true>), rg, dim3(512), 0, stream, gr); \
    else ZK_GO((gemm_ringx_kernel<256, 256, 2, 4, 256, 5, MODE_, false>), rg, dim3(512), 0, stream, gr); \
  } while (0)
        if (P.ring_mode == 1) ZK_RINGX(1);
        else if (P.ring_mode == 2) ZK_RINGX(2);
        else if (P.ring_mode == 3) ZK_RINGX(3);
        else ZK_RINGX(0);
#undef ZK_RINGX
      } else {
        launch_cfg<256, 256, 2, 4>(g, transA, transB, stream, P);
      }
      break;
    }
    case 1:
      if (!transA && !transB && P.ring) {
        GemmArgs gr = g;
        gr.tiles_m = (g.M + 255) / 256; gr.tiles_n = (g.N + 127) / 128;
        const dim3 rg(gr.tiles_m * gr.tiles_n);
        g_last_plan = 1000 + tile * 100;
        if (P.ring_mode == 1) ZK_GO((gemm_ring_kernel<256, 128, 4, 2, 3, 1>), rg, dim3(512), 0, stream, gr);
        else if (P.ring_mode == 2) ZK_GO((gemm_ring_kernel<256, 128, 4, 2, 3, 2>), rg, dim3(512), 0, stream, gr);
        else ZK_GO((gemm_ring_kernel<256, 128, 4, 2, 3>), rg, dim3(512), 0, stream, gr);
      } else {
        launch_cfg<256, 128, 4, 2>(g, transA, transB, stream, P);
      }
      break;
    default: launch_cfg<128, 128, 2, 2>(g, transA, transB, stream, P); break;
  }
  ZK_CHECK_LAUNCH_G();
  return ZK_OK;
}
#endif  // ZK_KERNEL_PROBE
