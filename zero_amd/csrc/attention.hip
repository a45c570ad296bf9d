// Flash-style fused attention forward / backward for gfx950 (CDNA4), bf16 in, fp32 softmax.
//
// Replaces F.scaled_dot_product_attention (tf:integrations/sdpa_attention.py:154-163) for
//   * the CLIP encoder: bidirectional, 16 heads x head_dim 64, T = 577 (tf:clip:317-330)
//   * the Llama surrogate: causal, GQA (8 kv heads), head_dim 128 (tf:llama:264-277)
// and its autograd backward.
//
// Layout idea (all three kernels): every product is a v_mfma_f32_32x32x16_bf16 whose
// accumulator already IS the next product's operand, so the probabilities never travel
// through LDS:
//   fwd / dq : S^T[key][query] = K Q^T  -> query on the lane: row max / row sum / LSE / delta
//              are per-lane scalars; P^T (resp. dS^T) is packed to bf16 in registers and
//              used as the B operand of  O^T += V^T P^T  (resp. dQ^T += K^T dS^T).
//   dkdv     : S[query][key] = Q K^T    -> key on the lane; P and dS are the B operands of
//              dV^T += dO^T P and dK^T += Q^T dS.
// The "transposed" A operands (V^T, K^T, dO^T, Q^T) are read from the row-major LDS tiles
// with ds_read_b64_tr_b16.  An accumulator used as an operand has a permuted k order:
// element j of lane-half h is row 16s + 8(j>>2) + 4h + (j&3) of k-step s; the tr reads
// below fetch exactly those rows.
// K/V (resp. Q/dO) tiles are double-buffered in LDS, staged HBM -> VGPR -> LDS with the
// next tile's loads in flight during the MFMAs; one barrier per tile.
// dQ is produced by its own pass (recomputing S and dP) instead of fp32 atomics: it is
// deterministic, and attention is < 3 % of the step's FLOPs.
#include <atomic>
#include "attn_common.h"
#include "../../include/zero_hip.h"

using namespace zkattn;

namespace {

// =========================================================================================
// forward and dQ share the K/V stream: 64-key tiles (K rows then V rows, unpadded, chunk-swizzled,
// see swz_key) brought in by LDS-DMA into a ring of NST stages; 4 waves x 32 queries per block,
// two blocks per CU.  One barrier per tile: "tile t landed everywhere and tile t-1 is consumed",
// after which the stage of tile t-1 is refilled with tile t+NST-1.
// =========================================================================================
template <int HD, int NST_>
struct KvRing {
  static constexpr int KV = 64;
  static constexpr int RB = HD * 2;
  static constexpr int TILE = KV * RB;
  static constexpr int STAGE = 2 * TILE;
  static constexpr int NST = NST_;   // hd 128: 2 stages (64 KiB, two blocks per CU); hd 64: 2 in the forward (32 KiB, four
                                     // blocks per CU at 128 VGPRs: 165 -> 158 us at B = 64), 3 in dQ (48 KiB, three blocks)
  static constexpr int NQ = KV * (HD / 8) / 256;           // 16-byte DMA instructions per thread per operand tile
  static constexpr int PER = 2 * NQ;

  __device__ static __forceinline__ void issue(unsigned char* smem, __amdgpu_buffer_rsrc_t rk, __amdgpu_buffer_rsrc_t rv,
                                               int t, int S, int ldk, int ldv, int col0, int tid, int wv) {
    unsigned char* st = smem + (t % NST) * STAGE;
    const int kn = t * KV;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int sidx = tid + 256 * i;
      const int row = sidx / (HD / 8), ch = (sidx % (HD / 8)) ^ swz_key<HD>(row);
      const bool ok = kn + row < S;
      unsigned char* dst = st + (wv * 64 + 256 * i) * 16;   // wave-uniform; the hardware adds lane * 16
      lds_dma<16>(rk, dst,
                                               ok ? (unsigned)(((kn + row) * ldk + col0 + ch * 8) * 2) : 0xFFFFFFF0u);
      lds_dma<16>(rv, (dst + TILE),
                                               ok ? (unsigned)(((kn + row) * ldv + col0 + ch * 8) * 2) : 0xFFFFFFF0u);
    }
  }
  // at most n tiles may stay in flight (n <= NST - 2)
  __device__ static __forceinline__ void wait_tiles(int n) {
    if (NST >= 4 && n >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else if (NST >= 3 && n >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
};

#ifdef ZK_ATTN_STAMPS
// diagnostic build only (tools/attn_stamps.py): where a forward workgroup's time goes.  100 MHz wall-clock stamps of
// wave 0 at entry / first tile landed / loop end / stores issued / stores drained, in a buffer nothing else reads.
static __device__ unsigned long long g_stamps[8 * 16384];
#define ZK_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 16384) g_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
static __device__ unsigned long long g_stamps_dq[8 * 16384], g_stamps_dkdv[8 * 16384];
#define ZK_STAMP_A(arr, k) do { if (threadIdx.x == 0 && blockIdx.x < 16384) arr[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define ZK_STAMP_V(arr, k, v) do { if (threadIdx.x == 0 && blockIdx.x < 16384) arr[blockIdx.x * 8 + (k)] = (unsigned long long)(v); } while (0)
#else
#define ZK_STAMP(k) do { } while (0)
#define ZK_STAMP_A(arr, k) do { } while (0)
#define ZK_STAMP_V(arr, k, v) do { } while (0)
#endif

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, HD == 64 ? 4 : 2) void attn_fwd_kernel(AttnArgs a) {
  ZK_STAMP(0);
  using C = Cfg<HD>;
  using R = KvRing<HD, 2>;
  constexpr int KV = R::KV;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[R::NST * R::STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  // 1-D grid, heaviest (causal: latest) query blocks first, so the tail of the launch is made of
  // the short blocks.  (A 3-D grid with the query block on x interleaves heavy and light blocks,
  // and with 6 query blocks the round-robin over 8 XCDs hands even XCDs only the even ones.)
  const int nbh = a.Hq * a.B;
  const int qblk = (int)(gridDim.x / nbh) - 1 - (int)(blockIdx.x / nbh);
  const int head = (blockIdx.x % nbh) % a.Hq, b = (blockIdx.x % nbh) / a.Hq;
  const int kvh = head / (a.Hq / a.Hkv);
  const int q0 = qblk * 128 + wave * 32;
  const int qi = q0 + (lane & 31);
  const float sc = a.scale * LOG2E;

  __amdgpu_buffer_rsrc_t rk = make_rsrc(a.k + (size_t)b * a.S * a.ldk, 0x7FFFFFF0u);
  __amdgpu_buffer_rsrc_t rv = make_rsrc(a.v + (size_t)b * a.S * a.ldv, 0x7FFFFFF0u);
  int kv_end = a.S;
  if (CAUSAL) kv_end = min(a.S, qblk * 128 + 128);
  const int nt = (kv_end + KV - 1) / KV;
  int nt_wave = nt;
  if (CAUSAL) nt_wave = min(nt, (q0 + 32 + KV - 1) / KV);
  for (int t = 0; t < R::NST - 1 && t < nt; ++t) R::issue(smem, rk, rv, t, a.S, a.ldk, a.ldv, kvh * HD, tid, wv);

  bf16x8 qf[C::DS];
  {
    const bf16_t* qp = a.q + ((size_t)b * a.S + qi) * a.ldq + head * HD + 8 * h;
#pragma unroll
    for (int ds = 0; ds < C::DS; ++ds) qf[ds] = gfrag(qp + ds * 16, qi < a.S);
  }
  f32x16 o[C::DB];
#pragma unroll
  for (int d = 0; d < C::DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;   // running max in the scaled log2 domain
  // Q has arrived before the loop starts: a first use INSIDE the loop makes the compiler wait there with vmcnt(0) on
  // every iteration (it cannot count across the back edge), which would also drain the K/V DMA it does not know about
#pragma unroll
  for (int ds = 0; ds < C::DS; ++ds) asm volatile("" ::"v"(qf[ds]));

  for (int t = 0; t < nt; ++t) {
    R::wait_tiles(min(R::NST - 2, nt - 1 - t));
    __syncthreads();
    if (t == 0) ZK_STAMP(1);
    if (t + R::NST - 1 < nt) R::issue(smem, rk, rv, t + R::NST - 1, a.S, a.ldk, a.ldv, kvh * HD, tid, wv);
    if (t >= nt_wave) continue;
    const unsigned char* kl = smem + (t % R::NST) * R::STAGE;
    const unsigned char* vl = kl + R::TILE;
    const int k0 = t * KV;
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int ds = 0; ds < C::DS; ++ds)
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_sw<HD>(kl, kb * 32, ds, lane), qf[ds], s[kb], 0, 0, 0);
    }
    // raw scores; the softmax scale rides in the exponent's fma
    if ((k0 + KV > a.S) || (CAUSAL && (k0 + KV - 1 > q0))) {   // wave-uniform: only edge / diagonal tiles mask
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + kb * 32 + acc_row(r, h);
          if (key >= a.S || (CAUSAL && key > qi)) s[kb][r] = -INFINITY;
        }
    }
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s[0][r], s[1][r]));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * sc);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = fast_exp2(m_run - m_use);
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(__builtin_fmaf(s[kb][r], sc, -m_use));
        s[kb][r] = p;
        rs += p;
      }
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < C::DB; ++d) o[d] = o[d] * alpha;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        bf16x8 pf = pack_acc(s[kb], st);
#pragma unroll
        for (int d = 0; d < C::DB; ++d)
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_sw<HD>(vl, kb * 32, st, d, lane), pf, o[d], 0, 0, 0);
      }
  }

  ZK_STAMP(2);
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = (l_tot > 0.f) ? 1.0f / l_tot : 0.f;
  if (qi < a.S) {
    bf16_t* op = a.out + ((size_t)b * a.S + qi) * a.ldo + head * HD;
#pragma unroll
    for (int d = 0; d < C::DB; ++d)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        u32x2 w;
        w[0] = pack2bf(o[d][4 * r4] * inv, o[d][4 * r4 + 1] * inv);
        w[1] = pack2bf(o[d][4 * r4 + 2] * inv, o[d][4 * r4 + 3] * inv);
        *reinterpret_cast<u32x2*>(op + d * 32 + 8 * r4 + 4 * h) = w;
      }
    if (h == 0 && a.lse) a.lse[((size_t)b * a.Hq + head) * a.S + qi] = m_run * LN2 + logf(l_tot);
  }
  ZK_STAMP(3);
#ifdef ZK_ATTN_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ZK_STAMP(4);
  if (threadIdx.x == 0 && blockIdx.x < 16384) { g_stamps[blockIdx.x * 8 + 5] = (unsigned long long)qblk; g_stamps[blockIdx.x * 8 + 6] = (unsigned long long)nt; }
#endif
}

#ifdef ZK_ATTN_STAMPS
extern "C" int zk_debug_read_stamps(void* host_dst, int nblocks) {
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_stamps), (size_t)nblocks * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -5;
}
extern "C" int zk_debug_read_stamps_bwd(void* host_dst, int nblocks, int which) {
  const size_t n = (size_t)nblocks * 8 * sizeof(unsigned long long);
  const hipError_t e = which == 0 ? hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_stamps_dq), n)
                                  : hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_stamps_dkdv), n);
  return e == hipSuccess ? 0 : -5;
}
#endif

// =========================================================================================
// backward: dQ  (per 128-query block, sweep over key tiles; same K/V ring as the forward)
// =========================================================================================
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, HD == 64 ? 3 : 2) void attn_bwd_dq_kernel(AttnArgs a) {
  ZK_STAMP_A(g_stamps_dq, 0);
  using C = Cfg<HD>;
  using R = KvRing<HD, HD == 128 ? 2 : 3>;
  constexpr int KV = R::KV;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[R::NST * R::STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int nbh = a.Hq * a.B;   // 1-D grid, heaviest query blocks first (see attn_fwd_kernel)
  const int qblk = (int)(gridDim.x / nbh) - 1 - (int)(blockIdx.x / nbh);
  const int head = (blockIdx.x % nbh) % a.Hq, b = (blockIdx.x % nbh) / a.Hq;
  const int kvh = head / (a.Hq / a.Hkv);
  const int q0 = qblk * 128 + wave * 32;
  const int qi = q0 + (lane & 31);
  const float sc = a.scale * LOG2E;
  const bool qok = qi < a.S;

  __amdgpu_buffer_rsrc_t rk = make_rsrc(a.k + (size_t)b * a.S * a.ldk, 0x7FFFFFF0u);
  __amdgpu_buffer_rsrc_t rv = make_rsrc(a.v + (size_t)b * a.S * a.ldv, 0x7FFFFFF0u);
  int kv_end = a.S;
  if (CAUSAL) kv_end = min(a.S, qblk * 128 + 128);
  const int nt = (kv_end + KV - 1) / KV;
  int nt_wave = nt;
  if (CAUSAL) nt_wave = min(nt, (q0 + 32 + KV - 1) / KV);
  for (int t = 0; t < R::NST - 1 && t < nt; ++t) R::issue(smem, rk, rv, t, a.S, a.ldk, a.ldv, kvh * HD, tid, wv);

  bf16x8 qf[C::DS], dof[C::DS];
  {
    const bf16_t* qp = a.q + ((size_t)b * a.S + qi) * a.ldq + head * HD + 8 * h;
    const bf16_t* dp = a.d_o + ((size_t)b * a.S + qi) * a.ldo + head * HD + 8 * h;
#pragma unroll
    for (int ds = 0; ds < C::DS; ++ds) { qf[ds] = gfrag(qp + ds * 16, qok); dof[ds] = gfrag(dp + ds * 16, qok); }
  }
  // rows past S: q = dO = 0 gives s = dP = 0, p = 1, dS = 0: harmless, and never stored
  const size_t stat = ((size_t)b * a.Hq + head) * a.S + (qok ? qi : 0);
  const float lse2 = qok ? a.lse_in[stat] * LOG2E : 0.f;
  // delta = rowsum(dO * O) is formed here (each row belongs to exactly one lane pair of one workgroup) and
  // written for the dK/dV kernel, which runs after this one: no separate pass over O and dO
  float dlt = 0.f;
  {
    const bf16_t* op_ = a.o + ((size_t)b * a.S + qi) * a.ldo + head * HD + 8 * h;
#pragma unroll
    for (int ds = 0; ds < C::DS; ++ds) {
      const bf16x8 of = gfrag(op_ + ds * 16, qok);
#pragma unroll
      for (int j = 0; j < 8; ++j) dlt = __builtin_fmaf((float)of[j], (float)dof[ds][j], dlt);
    }
    dlt += __shfl_xor(dlt, 32, 64);
    if (qok && h == 0) a.delta_out[stat] = dlt;
  }

  f32x16 dq[C::DB];
#pragma unroll
  for (int d = 0; d < C::DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[d][r] = 0.f;
  // (every register the prologue loaded is waited for here, not at its first use inside the loop: see attn_fwd_kernel)
#pragma unroll
  for (int ds = 0; ds < C::DS; ++ds) asm volatile("" ::"v"(qf[ds]), "v"(dof[ds]));
  asm volatile("" ::"v"(lse2), "v"(dlt));
  ZK_STAMP_A(g_stamps_dq, 1);
  ZK_STAMP_V(g_stamps_dq, 5, qblk); ZK_STAMP_V(g_stamps_dq, 6, nt);

  for (int t = 0; t < nt; ++t) {
    R::wait_tiles(min(R::NST - 2, nt - 1 - t));
    __syncthreads();
    if (t + R::NST - 1 < nt) R::issue(smem, rk, rv, t + R::NST - 1, a.S, a.ldk, a.ldv, kvh * HD, tid, wv);
    if (t >= nt_wave) continue;
    const unsigned char* kl = smem + (t % R::NST) * R::STAGE;
    const unsigned char* vl = kl + R::TILE;
    const int k0 = t * KV;
    const bool need_mask = (k0 + KV > a.S) || (CAUSAL && (k0 + KV - 1 > q0));   // wave-uniform
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ds = 0; ds < C::DS; ++ds) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_sw<HD>(kl, kb * 32, ds, lane), qf[ds], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_sw<HD>(vl, kb * 32, ds, lane), dof[ds], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = fast_exp2(__builtin_fmaf(s[r], sc, -lse2));
        if (need_mask) {
          const int key = k0 + kb * 32 + acc_row(r, h);
          if (key >= a.S || (CAUSAL && key > qi)) p = 0.f;
        }
        s[r] = p * (dp[r] - dlt);   // dS^T (without the softmax scale)
      }
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        bf16x8 df = pack_acc(s, st);
#pragma unroll
        for (int d = 0; d < C::DB; ++d)
          dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_sw<HD>(kl, kb * 32, st, d, lane), df, dq[d], 0, 0, 0);
      }
    }
  }
  ZK_STAMP_A(g_stamps_dq, 2);
  // dQ leaves through LDS as whole rows (a lane owns 8 bytes of 16 places in one row of the accumulator layout: 32 rows x 16 bytes per store instruction otherwise): the ring is idle once every wave has
  // passed its last tile; each wave owns a [32][HD] bf16 patch, chunk c of row q at chunk c ^ (q % chunks-per-row)
  constexpr int RPI = 64 / C::CPR;                 // rows per wave-instruction
  // rotary tables for the rows this lane stores: requested here, ahead of the barrier and the LDS round trip,
  // so the pass loop below waits for them once instead of once per pass.  The sines are stored signed for the
  // INVERSE rotation (rope_chunk): +sin for the lower half's lanes, -sin for the upper half's
  // (hd 128: 8 passes x 16 values would not fit beside the live accumulators — the compiler spilled — so the
  // second half is requested once the accumulators have gone to LDS)
  constexpr int NPS = 32 / RPI, NPRE = NPS > 4 ? 4 : NPS;
  float rcs[NPS][8], rsn[NPS][8];
  auto load_tab = [&](auto ps_c) {
    constexpr int ps = decltype(ps_c)::value;
    const int row = min(q0 + ps * RPI + lane / C::CPR, a.S - 1);
    const size_t t = (size_t)row * (HD / 2) + ((lane % C::CPR) & (C::CPR / 2 - 1)) * 8;
#pragma unroll
    for (int j4 = 0; j4 < 2; ++j4) {
      const f32x4 c4 = *reinterpret_cast<const f32x4*>(a.cos_t + t + 4 * j4);
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(a.sin_t + t + 4 * j4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { rcs[ps][4 * j4 + j] = c4[j]; rsn[ps][4 * j4 + j] = (lane % C::CPR) >= C::CPR / 2 ? -s4[j] : s4[j]; }
    }
  };
  if (a.cos_t) static_for<NPRE>([&](auto ps_c) { load_tab(ps_c); });
  __syncthreads();
  {
    unsigned char* ob = smem + wave * (32 * HD * 2);
    const int ql = lane & 31;
#pragma unroll
    for (int d = 0; d < C::DB; ++d)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        u32x2 w;
        w[0] = pack2bf(dq[d][4 * r4] * a.scale, dq[d][4 * r4 + 1] * a.scale);
        w[1] = pack2bf(dq[d][4 * r4 + 2] * a.scale, dq[d][4 * r4 + 3] * a.scale);
        *reinterpret_cast<u32x2*>(ob + ql * (HD * 2) + (((d * 4 + r4) ^ (ql & (C::CPR - 1))) << 4) + 8 * h) = w;
      }
    if (a.cos_t) static_for<NPS - NPRE>([&](auto k_c) { load_tab(std::integral_constant<int, NPRE + decltype(k_c)::value>{}); });
    bf16_t* op = a.dq + ((size_t)b * a.S + q0) * a.lddq + head * HD;
    static_for<NPS>([&](auto ps_c) {
      constexpr int ps = decltype(ps_c)::value;
      const int row = ps * RPI + lane / C::CPR, ch = lane % C::CPR;
      u32x4 v = *reinterpret_cast<const u32x4*>(ob + row * (HD * 2) + ((ch ^ (row & (C::CPR - 1))) << 4));
      if (q0 + row < a.S) {
        if (a.cos_t) {   // the rounded row is rotated exactly as zk_rope(inverse) would rotate it in HBM
          const u32x4 pv = *reinterpret_cast<const u32x4*>(ob + row * (HD * 2) + (((ch ^ (C::CPR / 2)) ^ (row & (C::CPR - 1))) << 4));
          v = rope_chunk(v, pv, rcs[ps], rsn[ps]);
        }
        *reinterpret_cast<u32x4*>(op + (size_t)row * a.lddq + ch * 8) = v;
      }
    });
  }
  ZK_STAMP_A(g_stamps_dq, 3);
}

// =========================================================================================
// backward: dK, dV  (per 128-key block of one kv head, sweep over its q heads x 64-query tiles)
//
// 8 waves: wave (g, w) owns keys 32w..32w+31 of the block and the rows 32g..32g+31 of every
// 64-query tile, so two waves per SIMD work on independent halves (one's exp/VALU phase overlaps
// the other's MFMAs); their dK/dV partial sums meet in LDS at the end.  The Q / dO tiles and the
// per-row lse / delta arrive by LDS-DMA (buffer_load ... lds, no VGPR staging) into a 4-stage
// ring: three tiles in flight, one barrier per tile, counted s_waitcnt vmcnt.  The LDS image of a
// tile is unpadded (the DMA writes 1 KiB lane-linear per wave-instruction); bank conflicts of
// both read kinds (b128 row fragments, b64 transposed fragments) are removed by XOR-swizzling
// the 16-byte chunk index on the DMA's SOURCE address and again on the reads.
// =========================================================================================
// GW = query halves spread ACROSS waves: 2 = the 8-wave block above; 1 = a 4-wave block whose waves walk both 32-row
// halves of a tile themselves (hd 64: two independent S / dP / softmax chains per wave and tile give the in-order
// wave something to issue while the other chain waits, two blocks per CU, and the halves already meet in the
// accumulators, so the exchange only reshapes).
template <int HD, bool CAUSAL, int GW>
__global__ __launch_bounds__(256 * GW, 2) void attn_bwd_dkdv_kernel(AttnArgs a) {
  ZK_STAMP_A(g_stamps_dkdv, 0);
  using C = Cfg<HD>;
  constexpr int QT = 64, NST = 3, NT = 256 * GW;
  constexpr int RB = HD * 2;                       // LDS row bytes (unpadded)
  constexpr int TILE = QT * RB;
  constexpr int STAGE = 2 * TILE + 2048;           // Q tile, dO tile, 8 x 256 B of per-wave dword DMA (lse, delta, 6 unused)
  constexpr int VBYTES = 128 * RB;                 // the block's V rows stay in LDS (their fragments would not fit in VGPRs beside K's)
  constexpr int NQ = QT * C::CPR / NT;             // 16-byte DMA instructions per thread per tile (Q; same for dO)
  static_assert(NQ * NT == QT * C::CPR && (128 * C::CPR) % NT == 0, "tile pieces per thread");
  constexpr int PER = 2 * NQ + 1;
  constexpr int XROW = HD * 4 + 16;                // exchange patch row (fp32 + pad)
  static_assert(NST * STAGE + VBYTES >= 4 * GW * 32 * XROW, "the dK/dV exchange reuses the ring");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE + VBYTES];
  unsigned char* const vl = smem + NST * STAGE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int g0 = GW == 2 ? wave >> 2 : 0, w = wave & 3;
  const int nbh = a.Hkv * a.B;  // 1-D grid, heaviest (causal: earliest) key blocks first
  const int kblk = CAUSAL ? (int)(blockIdx.x / nbh) : (int)(gridDim.x / nbh) - 1 - (int)(blockIdx.x / nbh);
  const int kvh = (blockIdx.x % nbh) % a.Hkv, b = (blockIdx.x % nbh) / a.Hkv;
  const int rep = a.Hq / a.Hkv;
  const int k0w = kblk * 128 + w * 32;
  const int ki = k0w + (lane & 31);
  const bool kok = ki < a.S;
  const float sc = a.scale * LOG2E;

  bf16x8 kf[C::DS];
  {
    const bf16_t* kp = a.k + ((size_t)b * a.S + ki) * a.ldk + kvh * HD + 8 * h;
#pragma unroll
    for (int ds = 0; ds < C::DS; ++ds) kf[ds] = gfrag(kp + ds * 16, kok);
  }
  f32x16 dk[C::DB], dv[C::DB];
#pragma unroll
  for (int d = 0; d < C::DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[d][r] = 0.f; dv[d][r] = 0.f; }

  const int nqt_all = (a.S + QT - 1) / QT;
  const int qt0 = CAUSAL ? (kblk * 128) / QT : 0;   // queries before the block's first key see none of it
  const int nqt = nqt_all - qt0;
  const int niter = rep * nqt;

  // buffer resources over this batch element's rows (32-bit byte offsets; checked on the host)
  __amdgpu_buffer_rsrc_t rq = make_rsrc(a.q + (size_t)b * a.S * a.ldq, 0x7FFFFFF0u);
  __amdgpu_buffer_rsrc_t rdo = make_rsrc(a.d_o + (size_t)b * a.S * a.ldo, 0x7FFFFFF0u);
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  __amdgpu_buffer_rsrc_t rstat = make_rsrc((wv == 0 ? a.lse_in : a.delta) + (size_t)b * a.Hq * a.S, 0x7FFFFFF0u);
  auto issue = [&](int it) {
    unsigned char* st = smem + (it % NST) * STAGE;
    const int hq = kvh * rep + it / nqt;
    const int qb = (qt0 + it % nqt) * QT;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int sidx = tid + NT * i;
      const int row = sidx / C::CPR, ch = (sidx % C::CPR) ^ swz_key<HD>(row);
      const bool ok = qb + row < a.S;
      unsigned char* dst = st + ((tid & ~63) + NT * i) * 16;   // wave-uniform; the hardware adds lane * 16
      lds_dma<16>(rq, dst,
                                               ok ? (unsigned)(((qb + row) * a.ldq + hq * HD + ch * 8) * 2) : 0xFFFFFFF0u);
      lds_dma<16>(rdo, (dst + TILE),
                                               ok ? (unsigned)(((qb + row) * a.ldo + hq * HD + ch * 8) * 2) : 0xFFFFFFF0u);
    }
    // one dword per lane: wave 0 brings lse[qb..qb+63], wave 1 delta[...]; the others a zero fill
    // (every wave issues the same number of loads, so one counted vmcnt serves all)
    const bool sok = wv < 2 && qb + lane < a.S;
    lds_dma<4>(rstat, (st + 2 * TILE + wv * 256),
                                               sok ? (unsigned)((hq * a.S + qb + lane) * 4) : 0xFFFFFFF0u);
  };
  auto wait_tiles = [&](int n) {   // at most n tiles may stay in flight
    if (n >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  static_assert(NST == 3, "wait_tiles covers NST - 2 = 1 tile in flight");
  {
    // V rows of the block, same swizzled image (issued first: landed before tile 0 is waited for)
    __amdgpu_buffer_rsrc_t rv = make_rsrc(a.v + (size_t)b * a.S * a.ldv, 0x7FFFFFF0u);
#pragma unroll
    for (int i = 0; i < 128 * C::CPR / NT; ++i) {
      const int sidx = tid + NT * i;
      const int row = sidx / C::CPR, ch = (sidx % C::CPR) ^ swz_key<HD>(row);
      const bool ok = kblk * 128 + row < a.S;
      lds_dma<16>(rv, (vl + ((tid & ~63) + NT * i) * 16),
                                               ok ? (unsigned)(((kblk * 128 + row) * a.ldv + kvh * HD + ch * 8) * 2) : 0xFFFFFFF0u);
    }
  }
  for (int it = 0; it < NST - 1 && it < niter; ++it) issue(it);
  // (K fragments waited for here, not at their first use inside the loop: see attn_fwd_kernel)
#pragma unroll
  for (int ds = 0; ds < C::DS; ++ds) asm volatile("" ::"v"(kf[ds]));
  ZK_STAMP_A(g_stamps_dkdv, 1);
  ZK_STAMP_V(g_stamps_dkdv, 5, kblk); ZK_STAMP_V(g_stamps_dkdv, 6, niter);

  for (int it = 0; it < niter; ++it) {
    // tile `it` has landed (each wave waits for its own pieces, the barrier covers the others'),
    // and every wave is done with tile it-1, whose stage takes tile it+3
    wait_tiles(min(NST - 2, niter - 1 - it));
    __syncthreads();
    if (it + NST - 1 < niter) issue(it + NST - 1);
    const unsigned char* st = smem + (it % NST) * STAGE;
    const unsigned char* ql = st;
    const unsigned char* dol = st + TILE;
    const float* lsl = reinterpret_cast<const float*>(st + 2 * TILE);
    const float* dll = reinterpret_cast<const float*>(st + 2 * TILE + 256);
#pragma unroll
    for (int gg = 0; gg < 3 - GW; ++gg) {
    const int g = g0 + gg;
    const int qb = (qt0 + it % nqt) * QT + 32 * g;   // this wave's 32 rows
    // causal: this wave's keys are all after its last query -> nothing to do; same for rows past S.  (A wave that
    // walks both halves keeps them in one basic block when nothing is causal: rows past S arrive as zeros, their
    // p is forced to 0 below, and the compiler may interleave the two chains.)
    const bool skip = (CAUSAL && (k0w > qb + 31)) || ((GW == 2 || CAUSAL) && qb >= a.S);
    if (!skip) {
      // The ~40 swizzled fragment addresses are loop-invariant; hoisted they cost 40 VGPRs the
      // kernel does not have (spills, whose scratch loads would also break the counted vmcnt).
      // An opaque copy of the lane id makes the compiler recompute them (2 VALU ops per read).
      int ln = lane;
      asm volatile("" : "+v"(ln));
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ds = 0; ds < C::DS; ++ds) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_sw<HD>(ql, 32 * g, ds, ln), kf[ds], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag_sw<HD>(dol, 32 * g, ds, ln), row_frag_sw<HD>(vl, 32 * w, ds, ln), dp, 0, 0, 0);
      }
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const f32x4 l4 = *reinterpret_cast<const f32x4*>(lsl + 32 * g + 8 * r4 + 4 * h);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(dll + 32 * g + 8 * r4 + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * r4 + e;
          const int qidx = qb + 8 * r4 + 4 * h + e;
          float p = fast_exp2(s[r] * sc - l4[e] * LOG2E);
          if (!kok || qidx >= a.S || (CAUSAL && ki > qidx)) p = 0.f;
          s[r] = p;
          dp[r] = p * (dp[r] - d4[e]);
        }
      }
#pragma unroll
      for (int stp = 0; stp < 2; ++stp) {
        bf16x8 pf = pack_acc(s, stp);
        bf16x8 df = pack_acc(dp, stp);
#pragma unroll
        for (int d = 0; d < C::DB; ++d) {
          dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_sw<HD>(dol, 32 * g, stp, d, ln), pf, dv[d], 0, 0, 0);
          dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag_sw<HD>(ql, 32 * g, stp, d, ln), df, dk[d], 0, 0, 0);
        }
      }
    }
    }
  }

  ZK_STAMP_A(g_stamps_dkdv, 2);
  // ---- the two query halves meet: every wave parks one accumulator set in its fp32 patch
  // ([32 keys][HD + pad]); wave (g, w) then sums rows 16g..16g+15 of patches (0, w) and (1, w)
  // and writes whole rows (8 consecutive d per lane).  dK first, then dV through the same patches.
  float* patch = reinterpret_cast<float*>(smem + wave * (32 * XROW));
  const float* p0 = reinterpret_cast<const float*>(smem + w * (32 * XROW));
  const float* p1 = reinterpret_cast<const float*>(smem + ((GW == 2 ? 4 : 0) + w) * (32 * XROW));   // (GW 1: unused)
  constexpr int LPK = HD / 8;          // lanes per key row
  constexpr int KPP = 64 / LPK;        // keys per pass
  constexpr int RPW = 32 / GW;         // key rows a wave finishes
  // rotary tables of the dK rows this lane stores, requested ahead of the exchange (see the dQ kernel)
  float rcs[RPW / KPP][8], rsn[RPW / KPP][8];
  if (a.cos_t) {
#pragma unroll
    for (int ps = 0; ps < RPW / KPP; ++ps) {
      const int key = min(kblk * 128 + w * 32 + RPW * g0 + ps * KPP + lane / LPK, a.S - 1);
      const size_t t = (size_t)key * (HD / 2) + ((lane % LPK) & (LPK / 2 - 1)) * 8;
#pragma unroll
      for (int j4 = 0; j4 < 2; ++j4) {
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(a.cos_t + t + 4 * j4);
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(a.sin_t + t + 4 * j4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { rcs[ps][4 * j4 + j] = c4[j]; rsn[ps][4 * j4 + j] = (lane % LPK) >= LPK / 2 ? -s4[j] : s4[j]; }
      }
    }
  }
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    __syncthreads();                   // ring (or the previous exchange) no longer read
#pragma unroll
    for (int d = 0; d < C::DB; ++d)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const f32x16& src = which == 0 ? dk[d] : dv[d];
        f32x4 v4 = {src[4 * r4], src[4 * r4 + 1], src[4 * r4 + 2], src[4 * r4 + 3]};
        *reinterpret_cast<f32x4*>(patch + (lane & 31) * (XROW / 4) + d * 32 + 8 * r4 + 4 * h) = v4;
      }
    __syncthreads();
    const float mul = which == 0 ? a.scale : 1.0f;
    bf16_t* outp = which == 0 ? a.dk : a.dv;
    const int ldo_ = which == 0 ? a.lddk : a.lddv;
#pragma unroll
    for (int ps = 0; ps < RPW / KPP; ++ps) {
      const int kr = RPW * g0 + ps * KPP + lane / LPK, ch = lane % LPK;
      const int key = kblk * 128 + w * 32 + kr;
      if (key >= a.S) continue;
      auto sum8 = [&](int c) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(p0 + kr * (XROW / 4) + c * 8);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(p0 + kr * (XROW / 4) + c * 8 + 4);
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (GW == 2) {
          b0 = *reinterpret_cast<const f32x4*>(p1 + kr * (XROW / 4) + c * 8);
          b1 = *reinterpret_cast<const f32x4*>(p1 + kr * (XROW / 4) + c * 8 + 4);
        }
        u32x4 r;
        r[0] = pack2bf((a0[0] + b0[0]) * mul, (a0[1] + b0[1]) * mul);
        r[1] = pack2bf((a0[2] + b0[2]) * mul, (a0[3] + b0[3]) * mul);
        r[2] = pack2bf((a1[0] + b1[0]) * mul, (a1[1] + b1[1]) * mul);
        r[3] = pack2bf((a1[2] + b1[2]) * mul, (a1[3] + b1[3]) * mul);
        return r;
      };
      u32x4 o4 = sum8(ch);
      if (which == 0 && a.cos_t) {   // dK: the rounded row rotated as zk_rope(inverse) would
        o4 = rope_chunk(o4, sum8(ch ^ (LPK / 2)), rcs[ps], rsn[ps]);
      }
      *reinterpret_cast<u32x4*>(outp + ((size_t)b * a.S + key) * ldo_ + kvh * HD + ch * 8) = o4;
    }
  }
  ZK_STAMP_A(g_stamps_dkdv, 3);
}

bool bad_attn(int64_t B, int64_t S, int64_t Hq, int64_t Hkv, int64_t hd, int64_t l0, int64_t l1, int64_t l2, int64_t l3) {
  if (B <= 0 || S <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv) return true;
  if (hd != 64 && hd != 128) return true;
  if ((l0 | l1 | l2 | l3) & 7) return true;
  if (((S + 127) / 128) * Hq * B > 0x7FFFFFFFLL) return true;
  return false;
}

// =========================================================================================
// Single-query attention over a KV cache (greedy decoding, tf generate(use_cache=True):
// ref:inference/vlm_infer.py:259-270).  One block per (sequence, query head): HBM-bound byte work
// (the cache rows are read once), no MFMA.  Pass 1: a thread per key, q broadcast from LDS, 16-byte
// key loads -> scores in LDS, block max / sum; pass 2: 16-byte V loads, a wave covers 64/(hd/8) keys per
// instruction, partial sums merged by lane shuffles and through LDS.
// =========================================================================================
template <int HD>
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* q, const bf16_t* kc, const bf16_t* vc,
                                                          bf16_t* out, const int* lens, int Hq, int Hkv, int ldq,
                                                          int ldk, int ldv, int ldo, long long bsk, long long bsv,
                                                          float scale) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  float* qs = dsm;                 // [HD]
  float* red = dsm + HD;           // [4 * HD] (also the 4-entry block reductions)
  float* sc = dsm + 5 * HD;        // [len]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / Hq, h = blockIdx.x % Hq;
  const int kvh = h / (Hq / Hkv);
  const int len = lens[b];
  const bf16_t* kb = kc + (size_t)b * bsk + (size_t)kvh * HD;
  const bf16_t* vb = vc + (size_t)b * bsv + (size_t)kvh * HD;
  if (tid < HD) qs[tid] = bf2f(q[(size_t)b * ldq + (size_t)h * HD + tid]) * scale;
  __syncthreads();
  float mx = -3.0e38f;
  for (int j = tid; j < len; j += 256) {
    const bf16_t* kr = kb + (size_t)j * ldk;
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < HD / 8; ++c) {
      const u32x4 kv = *reinterpret_cast<const u32x4*>(kr + c * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) dot += lo_bf(kv[e]) * qs[c * 8 + 2 * e] + hi_bf(kv[e]) * qs[c * 8 + 2 * e + 1];
    }
    sc[j] = dot;
    mx = fmaxf(mx, dot);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int j = tid; j < len; j += 256) {
    const float p = __expf(sc[j] - mx);
    sc[j] = p;
    sum += p;
  }
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  __syncthreads();
  // pass 2: LPR = HD/8 lanes cover one V row with 16-byte loads, so a wave takes 64/LPR keys per
  // instruction and the block 4x that; the loads of 4 steps are in flight together
  constexpr int LPR = HD / 8, KS = 64 / LPR;
  const int dch = lane % LPR, ks = lane / LPR;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int j = wave * KS + ks; j < len; j += 4 * KS) {
    const u32x4 vv = *reinterpret_cast<const u32x4*>(vb + (size_t)j * ldv + dch * 8);
    const float p = sc[j];
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[2 * e] += p * lo_bf(vv[e]); acc[2 * e + 1] += p * hi_bf(vv[e]); }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) acc[e] += __shfl_xor(acc[e], o, 64);
  }
  if (lane < LPR) {
#pragma unroll
    for (int e = 0; e < 8; ++e) red[wave * HD + dch * 8 + e] = acc[e];
  }
  __syncthreads();
  if (tid < HD) {
    const float v = (red[tid] + red[HD + tid] + red[2 * HD + tid] + red[3 * HD + tid]) * inv;
    out[(size_t)b * ldo + (size_t)h * HD + tid] = f2bf(v);
  }
}

// =========================================================================================
// The same single-query attention, split over the cache (flash-decoding) and with the new token's
// RoPE + cache append folded in: one block per (64-key chunk, kv head, sequence) instead of one per
// query head, so a tensor-parallel shard with one kv head (70B / 8) spreads over len/64 CUs instead of
// 8, and the q heads of a GQA group share every K / V row they read.  Partials (un-normalised output,
// chunk max, chunk sum) go to the caller's fp32 workspace; a second small kernel merges them.
//   qkv: the new token's packed q|k|v rows BEFORE RoPE (tf:llama:254-259); the block of the last
//   chunk rotates k, appends k / v to the caches (row `pos`) and then reads them like any other row.
// =========================================================================================
template <int HD>
__global__ __launch_bounds__(256) void attn_decode_split_kernel(const bf16_t* qkv, int ldqkv, const float* cosr,
                                                                const float* sinr, bf16_t* kc, bf16_t* vc,
                                                                long long bstride, int kvw, float* ws, int Hq, int Hkv,
                                                                int pos, int nchunks, float scale) {
  constexpr int CK = 64, HALF = HD / 2, LPR = HD / 8, KS = 64 / LPR, MAXG = 16;
  __shared__ float qs[MAXG][HD];
  __shared__ float ps[MAXG][CK];
  __shared__ __attribute__((aligned(16))) bf16_t knew[HD];
  __shared__ __attribute__((aligned(16))) bf16_t vnew[HD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
  const int G = Hq / Hkv;
  const bf16_t* row = qkv + (size_t)b * ldqkv;
  bf16_t* kb = kc + (size_t)b * bstride + (size_t)kvh * HD;
  bf16_t* vb = vc + (size_t)b * bstride + (size_t)kvh * HD;
  const int j0 = chunk * CK;
  const bool last = chunk == nchunks - 1;
  // Every K / V row this thread will need is requested before anything else (the loads do not depend on
  // q or on the scores): one memory round trip for the whole block instead of one per phase.  Row `pos`
  // is not in the cache yet: the last chunk's block patches it in from the new token's k / v below.
  const int jk = j0 + lane;                       // this lane's key (scores)
  const bool valid = jk <= pos;
  const int dch = lane % LPR, sl = lane / LPR;    // this lane's V columns / key slice (output)
  u32x4 kr[HD / 8], vr[CK / KS];
  {
    const bf16_t* kp = kb + (size_t)(valid && jk != pos ? jk : 0) * kvw;
#pragma unroll
    for (int c = 0; c < HD / 8; ++c) kr[c] = *reinterpret_cast<const u32x4*>(kp + c * 8);
#pragma unroll
    for (int i = 0; i < CK / KS; ++i) {
      const int j = j0 + sl + KS * i;
      const u32x4 t = *reinterpret_cast<const u32x4*>(vb + (size_t)(j < pos ? j : 0) * kvw + dch * 8);
      vr[i] = j < pos ? t : u32x4{0u, 0u, 0u, 0u};      // never multiply a stale row (0 x NaN)
    }
  }
  // q of the group: rotate, round to bf16 (what the separate RoPE kernel stores), scale
  for (int i = tid; i < G * HALF; i += 256) {
    const int h = i / HALF, j = i % HALF;
    const bf16_t* qh = row + (size_t)(kvh * G + h) * HD;
    const float a = bf2f(qh[j]), c = bf2f(qh[j + HALF]), cs = cosr[j], sn = sinr[j];
    qs[h][j] = bf2f(f2bf(a * cs - c * sn)) * scale;
    qs[h][j + HALF] = bf2f(f2bf(c * cs + a * sn)) * scale;
  }
  if (last) {      // the chunk that holds row `pos`: rotate k, append k / v, keep them for this block
    const bf16_t* kn = row + (size_t)Hq * HD + (size_t)kvh * HD;
    const bf16_t* vn = row + (size_t)(Hq + Hkv) * HD + (size_t)kvh * HD;
    if (tid < HALF) {
      const float a = bf2f(kn[tid]), c = bf2f(kn[tid + HALF]), cs = cosr[tid], sn = sinr[tid];
      const bf16_t k1 = f2bf(a * cs - c * sn), k2 = f2bf(c * cs + a * sn);
      kb[(size_t)pos * kvw + tid] = k1;
      kb[(size_t)pos * kvw + tid + HALF] = k2;
      knew[tid] = k1;
      knew[tid + HALF] = k2;
    } else if (tid >= 128 && tid < 128 + HD) {
      const bf16_t v = vn[tid - 128];
      vb[(size_t)pos * kvw + (tid - 128)] = v;
      vnew[tid - 128] = v;
    }
  }
  __syncthreads();
  if (last) {
    if (jk == pos) {
#pragma unroll
      for (int c = 0; c < HD / 8; ++c) kr[c] = *reinterpret_cast<const u32x4*>(&knew[c * 8]);
    }
#pragma unroll
    for (int i = 0; i < CK / KS; ++i)
      if (j0 + sl + KS * i == pos) vr[i] = *reinterpret_cast<const u32x4*>(&vnew[dch * 8]);
  }
  float* wsb = ws + ((size_t)(b * Hq + kvh * G) * nchunks + chunk) * (HD + 2);
  const size_t hstride = (size_t)nchunks * (HD + 2);
  for (int h = wave; h < G; h += 4) {
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < HD / 8; ++c) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        dot += lo_bf(kr[c][e]) * qs[h][c * 8 + 2 * e] + hi_bf(kr[c][e]) * qs[h][c * 8 + 2 * e + 1];
    }
    const float sc = valid ? dot : -3.0e38f;
    const float m = wave_max(sc);
    const float p = valid ? __expf(sc - m) : 0.f;
    const float l = wave_sum(p);
    ps[h][lane] = p;                 // 0 for keys beyond pos: their (clamped) V rows add nothing
    if (lane == 0) {
      wsb[h * hstride + HD] = m;
      wsb[h * hstride + HD + 1] = l;
    }
  }
  __syncthreads();
  {
    float acc[4][8];
#pragma unroll
    for (int hh = 0; hh < 4; ++hh)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[hh][e] = 0.f;
#pragma unroll
    for (int i = 0; i < CK / KS; ++i) {
      const int key = sl + KS * i;
#pragma unroll
      for (int hh = 0; hh < 4; ++hh) {
        const int h = wave + 4 * hh;
        if (h < G) {
          const float p = ps[h][key];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[hh][2 * e] += p * lo_bf(vr[i][e]);
            acc[hh][2 * e + 1] += p * hi_bf(vr[i][e]);
          }
        }
      }
    }
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) {
      const int h = wave + 4 * hh;
      if (h < G) {      // wave-uniform
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
          for (int o = LPR; o < 64; o <<= 1) acc[hh][e] += __shfl_xor(acc[hh][e], o, 64);
        }
        if (lane < LPR) {
          float* o = wsb + h * hstride + dch * 8;
          *reinterpret_cast<f32x4*>(o) = f32x4{acc[hh][0], acc[hh][1], acc[hh][2], acc[hh][3]};
          *reinterpret_cast<f32x4*>(o + 4) = f32x4{acc[hh][4], acc[hh][5], acc[hh][6], acc[hh][7]};
        }
      }
    }
  }
}

// out[b][h][:] = sum_c O_c e^(m_c - M) / sum_c l_c e^(m_c - M): one block of HD threads per (b, h)
template <int HD>
__global__ __launch_bounds__(HD) void attn_decode_merge_kernel(const float* ws, bf16_t* out, int ldo, int Hq,
                                                               int nchunks) {
  __shared__ float wgt[256];       // e^(m_c - M) of up to 256 chunks per round
  __shared__ float red[HD / 64 * 2];
  const int b = blockIdx.x / Hq, h = blockIdx.x % Hq, d = threadIdx.x, lane = d & 63, wave = d >> 6;
  const float* w = ws + (size_t)blockIdx.x * nchunks * (HD + 2);
  // chunk maxima: one load per thread and round (independent loads), block max through LDS
  float M = -3.0e38f;
  for (int c = d; c < nchunks; c += HD) M = fmaxf(M, w[(size_t)c * (HD + 2) + HD]);
  M = wave_max(M);
  if (lane == 0) red[wave] = M;
  __syncthreads();
  M = red[0];
#pragma unroll
  for (int i = 1; i < HD / 64; ++i) M = fmaxf(M, red[i]);
  __syncthreads();
  float L = 0.f, o = 0.f;
  for (int c0 = 0; c0 < nchunks; c0 += 256) {
    const int nc = min(256, nchunks - c0);
    for (int c = d; c < nc; c += HD) {
      const float f = __expf(w[(size_t)(c0 + c) * (HD + 2) + HD] - M);
      wgt[c] = f;
      L += w[(size_t)(c0 + c) * (HD + 2) + HD + 1] * f;
    }
    __syncthreads();
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
    int c = 0;
    for (; c + 4 <= nc; c += 4) {      // four independent loads in flight
      o0 += w[(size_t)(c0 + c) * (HD + 2) + d] * wgt[c];
      o1 += w[(size_t)(c0 + c + 1) * (HD + 2) + d] * wgt[c + 1];
      o2 += w[(size_t)(c0 + c + 2) * (HD + 2) + d] * wgt[c + 2];
      o3 += w[(size_t)(c0 + c + 3) * (HD + 2) + d] * wgt[c + 3];
    }
    for (; c < nc; ++c) o0 += w[(size_t)(c0 + c) * (HD + 2) + d] * wgt[c];
    o += (o0 + o1) + (o2 + o3);
    __syncthreads();
  }
  L = wave_sum(L);
  if (lane == 0) red[HD / 64 + wave] = L;
  __syncthreads();
  L = red[HD / 64];
#pragma unroll
  for (int i = 1; i < HD / 64; ++i) L += red[HD / 64 + i];
  out[(size_t)b * ldo + (size_t)h * HD + d] = f2bf(o / L);
}

}  // namespace

namespace {
int decode_fused_launch(const void* qkv, const float* cos_row, const float* sin_row, void* k_cache, void* v_cache,
                        void* out, float* ws, int64_t ws_floats, int64_t B, int64_t Hq, int64_t Hkv, int64_t head_dim,
                        int64_t ldqkv, int64_t ldo, int64_t cache_rows, int64_t pos, float scale, hipStream_t stream) {
  if (!qkv || !cos_row || !sin_row || !k_cache || !v_cache || !out || !ws) return ZK_EINVAL;
  if (B <= 0 || B > 65535 || Hq <= 0 || Hkv <= 0 || Hkv > 65535 || Hq % Hkv || Hq / Hkv > 16) return ZK_EINVAL;
  if ((head_dim != 64 && head_dim != 128) || pos < 0 || pos >= cache_rows || (ldqkv & 7)) return ZK_EINVAL;
  if (ldqkv < (Hq + 2 * Hkv) * head_dim || ldo < Hq * head_dim) return ZK_EINVAL;
  if ((reinterpret_cast<uintptr_t>(k_cache) | reinterpret_cast<uintptr_t>(v_cache) | reinterpret_cast<uintptr_t>(qkv) |
       reinterpret_cast<uintptr_t>(ws)) & 15)
    return ZK_EINVAL;
  const int64_t nchunks = pos / 64 + 1;
  if (nchunks > 0x7FFFFFFF || ws_floats < B * Hq * nchunks * (head_dim + 2)) return ZK_EINVAL;
  if (cache_rows * Hkv * head_dim >= 0x7FFFFFFFFFFFLL) return ZK_EINVAL;
  const int kvw = (int)(Hkv * head_dim);
  dim3 grid((unsigned)nchunks, (unsigned)Hkv, (unsigned)B);
  if (head_dim == 128) {
    hipLaunchKernelGGL((attn_decode_split_kernel<128>), grid, dim3(256), 0, stream, (const bf16_t*)qkv, (int)ldqkv,
                       cos_row, sin_row, (bf16_t*)k_cache, (bf16_t*)v_cache, (long long)(cache_rows * kvw), kvw, ws,
                       (int)Hq, (int)Hkv, (int)pos, (int)nchunks, scale);
    hipLaunchKernelGGL((attn_decode_merge_kernel<128>), dim3((unsigned)(B * Hq)), dim3(128), 0, stream,
                       (const float*)ws, (bf16_t*)out, (int)ldo, (int)Hq, (int)nchunks);
  } else {
    hipLaunchKernelGGL((attn_decode_split_kernel<64>), grid, dim3(256), 0, stream, (const bf16_t*)qkv, (int)ldqkv,
                       cos_row, sin_row, (bf16_t*)k_cache, (bf16_t*)v_cache, (long long)(cache_rows * kvw), kvw, ws,
                       (int)Hq, (int)Hkv, (int)pos, (int)nchunks, scale);
    hipLaunchKernelGGL((attn_decode_merge_kernel<64>), dim3((unsigned)(B * Hq)), dim3(64), 0, stream,
                       (const float*)ws, (bf16_t*)out, (int)ldo, (int)Hq, (int)nchunks);
  }
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}
}  // namespace

extern "C" int zk_attn_decode_fused(const void* qkv, const float* cos_row, const float* sin_row, void* k_cache,
                                    void* v_cache, void* out, float* ws, int64_t ws_floats, int64_t B, int64_t Hq,
                                    int64_t Hkv, int64_t head_dim, int64_t ldqkv, int64_t ldo, int64_t cache_rows,
                                    int64_t pos, float scale, hipStream_t stream) {
  return decode_fused_launch(qkv, cos_row, sin_row, k_cache, v_cache, out, ws, ws_floats, B, Hq, Hkv, head_dim, ldqkv,
                             ldo, cache_rows, pos, scale, stream);
}

// ---- the same step for callers that cannot allocate (HIP graph capture, serving loops): the fp32 partials
// (chunk max, chunk sum, un-normalised output per 64-key chunk and q head) live in a device array of the code
// object instead of a caller workspace.  `scratch` is a HOST-SIDE STAND-IN here, not a device address: host code
// only ever uses the per-device pointer hipGetSymbolAddress returned (each GPU holds its own copy of the array at
// its own address).  Nothing in it has to start at zero: the partial pass writes every word the merge pass reads.
// Calls that overlap on one GPU (two streams) race on the array.
static __device__ __align__(256) float scratch[ZK_ATTN_DECODE_STATIC_FLOATS];

namespace {
constexpr int kMaxDevices = 64;
std::atomic<float*> g_scratch_ptr[kMaxDevices];     // per device, set once by the first call on that device

int scratch_for_current_device(float** out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return ZK_EINVAL;
  float* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (!p) {
    void* sym = nullptr;
    if (hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch)) != hipSuccess || !sym) return ZK_ELAUNCH;
    p = static_cast<float*>(sym);
    g_scratch_ptr[dev].store(p, std::memory_order_release);   // racing first calls look up the same address
  }
  *out = p;
  return ZK_OK;
}
}  // namespace

extern "C" int zk_attn_decode_fused_static(const void* qkv, const float* cos_row, const float* sin_row, void* k_cache,
                                           void* v_cache, void* out, int64_t B, int64_t Hq, int64_t Hkv,
                                           int64_t head_dim, int64_t ldqkv, int64_t ldo, int64_t cache_rows,
                                           int64_t pos, float scale, hipStream_t stream) {
  // argument errors first (-22), then the size limit (-7): an invalid AND oversized call is an invalid call
  if (!qkv || !cos_row || !sin_row || !k_cache || !v_cache || !out) return ZK_EINVAL;
  if (B <= 0 || B > 65535 || Hq <= 0 || Hkv <= 0 || Hkv > 65535 || Hq % Hkv || Hq / Hkv > 16) return ZK_EINVAL;
  if ((head_dim != 64 && head_dim != 128) || pos < 0 || pos >= cache_rows || (ldqkv & 7)) return ZK_EINVAL;
  if (ldqkv < (Hq + 2 * Hkv) * head_dim || ldo < Hq * head_dim) return ZK_EINVAL;
  const int64_t nchunks = pos / 64 + 1;
  // reject what does not fit BEFORE touching the device (no silent truncation; -7 = argument list too big)
  if (B * Hq > ZK_ATTN_DECODE_STATIC_FLOATS / (head_dim + 2) ||
      nchunks > ZK_ATTN_DECODE_STATIC_FLOATS / ((head_dim + 2) * B * Hq))
    return ZK_E2BIG;
  // the array's address is per device: the CURRENT device must be the one `stream` belongs to (the kernels run
  // there); a stream of another GPU would be handed this GPU's address
  if (stream) {
    hipDevice_t sdev = -1;
    int cur = -1;
    if (hipStreamGetDevice(stream, &sdev) != hipSuccess || hipGetDevice(&cur) != hipSuccess || (int)sdev != cur)
      return ZK_EINVAL;
  }
  float* ws = nullptr;
  const int rc = scratch_for_current_device(&ws);
  if (rc != ZK_OK) return rc;
  return decode_fused_launch(qkv, cos_row, sin_row, k_cache, v_cache, out, ws, ZK_ATTN_DECODE_STATIC_FLOATS, B, Hq, Hkv,
                             head_dim, ldqkv, ldo, cache_rows, pos, scale, stream);
}

extern "C" int zk_attn_decode(const void* q, const void* k_cache, const void* v_cache, void* out, const int* lens,
                              int64_t B, int64_t Hq, int64_t Hkv, int64_t head_dim, int64_t ldq, int64_t ldk,
                              int64_t ldv, int64_t ldo, int64_t batch_stride_k, int64_t batch_stride_v,
                              int64_t max_len, float scale, hipStream_t stream) {
  if (!q || !k_cache || !v_cache || !out || !lens || B <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv) return ZK_EINVAL;
  if ((head_dim != 64 && head_dim != 128) || ((ldk | ldv) & 7) || max_len <= 0 || max_len > 15000) return ZK_EINVAL;   // scores live in <= 64 KiB of LDS
  if ((reinterpret_cast<uintptr_t>(k_cache) | reinterpret_cast<uintptr_t>(v_cache)) & 15) return ZK_EINVAL;
  if ((batch_stride_k | batch_stride_v) & 7) return ZK_EINVAL;
  const size_t smem = (size_t)(5 * head_dim + max_len) * sizeof(float);
  dim3 grid((unsigned)(B * Hq)), block(256);
  if (head_dim == 128)
    hipLaunchKernelGGL((attn_decode_kernel<128>), grid, block, smem, stream, (const bf16_t*)q, (const bf16_t*)k_cache,
                       (const bf16_t*)v_cache, (bf16_t*)out, lens, (int)Hq, (int)Hkv, (int)ldq, (int)ldk, (int)ldv,
                       (int)ldo, (long long)batch_stride_k, (long long)batch_stride_v, scale);
  else
    hipLaunchKernelGGL((attn_decode_kernel<64>), grid, block, smem, stream, (const bf16_t*)q, (const bf16_t*)k_cache,
                       (const bf16_t*)v_cache, (bf16_t*)out, lens, (int)Hq, (int)Hkv, (int)ldq, (int)ldk, (int)ldv,
                       (int)ldo, (long long)batch_stride_k, (long long)batch_stride_v, scale);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

namespace zkattn { int launch_attn_fwd_w64(const AttnArgs& a, int head_dim, int causal, hipStream_t stream); }
namespace {
int attn_fwd_args(AttnArgs& a, const void* q, const void* k, const void* v, void* o, float* lse, int64_t B, int64_t S,
                  int64_t Hq, int64_t Hkv, int64_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                  float scale) {
  if (!q || !k || !v || !o || bad_attn(B, S, Hq, Hkv, head_dim, ldq, ldk, ldv, ldo)) return ZK_EINVAL;
  // K / V rows of one batch element are addressed with 32-bit byte offsets by the LDS-DMA
  if (S * ldk * 2 >= 0x7FFFFFF0LL || S * ldv * 2 >= 0x7FFFFFF0LL) return ZK_EINVAL;
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v)) & 15) return ZK_EINVAL;
  a = AttnArgs{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.out = (bf16_t*)o; a.lse = lse;
  a.B = (int)B; a.S = (int)S; a.Hq = (int)Hq; a.Hkv = (int)Hkv;
  a.ldq = (int)ldq; a.ldk = (int)ldk; a.ldv = (int)ldv; a.ldo = (int)ldo; a.scale = scale;
  return ZK_OK;
}
}  // namespace

extern "C" int zk_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int64_t B, int64_t S,
                           int64_t Hq, int64_t Hkv, int64_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv,
                           int64_t ldo, float scale, int causal, hipStream_t stream) {
  AttnArgs a;
  const int rc = attn_fwd_args(a, q, k, v, o, lse, B, S, Hq, Hkv, head_dim, ldq, ldk, ldv, ldo, scale);
  if (rc != ZK_OK) return rc;
  dim3 grid((unsigned)(((S + 127) / 128) * Hq * B)), block(256);
  if (head_dim == 64) {
    if (causal) hipLaunchKernelGGL((attn_fwd_kernel<64, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<64, false>), grid, block, 0, stream, a);
  } else {
    if (causal) hipLaunchKernelGGL((attn_fwd_kernel<128, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<128, false>), grid, block, 0, stream, a);
  }
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

extern "C" int zk_attn_fwd_w64(const void* q, const void* k, const void* v, void* o, float* lse, int64_t B, int64_t S,
                               int64_t Hq, int64_t Hkv, int64_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv,
                               int64_t ldo, float scale, int causal, hipStream_t stream) {
  AttnArgs a;
  const int rc = attn_fwd_args(a, q, k, v, o, lse, B, S, Hq, Hkv, head_dim, ldq, ldk, ldv, ldo, scale);
  if (rc != ZK_OK) return rc;
  launch_attn_fwd_w64(a, (int)head_dim, causal, stream);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

extern "C" int zk_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                           const float* lse, float* delta, void* dq, void* dk, void* dv, int64_t B, int64_t S,
                           int64_t Hq, int64_t Hkv, int64_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv,
                           int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, float scale, int causal,
                           hipStream_t stream) {
  return zk_attn_bwd_rope(q, k, v, o, d_o, lse, delta, dq, dk, dv, B, S, Hq, Hkv, head_dim, ldq, ldk, ldv, ldo, lddq,
                          lddk, lddv, scale, causal, nullptr, nullptr, stream);
}

extern "C" int zk_attn_bwd_rope(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                const float* lse, float* delta, void* dq, void* dk, void* dv, int64_t B, int64_t S,
                                int64_t Hq, int64_t Hkv, int64_t head_dim, int64_t ldq, int64_t ldk, int64_t ldv,
                                int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, float scale, int causal,
                                const float* cos_t, const float* sin_t, hipStream_t stream) {
  if (!q || !k || !v || !o || !d_o || !lse || !delta || !dq || !dk || !dv) return ZK_EINVAL;
  if ((cos_t == nullptr) != (sin_t == nullptr)) return ZK_EINVAL;
  if ((reinterpret_cast<uintptr_t>(cos_t) | reinterpret_cast<uintptr_t>(sin_t)) & 15) return ZK_EINVAL;
  if (bad_attn(B, S, Hq, Hkv, head_dim, ldq, ldk, ldv, ldo) || ((lddq | lddk | lddv) & 7)) return ZK_EINVAL;
  // the dK/dV kernel addresses one batch element's rows with 32-bit byte offsets
  if (S * ldq * 2 >= 0x7FFFFFF0LL || S * ldo * 2 >= 0x7FFFFFF0LL || S * ldv * 2 >= 0x7FFFFFF0LL || S * ldk * 2 >= 0x7FFFFFF0LL ||
      Hq * S * 4 >= 0x7FFFFFF0LL) return ZK_EINVAL;
  if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(d_o) | reinterpret_cast<uintptr_t>(dk) |
       reinterpret_cast<uintptr_t>(dv) | reinterpret_cast<uintptr_t>(dq)) & 15) return ZK_EINVAL;
  AttnArgs a{};
  a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (const bf16_t*)o;
  a.d_o = (const bf16_t*)d_o; a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
  a.lse_in = lse; a.delta = delta; a.delta_out = delta; a.cos_t = cos_t; a.sin_t = sin_t;
  a.B = (int)B; a.S = (int)S; a.Hq = (int)Hq; a.Hkv = (int)Hkv;
  a.ldq = (int)ldq; a.ldk = (int)ldk; a.ldv = (int)ldv; a.ldo = (int)ldo;
  a.lddq = (int)lddq; a.lddk = (int)lddk; a.lddv = (int)lddv; a.scale = scale;
  dim3 block(256);
  dim3 gq((unsigned)(((S + 127) / 128) * Hq * B));
  dim3 gk((unsigned)(((S + 127) / 128) * Hkv * B)), bk(512);
  if (head_dim == 64) {
    // dK/dV block shape: 8 waves (one block per CU, 256 slots) or 4 waves walking both query halves (two blocks per
    // CU, 512 slots, each twice as long but ~15 % cheaper per unit of work: 721 -> 653 us at B = 64).  Rounds x
    // block time decides: small grids (B = 8: 640 blocks = 3 short rounds against 2 long ones) keep the 8-wave block.
    const double r8 = (double)((gk.x + 255) / 256), r4 = (double)((gk.x + 511) / 512) * 2.0 * 0.85;
    const bool four = r4 < r8;
    if (causal) {
      hipLaunchKernelGGL((attn_bwd_dq_kernel<64, true>), gq, block, 0, stream, a);
      if (four) hipLaunchKernelGGL((attn_bwd_dkdv_kernel<64, true, 1>), gk, dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((attn_bwd_dkdv_kernel<64, true, 2>), gk, bk, 0, stream, a);
    } else {
      hipLaunchKernelGGL((attn_bwd_dq_kernel<64, false>), gq, block, 0, stream, a);
      if (four) hipLaunchKernelGGL((attn_bwd_dkdv_kernel<64, false, 1>), gk, dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((attn_bwd_dkdv_kernel<64, false, 2>), gk, bk, 0, stream, a);
    }
  } else {
    if (causal) {
      hipLaunchKernelGGL((attn_bwd_dq_kernel<128, true>), gq, block, 0, stream, a);
      hipLaunchKernelGGL((attn_bwd_dkdv_kernel<128, true, 2>), gk, bk, 0, stream, a);
    } else {
      hipLaunchKernelGGL((attn_bwd_dq_kernel<128, false>), gq, block, 0, stream, a);
      hipLaunchKernelGGL((attn_bwd_dkdv_kernel<128, false, 2>), gk, bk, 0, stream, a);
    }
  }
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}
