// Flash attention forward, "64 query rows per wave" structure for gfx950 (CDNA4).
//
// Same math, operand layouts and LDS images as attn_fwd_kernel (attention.hip); what changes is who does what:
//   * a workgroup = 4 waves = ONE wave per SIMD = a 256-row query block of one (batch, head); each wave owns 64 rows
//     as two 32-row halves (A, B) and the SIMD's whole register file (O^T of both halves: 2 x hd fp32 registers);
//   * every K / V fragment read from LDS feeds TWO MFMAs (one per half): half the LDS reads per flop of the
//     32-rows-per-wave kernel, and two independent accumulator chains per wave;
//   * the tile loop is software-pipelined inside the wave: while the matrix pipe runs S(t+1) = K(t+1) Q^T (32 MFMAs),
//     the vector pipe packs P(t) to bf16 and rescales O by alpha(t); while it runs O += V(t)^T P(t) (32 MFMAs), the
//     vector pipe does the online-softmax step of S(t+1) (mask, row max, exp2, row sum).  The in-order wave therefore
//     always has independent matrix and vector work in one basic block; sched_group_barrier pins the interleave.
//   * the O rescale is conditional (the max-growth threshold of FlashAttention-4 / the guide's T13): a row keeps its
//     reference maximum m until a tile raises the row maximum by more than 2^THR; then (rare: never on the first
//     tile, whose O is zero) the wave rescales O and l by alpha = 2^(m_old - m_new).  P = 2^(s - m) is then bounded by
//     2^THR instead of 1 — bf16's relative precision does not depend on the scale, l and O are fp32, and O / l is
//     independent of the reference — so the result differs from the always-rescale form only by rounding.  O lives
//     in the accumulator half of the register file, which the vector ALU cannot address: an unconditional rescale
//     would cost three vector instructions per element (read, multiply, write back).
//   * K runs one tile ahead of V: two 3-stage LDS-DMA rings (K ring, V ring), one barrier per tile.
// Causal blocks: wave w of the block needs 4 qb + w + 1 key tiles (its rows are 64-aligned, so both halves share
// the count and only the wave's LAST tile is masked); waves that have finished keep issuing their share of the DMA.
#include "attn_common.h"

using namespace zkattn;

namespace {

#ifndef ZK_W64_PF
#define ZK_W64_PF 6
#endif

// one 64-row operand tile by LDS-DMA into the chunk-swizzled image row_frag_sw / tr_frag_sw read
template <int HD>
__device__ __forceinline__ void dma_tile(__amdgpu_buffer_rsrc_t r, unsigned char* dst_tile, int row0, int S, int ld,
                                         int col0, int tid, int wv, bool valid) {
  constexpr int CPR = HD / 8, NQ = 64 * CPR / 256;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int sidx = tid + 256 * i;
    const int row = sidx / CPR, ch = (sidx % CPR) ^ swz_key<HD>(row);
    const bool ok = valid && row0 + row < S;
    unsigned char* dst = dst_tile + (wv * 64 + 256 * i) * 16;   // wave-uniform; the hardware adds lane * 16
    lds_dma<16, true>(r, dst,
                                               ok ? (unsigned)(((row0 + row) * ld + col0 + ch * 8) * 2) : 0xFFFFFFF0u);
  }
}

// An empty statement the values pass through: whatever is computed from them afterwards cannot be moved above this
// point by the compiler's middle end (which otherwise hoists the whole softmax step out of its MFMA slots to where
// its operands are defined), and the operands sit in vector registers here.
#define ZK_PIN2X2(V0, V1, I0, I1)                                                  \
  do {                                                                             \
    float a_ = (V0)[I0], b_ = (V0)[I1], c_ = (V1)[I0], d_ = (V1)[I1];              \
    asm volatile("" : "+v"(a_), "+v"(b_), "+v"(c_), "+v"(d_));                     \
    (V0)[I0] = a_; (V0)[I1] = b_; (V1)[I0] = c_; (V1)[I1] = d_;                    \
  } while (0)
__device__ __forceinline__ void pin16(f32x16& v) {
  asm volatile("" : "+v"(v));
}

__device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));   // one instruction, no canonicalising v_max
  return r;
}

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, 1) void attn_fwd_w64_kernel(AttnArgs a) {
  using C = Cfg<HD>;
  constexpr int KV = 64, RB = HD * 2, TILE = KV * RB, NST = 4;
  constexpr int NQ = KV * C::CPR / 256, PER = 2 * NQ;
  constexpr int DS = C::DS, DB = C::DB;
  constexpr int NF = 2 * DB;                 // operand fragments (= double MFMA slots) per block; DS == 2 * DB
  static_assert(DS == NF, "QK and PV blocks have the same number of fragments");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * NST * TILE];
  unsigned char* const kring = smem;
  unsigned char* const vring = smem + NST * TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int nbh = a.Hq * a.B;   // 1-D grid, heaviest (causal: latest) query blocks first
  const int qblk = (int)(gridDim.x / nbh) - 1 - (int)(blockIdx.x / nbh);
  const int head = (blockIdx.x % nbh) % a.Hq, b = (blockIdx.x % nbh) / a.Hq;
  const int kvh = head / (a.Hq / a.Hkv);
  const int q0w = qblk * 256 + wv * 64;
  const int qi[2] = {q0w + (lane & 31), q0w + 32 + (lane & 31)};
  const float sc = a.scale * LOG2E;

  __amdgpu_buffer_rsrc_t rk = make_rsrc(a.k + (size_t)b * a.S * a.ldk, 0x7FFFFFF0u);
  __amdgpu_buffer_rsrc_t rv = make_rsrc(a.v + (size_t)b * a.S * a.ldv, 0x7FFFFFF0u);
  int kv_end = a.S;
  if (CAUSAL) kv_end = min(a.S, qblk * 256 + 256);
  const int nt = (kv_end + KV - 1) / KV;
  int nt_w = nt;
  if (CAUSAL) nt_w = min(nt, q0w / KV + 1);
  if (q0w >= a.S) nt_w = 0;                       // a wave with no rows only moves tiles

  // Q fragments first: every later counted vmcnt then also covers them
  bf16x8 qf[2][DS];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const bf16_t* qp = a.q + ((size_t)b * a.S + qi[x]) * a.ldq + head * HD + 8 * h;
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) qf[x][ds] = gfrag(qp + ds * 16, qi[x] < a.S);
  }
  // Q is only ever an MFMA operand: park it in the accumulator half of the register file (MFMA operands may come
  // from there), which leaves the vector half to S, P and the fragments — the values the vector ALU works on
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) asm volatile("" : "+a"(qf[x][ds]));
  // batch j = {K(j+1), V(j)} = what iteration j needs beyond iteration j-1; K(0) goes ahead of batch 0.
  // One DMA instruction (piece i of PER) at a time, so the pieces can sit between MFMAs.
  auto issue_piece = [&](int j, int i) __attribute__((always_inline)) {
    const bool isk = i < NQ;
    const int ii = isk ? i : i - NQ;
    const int tile = isk ? j + 1 : j;
    const int sidx = tid + 256 * ii;
    const int row = sidx / C::CPR, ch = (sidx % C::CPR) ^ swz_key<HD>(row);
    // branch-free (a divergent `ok ? offset : out-of-range` becomes control flow, and every basic-block boundary in
    // the step lets the compiler sink the softmax slices out of their MFMA slots): out-of-range rows get an offset
    // past the descriptor's end by arithmetic, and arrive as zeros
    const int live = (tile < nt ? a.S : 0) - tile * KV;             // rows of this tile that exist (<= 0: none)
    const unsigned oob = (unsigned)((live - 1 - row) >> 31);        // all ones when row >= live
    unsigned char* dst = (isk ? kring : vring) + (tile % NST) * TILE + (wv * 64 + 256 * ii) * 16;
    const int ld = isk ? a.ldk : a.ldv;
    const unsigned off = (unsigned)(((tile * KV + row) * ld + kvh * HD + ch * 8) * 2);
    lds_dma<16, true>(isk ? rk : rv, dst,
                                               (off & ~oob) | (0xFFFFFFF0u & oob));
  };
  auto issue_batch = [&](int j) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < PER; ++i) issue_piece(j, i);
  };
  dma_tile<HD>(rk, kring, 0, a.S, a.ldk, kvh * HD, tid, wv, true);
  issue_batch(0);
  issue_batch(1);     // (tiles past the end are zero fills of free slots: every wait below can then count the same way)

  f32x16 o[2][DB], s[2][2];
  bf16x8 pf[2][2][2];
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f}, alpha[2] = {1.f, 1.f};
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[x][d][r] = 0.f;

  // ---- the matrix side: one fragment read from LDS feeds two MFMAs (one per 32-row half)
  auto read_k = [&](int t, int kb, int ds) __attribute__((always_inline)) { return row_frag_sw<HD>(kring + (t % NST) * TILE, kb * 32, ds, lane); };
  auto read_v = [&](int t, int kb, int f) __attribute__((always_inline)) { return tr_frag_sw<HD>(vring + (t % NST) * TILE, kb * 32, f / DB, f % DB, lane); };
  auto mma_qk = [&](const bf16x8& kf, int kb, int ds) __attribute__((always_inline)) {
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      f32x16 acc = s[x][kb];
      if (ds == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      }
      s[x][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[x][ds], acc, 0, 0, 0);
    }
  };
  auto mma_pv = [&](const bf16x8& vf, int kb, int f) __attribute__((always_inline)) {
#pragma unroll
    for (int x = 0; x < 2; ++x)
      o[x][f % DB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[x][kb][f / DB], o[x][f % DB], 0, 0, 0);
  };

  // ---- the vector side: the online-softmax step of one 32-key sub-tile (both halves), cut into 16 slices that
  // are placed one (hd 128) or two (hd 64) per fragment slot of the two MFMA blocks it runs beside.
  constexpr float THR = 8.0f;
  float mx[2], m_use[2], rs[2];
  auto sm_slice = [&](int t, auto kb_c, auto mask_c, auto first_c, auto i_c) __attribute__((always_inline)) {
    constexpr int KB = decltype(kb_c)::value, I = decltype(i_c)::value;
    constexpr bool MASK = decltype(mask_c)::value, FIRST = decltype(first_c)::value;
    if constexpr (I <= 1) {                 // mask + running maximum of half I
      constexpr int x = I;
      pin16(s[x][KB]);
      if (MASK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * KV + KB * 32 + acc_row(r, h);
          if (key >= a.S || (CAUSAL && key > qi[x])) s[x][KB][r] = -INFINITY;
        }
      }
      float m0 = max3(s[x][KB][0], s[x][KB][1], s[x][KB][2]);
      float m1 = max3(s[x][KB][3], s[x][KB][4], s[x][KB][5]);
      m0 = max3(m0, s[x][KB][6], s[x][KB][7]);
      m1 = max3(m1, s[x][KB][8], s[x][KB][9]);
      m0 = max3(m0, s[x][KB][10], s[x][KB][11]);
      m1 = max3(m1, s[x][KB][12], s[x][KB][13]);
      m0 = max3(m0, s[x][KB][14], s[x][KB][15]);
      mx[x] = fmaxf(m0, m1);
    } else if constexpr (I == 2) {          // the row's other 16 keys sit in the other lane half; new reference?
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const float other = __shfl_xor(mx[x], 32, 64);
        const float cand = fmaxf(m_run[x], fmaxf(mx[x], other) * sc);
        if (FIRST) {
          alpha[x] = 1.0f;                  // O and l are zero: nothing to rescale
          m_run[x] = cand;
        } else {
          const bool big = cand - m_run[x] > THR;
          alpha[x] = big ? fast_exp2(m_run[x] - cand) : 1.0f;
          m_run[x] = big ? cand : m_run[x];
        }
        m_use[x] = (m_run[x] == -INFINITY) ? 0.f : m_run[x];   // every key masked so far: p = 0, not NaN
        rs[x] = 0.f;
      }
    } else if constexpr (I <= 10) {         // p = exp2(s * sc - m): two elements of each half
      constexpr int g = I - 3;
      ZK_PIN2X2(s[0][KB], s[1][KB], 2 * g, 2 * g + 1);
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int e = 2 * g; e < 2 * g + 2; ++e) s[x][KB][e] = fast_exp2(__builtin_fmaf(s[x][KB][e], sc, -m_use[x]));
    } else if constexpr (I <= 14) {         // row sums of four elements, bf16 packs of one (half, k-step) pair
      constexpr int g = I - 11;
      ZK_PIN2X2(s[0][KB], s[1][KB], 4 * g, 4 * g + 1);
      ZK_PIN2X2(s[0][KB], s[1][KB], 4 * g + 2, 4 * g + 3);
#pragma unroll
      for (int x = 0; x < 2; ++x) rs[x] += (s[x][KB][4 * g] + s[x][KB][4 * g + 1]) + (s[x][KB][4 * g + 2] + s[x][KB][4 * g + 3]);
      pf[g >> 1][KB][g & 1] = pack_acc(s[g >> 1][KB], g & 1);
    } else {
#pragma unroll
      for (int x = 0; x < 2; ++x) l_run[x] = l_run[x] * alpha[x] + rs[x];
    }
  };
  auto sm_slot = [&](int t, auto kb_c, auto mask_c, auto first_c, auto j_c) __attribute__((always_inline)) {    // slot j of 2 * NF
    constexpr int J = decltype(j_c)::value;
    if constexpr (NF == 8) {
      sm_slice(t, kb_c, mask_c, first_c, std::integral_constant<int, J>{});
    } else {
      sm_slice(t, kb_c, mask_c, first_c, std::integral_constant<int, 2 * J>{});
      sm_slice(t, kb_c, mask_c, first_c, std::integral_constant<int, 2 * J + 1>{});
    }
  };
  // rare: the reference maximum of some row of this wave moved by more than 2^THR
  auto maybe_rescale = [&]() __attribute__((always_inline)) {
    if (__builtin_amdgcn_ballot_w64(alpha[0] != 1.0f || alpha[1] != 1.0f) != 0) {
      // O lives in accumulator registers on every path (an "a" operand here, the MFMA's accumulator elsewhere): the
      // allocator then has nothing to copy at the merge.  The statements are opaque to hipcc's hazard recogniser:
      // the nops let the last PV MFMAs land before the reads and the writes settle before the next MFMA reads them.
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float e = o[x][d][r], tmp;
            asm volatile("v_accvgpr_read_b32 %0, %1\n\tv_mul_f32 %0, %0, %2\n\tv_accvgpr_write_b32 %1, %0"
                         : "=&v"(tmp), "+a"(e) : "v"(alpha[x]));
            o[x][d][r] = e;
          }
      asm volatile("s_nop 7" ::: "memory");
    }
  };

  // ---- one tile step of this wave.  Blocks (NF fragment slots = 2 NF MFMAs each), with the vector work beside them:
  //   B1  O += V(t-1)^T P(t-1), keys 32..63   ||  softmax of S(t) keys 0..31, first half  (+ the DMA of batch t+2)
  //   B2  S(t) keys 32..63 = K(t) Q^T         ||  softmax of S(t) keys 0..31, second half
  //   B3  O += V(t)^T P(t), keys 0..31        ||  softmax of S(t) keys 32..63, first half
  //   B4  S(t+1) keys 0..31 = K(t+1) Q^T      ||  softmax of S(t) keys 32..63, second half
  // Fragments are read from LDS PF slots ahead of their MFMAs, across block and step boundaries.
  constexpr int PF = ZK_W64_PF, NBUF = 8;
  bf16x8 fr[NBUF];
  auto frag_read = [&](int t, auto n_c, auto first_c, auto last_c) __attribute__((always_inline)) {   // fragment n of the step's 4 NF (may be >= 4 NF: next step)
    constexpr int N = decltype(n_c)::value;
    constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
    constexpr int blk = (N / NF) % 4, f = N % NF, nxt = N / (4 * NF);
    if constexpr (blk == 0) { if (!(FIRST && nxt == 0) && !(LAST && nxt == 1)) fr[N % NBUF] = read_v(t + nxt - 1, 1, f); }
    else if constexpr (blk == 1) { if (!(LAST && nxt == 1)) fr[N % NBUF] = read_k(t + nxt, 1, f); }
    else if constexpr (blk == 2) { if (!(LAST && nxt == 1)) fr[N % NBUF] = read_v(t + nxt, 0, f); }
    else { if (!LAST) fr[N % NBUF] = read_k(t + nxt + 1, 0, f); }
  };
  static_assert(NF >= PER, "one DMA piece per fragment slot of block B1");
  auto step = [&](int t, auto first_c, auto last_c) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
    using MaskC = std::integral_constant<bool, LAST>;     // only the wave's last tile can need a mask
    if constexpr (FIRST) {                                // nothing was prefetched by a previous step
      static_for<PF>([&](auto n_c) { frag_read(t, n_c, first_c, last_c); });
    }
    static_for<4 * NF>([&](auto n_c) {
      constexpr int N = decltype(n_c)::value;
      constexpr int blk = N / NF, f = N % NF;
      frag_read(t, std::integral_constant<int, N + PF>{}, first_c, last_c);
      if constexpr (blk == 0) {
        if (!FIRST) mma_pv(fr[N % NBUF], 1, f);
        if constexpr (f < PER) issue_piece(t + 2, f);
        sm_slot(t, std::integral_constant<int, 0>{}, MaskC{}, first_c, std::integral_constant<int, f>{});
      } else if constexpr (blk == 1) {
        mma_qk(fr[N % NBUF], 1, f);
        sm_slot(t, std::integral_constant<int, 0>{}, MaskC{}, first_c, std::integral_constant<int, NF + f>{});
      } else if constexpr (blk == 2) {
        if constexpr (f == 0) maybe_rescale();
        mma_pv(fr[N % NBUF], 0, f);
        sm_slot(t, std::integral_constant<int, 1>{}, MaskC{}, std::false_type{}, std::integral_constant<int, f>{});
      } else {
        if (!LAST) mma_qk(fr[N % NBUF], 0, f);
        sm_slot(t, std::integral_constant<int, 1>{}, MaskC{}, std::false_type{}, std::integral_constant<int, NF + f>{});
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    maybe_rescale();
  };

  // wait for batch t = {K(t+1), V(t)} and meet: every wave is then done with K(t-1) and V(t-2), whose slots batch
  // t+2 refills (the pieces go out between the MFMAs of block B1, or in one burst for a wave that has finished)
  auto sync = [&](int t) __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");      // batch t+1 (real or zero fill) may stay in flight
    __builtin_amdgcn_s_barrier();
  };

  // ---- prologue: S(0) keys 0..31
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
  __builtin_amdgcn_s_barrier();
  int t = 0;
  if (nt_w > 0) {
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) mma_qk(read_k(0, 0, ds), 0, ds);
    sync(0);
    if (nt_w == 1) {
      step(0, std::true_type{}, std::true_type{});
    } else {
      step(0, std::true_type{}, std::false_type{});
      for (t = 1; t < nt_w - 1; ++t) {      // the steady state: straight-line code between two barriers
        sync(t);
        step(t, std::false_type{}, std::false_type{});
      }
      sync(t);
      step(t, std::false_type{}, std::true_type{});
    }
    // the last tile's keys 32..63
    const int tl = nt_w - 1;
#pragma unroll
    for (int f = 0; f < NF; ++f) mma_pv(read_v(tl, 1, f), 1, f);
    t = nt_w;
  }
  for (; t < nt; ++t) {                     // a wave that has finished keeps moving its share of the tiles
    sync(t);
    issue_batch(t + 2);
  }

#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const float l_tot = l_run[x] + __shfl_xor(l_run[x], 32, 64);
    const float inv = (l_tot > 0.f) ? 1.0f / l_tot : 0.f;
    if (qi[x] < a.S) {
      bf16_t* op = a.out + ((size_t)b * a.S + qi[x]) * a.ldo + head * HD;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          u32x2 w;
          w[0] = pack2bf(o[x][d][4 * r4] * inv, o[x][d][4 * r4 + 1] * inv);
          w[1] = pack2bf(o[x][d][4 * r4 + 2] * inv, o[x][d][4 * r4 + 3] * inv);
          *reinterpret_cast<u32x2*>(op + d * 32 + 8 * r4 + 4 * h) = w;
        }
      if (h == 0 && a.lse) a.lse[((size_t)b * a.Hq + head) * a.S + qi[x]] = m_run[x] * LN2 + logf(l_tot);
    }
  }
}

}  // namespace

namespace zkattn {
// grid: one workgroup per (256-row query block, head, batch element)
int launch_attn_fwd_w64(const AttnArgs& a, int head_dim, int causal, hipStream_t stream) {
  dim3 grid((unsigned)(((a.S + 255) / 256) * a.Hq * a.B)), block(256);
  if (head_dim == 64) {
    if (causal) hipLaunchKernelGGL((attn_fwd_w64_kernel<64, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((attn_fwd_w64_kernel<64, false>), grid, block, 0, stream, a);
  } else {
    if (causal) hipLaunchKernelGGL((attn_fwd_w64_kernel<128, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((attn_fwd_w64_kernel<128, false>), grid, block, 0, stream, a);
  }
  return 0;
}
}  // namespace zkattn
