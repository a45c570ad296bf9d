// Shared device helpers for the gfx950 (CDNA4) kernels of zero_amd.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

typedef uint16_t bf16_t;  // raw bf16 bits in HBM

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define ZK_OK 0
#define ZK_EINVAL (-22)
#define ZK_ELAUNCH (-5)
#define ZK_E2BIG (-7)

#define ZK_CHECK_LAUNCH()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return ZK_ELAUNCH;               \
  } while (0)

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ float bf2f(bf16_t v) {
  return __uint_as_float(((unsigned int)v) << 16);
}
// round-to-nearest-even through the hardware convert (keeps NaN a NaN)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&b);
}
__device__ __forceinline__ unsigned int pack2bf(float lo, float hi) {
  bf16x2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return *reinterpret_cast<unsigned int*>(&v);
}
// 1 / (1 + e^-x) through the hardware reciprocal (1 ulp) instead of an IEEE division (v_div_scale / v_rcp / four fmas /
// v_div_fmas / v_div_fixup: 10 vector instructions of the ~23 per element that the SwiGLU-backward epilogue spent).
// Every sigmoid of the library goes through here, so fused epilogues and stand-alone kernels stay bit-identical.
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_fast(float x) { return x * sigmoid_fast(x); }
__device__ __forceinline__ float lo_bf(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }

// one rotary pair (x1 at d, x2 at d + hd/2).  Both products are rounded before the add (no fma
// contraction), so every kernel that rotates produces the same bits as zk_rope.
__device__ __forceinline__ void rope_pair(float x1, float x2, float cs, float sn, float& o1, float& o2) {
#pragma clang fp contract(off)
  o1 = x1 * cs - x2 * sn;
  o2 = x2 * cs + x1 * sn;
}
// rotate one 16-byte chunk of a row: `own` is this lane's chunk, `other` its partner hd/2 away; cs are the 8
// cosines of the chunk's d range and ss its SIGNED sines: -sin where own is the x1 half, +sin where it is the
// x2 half (both flipped for the inverse rotation).  own*cos + other*(+-sin) with both products rounded is
// rope_pair's o1 / o2 bit for bit (x - y*s == x + y*(-s) in IEEE arithmetic).
__device__ __forceinline__ u32x4 rope_chunk(const u32x4& own, const u32x4& other, const float* cs, const float* ss) {
#pragma clang fp contract(off)
  u32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    r[j] = pack2bf(lo_bf(own[j]) * cs[2 * j] + lo_bf(other[j]) * ss[2 * j],
                   hi_bf(own[j]) * cs[2 * j + 1] + hi_bf(other[j]) * ss[2 * j + 1]);
  return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// 16-byte global load through a buffer descriptor; out-of-range offsets return 0.
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned int byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// LDS-DMA (buffer_load ... lds), as the builtin or (ASM) as INLINE ASM.  The compiler's waitcnt pass cannot tell the
// DMA's destination from the source of a later ds_read_b64_tr_b16, so for the builtin it puts s_waitcnt vmcnt(0) in front
// of the first transposed read after every DMA: the tile requested at the top of an iteration has to LAND before the
// second half of the same iteration starts, whatever the ring depth.  With two workgroups per CU that wait is hidden
// (zk_attn_fwd / bwd: no difference between the two forms, profiles/r03_attn_dma_asm_ab.txt, so they keep the builtin);
// at one wave per SIMD it is not (zk_attn_fwd_w64: +7 % at S = 6144, +30 % at head_dim 64 with the asm form).  A kernel
// that uses the asm form orders DMA against reads by hand (counted s_waitcnt vmcnt + barrier) and must have waited for
// every register it loaded in its prologue BEFORE its main loop (a compiler-placed vmcnt(0) at a first use inside the
// loop would drain the DMA the compiler does not know about).  m0 holds the wave-uniform LDS destination; nothing else
// in these kernels uses m0 (gfx9 DS instructions do not).
template <int BYTES, bool ASM = false>
__device__ __forceinline__ void lds_dma(__amdgpu_buffer_rsrc_t r, const void* lds_dst, unsigned voff) {
  static_assert(BYTES == 16 || BYTES == 4, "dwordx4 or dword");
  if constexpr (!ASM) {
    auto* d = (__attribute__((address_space(3))) void*)const_cast<void*>(lds_dst);
    if constexpr (BYTES == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, d, 16, voff, 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, d, 4, voff, 0, 0, 0);
  } else {
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) void*)const_cast<void*>(lds_dst));
    if constexpr (BYTES == 16)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m), "v"(voff), "s"(r) : "memory");
    else
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(m), "v"(voff), "s"(r) : "memory");
  }
}

// LDS transposed read: per 16-lane group a 4(row) x 16(col) block of 16-bit elements; the lane
// 4q+p supplies the address of row q, cols 4p..4p+3; lane i receives column i, rows 0..3.
__device__ __forceinline__ short4_t lds_read_tr16(const void* lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) short4_t*)(lds_addr));
}
