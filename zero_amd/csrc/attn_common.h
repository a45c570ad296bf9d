// Pieces shared by the attention translation units (attention.hip, attention_w64.hip): argument block, fragment
// addressing of the chunk-swizzled LDS tiles, accumulator <-> operand packing.  See attention.hip's header for the
// layout idea.
#pragma once
#include "common.h"

namespace zkattn {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

struct AttnArgs {
  const bf16_t *q, *k, *v, *o, *d_o;
  bf16_t *out, *dq, *dk, *dv;
  float* lse;
  const float* lse_in;
  const float* delta;
  float* delta_out;
  const float *cos_t, *sin_t;    // backward only: when set, dQ and dK leave with the inverse rotary applied
  int B, S, Hq, Hkv;
  int ldq, ldk, ldv, ldo, lddq, lddk, lddv;
  float scale;
};

template <int HD>
struct Cfg {
  static constexpr int CPR = HD / 8;         // 16-byte chunks per row
  static constexpr int DS = HD / 16;         // k-steps over the head dim
  static constexpr int DB = HD / 32;         // 32-wide d blocks
};

__device__ __forceinline__ bf16x8 pack_acc(const f32x16& a, int s) {
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)a[8 * s + j];
  return r;
}
__device__ __forceinline__ bf16x8 gfrag(const bf16_t* p, bool ok) {
  if (!ok) { bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.0f; return z; }
  return *reinterpret_cast<const bf16x8*>(p);
}
// accumulator register r of lane-half h <-> tile row
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- swizzled, unpadded LDS tiles (filled by LDS-DMA) --------------------------------------
template <int HD>
__device__ __forceinline__ int swz_key(int row) {
  if (HD == 128) return ((row & 3) << 2) | ((row >> 2) & 3);
  return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
}
template <int HD>
__device__ __forceinline__ bf16x8 row_frag_sw(const unsigned char* lds, int row0, int ds, int lane) {
  const int row = row0 + (lane & 31), chunk = ds * 2 + (lane >> 5);
  return *reinterpret_cast<const bf16x8*>(lds + row * (HD * 2) + ((chunk ^ swz_key<HD>(row)) << 4));
}
template <int HD>
__device__ __forceinline__ bf16x8 tr_frag_sw(const unsigned char* lds, int k0, int s, int db, int lane) {
  const int h = lane >> 5, i = lane & 15;
  const int row = k0 + 16 * s + 4 * h + (i >> 2);
  const int chunk = db * 4 + 2 * ((lane >> 4) & 1) + ((i & 3) >> 1);
  const int sub = (i & 1) * 8;
  short4_t lo = lds_read_tr16(lds + row * (HD * 2) + ((chunk ^ swz_key<HD>(row)) << 4) + sub);
  short4_t hi = lds_read_tr16(lds + (row + 8) * (HD * 2) + ((chunk ^ swz_key<HD>(row + 8)) << 4) + sub);
  union { struct { short4_t a, b; } s2; bf16x8 v; } u;
  u.s2.a = lo;
  u.s2.b = hi;
  return u.v;
}

}  // namespace zkattn
