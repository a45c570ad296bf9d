// Native driver of one Llama decoder layer on a single-token decoding step (tf:llama:295-324 with a
// KV cache, as tf generate(use_cache=True) runs it for ref:inference/vlm_infer.py:259-270).
//
// A decoding step is ~12 short kernels per layer x 80 layers: issued one by one from Python the host
// (not the GPU) sets the token rate.  These two entry points enqueue each half of a layer from C —
// the attention half up to the o-projection partial sum, the MLP half up to the down-projection
// partial sum — so that a tensor-parallel host can all-reduce between them (2 per layer), and a
// single-GPU host pays two calls per layer instead of twelve.  Nothing is allocated: the caller owns
// the scratch rows.  All kernels are the library's own (zk_gemv_fused_bf16 with the RMSNorm / SwiGLU
// operand modes, zk_rope, zk_attn_decode) plus the cache append below.
#include "common.h"
#include "../../include/zero_hip.h"

namespace {

// k / v of the new token: packed qkv rows [B][(nh + 2 nkv) hd] -> cache row `pos` of every sequence
__global__ __launch_bounds__(256) void kv_append_kernel(const bf16_t* qkv, bf16_t* kc, bf16_t* vc, int B, int ldqkv,
                                                        int k_off, int kvw, long long rows, int pos) {
  const int per = kvw >> 3;                          // 16-byte chunks per k (or v) row
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * per * 2) return;
  const int which = i / (B * per), r = i % (B * per), b = r / per, c = r % per;
  const u32x4 val = *reinterpret_cast<const u32x4*>(qkv + (size_t)b * ldqkv + k_off + which * kvw + c * 8);
  bf16_t* dst = (which ? vc : kc) + ((size_t)b * rows + pos) * kvw + c * 8;
  *reinterpret_cast<u32x4*>(dst) = val;
}

inline bool bad_layer(const zk_decode_layer_t* L) {
  return !L || L->B <= 0 || L->B > 4 || L->d <= 0 || (L->d & 7) || L->nh <= 0 || L->nkv <= 0 || L->nh % L->nkv ||
         (L->hd != 64 && L->hd != 128) || L->F <= 0 || (L->F & 7) || L->pos < 0 || L->pos >= L->cache_rows ||
         !L->ln1_w || !L->wqkv || !L->wo || !L->ln2_w || !L->wgu || !L->wdown || !L->cos_t || !L->sin_t ||
         !L->k_cache || !L->v_cache || !L->lens || !L->h || !L->qkv || !L->ao || !L->gu || !L->hm;
}

}  // namespace

#define ZK_TRY(call) do { int rc__ = (call); if (rc__ != ZK_OK) return rc__; } while (0)

extern "C" int zk_llama_decode_attn(const zk_decode_layer_t* L, const void* x, void* out, int add_residual,
                                    hipStream_t stream) {
  if (bad_layer(L) || !x || !out) return ZK_EINVAL;
  const int64_t B = L->B, d = L->d, hd = L->hd, qw = L->nh * hd, kvw = L->nkv * hd, W = qw + 2 * kvw;
  // RMSNorm fused into the q|k|v product (d > 16384 would need the separate kernel: not a Llama shape)
  ZK_TRY(zk_gemv_fused_bf16(x, L->wqkv, L->qkv, B, W, d, d, d, W, nullptr, 0, 1, L->ln1_w, L->eps, stream));
  if (L->attn_ws && L->nh / L->nkv <= 16 && L->attn_ws_floats >= B * L->nh * (L->pos / 64 + 1) * (hd + 2)) {
    // RoPE + cache append + split-cache attention + merge: two launches
    ZK_TRY(zk_attn_decode_fused(L->qkv, L->cos_t + L->pos * (hd / 2), L->sin_t + L->pos * (hd / 2), L->k_cache,
                                L->v_cache, L->ao, L->attn_ws, L->attn_ws_floats, B, L->nh, L->nkv, hd, W, qw,
                                L->cache_rows, L->pos, 1.0f / sqrtf((float)hd), stream));
    return zk_gemv_bf16(L->ao, L->wo, out, B, d, qw, qw, qw, d, add_residual ? x : nullptr, d, stream);
  }
  // every row is position `pos`: a one-row table (S = 1) starting at that position
  ZK_TRY(zk_rope(L->qkv, L->cos_t + L->pos * (hd / 2), L->sin_t + L->pos * (hd / 2), B, 1, L->nh + L->nkv, hd, W, 0,
                 stream));
  const int total = (int)(B * (kvw >> 3) * 2);
  hipLaunchKernelGGL(kv_append_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                     (const bf16_t*)L->qkv, (bf16_t*)L->k_cache, (bf16_t*)L->v_cache, (int)B, (int)W, (int)qw, (int)kvw,
                     (long long)L->cache_rows, (int)L->pos);
  ZK_CHECK_LAUNCH();
  ZK_TRY(zk_attn_decode(L->qkv, L->k_cache, L->v_cache, L->ao, L->lens, B, L->nh, L->nkv, hd, W, kvw, kvw, qw,
                        L->cache_rows * kvw, L->cache_rows * kvw, L->pos + 1, 1.0f / sqrtf((float)hd), stream));
  return zk_gemv_bf16(L->ao, L->wo, out, B, d, qw, qw, qw, d, add_residual ? x : nullptr, d, stream);
}

extern "C" int zk_llama_decode_mlp(const zk_decode_layer_t* L, const void* xa, void* out, int add_residual,
                                   hipStream_t stream) {
  if (bad_layer(L) || !xa || !out) return ZK_EINVAL;
  const int64_t B = L->B, d = L->d, F = L->F;
  // RMSNorm fused into the gate|up product, SwiGLU fused into the down product's x operand
  ZK_TRY(zk_gemv_fused_bf16(xa, L->wgu, L->gu, B, 2 * F, d, d, d, 2 * F, nullptr, 0, 1, L->ln2_w, L->eps, stream));
  return zk_gemv_fused_bf16(L->gu, L->wdown, out, B, d, F, 2 * F, F, d, add_residual ? xa : nullptr, d, 2, nullptr, 0.f,
                            stream);
}
