/* zero_hip.h — C ABI of libzero_hip.so (gfx950 / MI355X).
 *
 * The reference (facebookresearch/zero) has no FFI of its own for this path: every
 * kernel on its hot path is dispatched by PyTorch/ATen from inside the transformers
 * model classes.  The entry points below are therefore the ATen
 * operator calls those classes make, one `extern "C"` function per kernel family; each
 * comment names the reference/transformers call site it replaces ("tf:" = the
 * transformers package the reference imports, "ref:" = /root/reference/src).
 *
 * Conventions (SURVEY.md §8b "Lower"):
 *   - plain device pointers + int64 sizes, no torch types;
 *   - bf16 tensors are raw uint16 bits, row-major, leading dimensions in ELEMENTS;
 *   - every call is asynchronous on `stream`, allocates nothing, never synchronises,
 *     never changes the current device; the caller owns every buffer;
 *   - returns 0 on success, a negative errno-style code otherwise
 *     (-22 invalid argument / unsupported shape, -5 launch failure, -7 beyond a stated size limit);
 *   - 16-byte aligned base pointers; feature dimensions are multiples of 8.
 */
#ifndef ZERO_HIP_H_
#define ZERO_HIP_H_

#include <stdint.h>
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZK_ABI_VERSION 6
int zk_abi_version(void);

/* ---- GEMM -------------------------------------------------------------------------
 * C[m][n] = epi( alpha * sum_k A(m,k) B(n,k) ), fp32 accumulate on MFMA, bf16 in/out.
 *   transA=0: A [M][lda] (k contiguous)   transA=1: A [K][lda] (m contiguous)
 *   transB=0: B [N][ldb] (k contiguous)   transB=1: B [K][ldb] (n contiguous)
 * Replaces F.linear in tf:models/clip/modeling_clip.py:309-311,333,347-349 (q/k/v/out,
 * fc1/fc2, biased), the patch conv tf:clip:209 (as im2col GEMM), ref:models/adapter.py:95
 * (adapter Linear->GELU->Linear), tf:models/llama/modeling_llama.py:175 (gate/up/down),
 * :254-256,280 (q/k/v/o), :480 (lm_head), and their autograd dgrad (transB=1) / wgrad
 * (transA=1, transB=1) products triggered by ref:train.py:567.
 * Epilogues: bias add; QUICK_GELU (tf:activations.py:117-123) / GELU_ERF (nn.GELU,
 * ref:adapter.py:36) with optional copy of the bf16 pre-activation to aux_out;
 * residual add (tf:clip:376,381; tf:llama:306,321); MUL_D*: multiply by the activation
 * derivative at aux_in (backward of the fused activation); accumulate: C += result.
 * workspace (optional, fp32 scratch owned by the caller): lets plain products with a small
 * output and a long contraction (weight gradients, lm_head dgrad) run split-K; every split
 * writes its own slab (no atomics: deterministic), a finishing pass sums them.
 */
enum {
  ZK_EPI_NONE = 0,
  ZK_EPI_QUICK_GELU = 1,
  ZK_EPI_GELU_ERF = 2,
  ZK_EPI_MUL_DQUICK_GELU = 3,
  ZK_EPI_MUL_DGELU_ERF = 4,
  /* SwiGLU fused into the two GEMMs around it (tf:llama:174-176, frozen surrogate MLP).  The gate|up
   * product is kept in a BLOCK-INTERLEAVED column order: columns [64j, 64j+32) are gate features
   * 32j..32j+31, columns [64j+32, 64j+64) the matching up features (the caller interleaves the
   * weight rows once; any K order is fine for the dgrad that follows as long as both operands agree).
   *   ZK_EPI_SWIGLU     : C[M][N = 2F] = interleaved gate|up (saved for backward), and
   *                       aux_out[M][F] (row stride ldaux) = bf16(silu(gate)) * up.  C may be NULL (this
   *                       epilogue only): the pre-activations are then not written (no-grad forward).
   *   ZK_EPI_SWIGLU_BWD : the GEMM computes dh[M][N = F]; C[M][2F] (interleaved, row stride ldc) gets
   *                       dgate = dh * up * silu'(gate), dup = dh * silu(gate); aux_in[M][2F] = the
   *                       interleaved gate|up of the forward (row stride ldaux).
   * Both: alpha = 1, no bias / residual / accumulate. */
  ZK_EPI_SWIGLU = 5,
  ZK_EPI_SWIGLU_BWD = 6
};
int zk_gemm_bf16(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K,
                 int64_t lda, int64_t ldb, int64_t ldc, int transA, int transB, int epilogue,
                 const void* bias, const void* residual, int64_t ldr, void* aux_out,
                 const void* aux_in, int64_t ldaux, float alpha, const float* alpha_dev,
                 int accumulate, void* workspace, int64_t workspace_bytes, hipStream_t stream);

/* Skinny product for decoding: y[M][N] = x[M][K] W[N][K]^T (+ residual[M][N]), M <= 4 rows, fp32
 * accumulate.  The same F.linear call sites (tf:llama:174-176,230-241,280,480) as they occur in the
 * single-token steps of greedy decoding (ref:inference/vlm_infer.py:259-270): HBM-bound, every weight
 * byte is read once per token. */
int zk_gemv_bf16(const void* x, const void* W, void* y, int64_t M, int64_t N, int64_t K, int64_t ldx,
                 int64_t ldw, int64_t ldy, const void* residual, int64_t ldr, hipStream_t stream);
/* The same with the x operand produced on the fly: xmode 1 = RMSNorm of x with weight
 * ln_w / eps (tf:llama:62-67, same roundings as zk_rmsnorm_fwd), xmode 2 = SwiGLU of a packed
 * gate|up row (x[k] = bf16(bf16(silu(xin[k])) * xin[K + k]), tf:llama:175; ldx >= 2K), 0 = plain. */
int zk_gemv_fused_bf16(const void* x, const void* W, void* y, int64_t M, int64_t N, int64_t K,
                       int64_t ldx, int64_t ldw, int64_t ldy, const void* residual, int64_t ldr,
                       int xmode, const void* ln_w, float eps, hipStream_t stream);

/* The same product under an explicit kernel plan (no counterpart in the reference: tests exercise every kernel
 * variant through it, benchmarks A/B them).  The library keeps no mutable tuning state: zk_gemm_bf16 is this call
 * with plan = NULL (built-in cost model).  Fields, -1 / 0 = built-in choice unless stated:
 *   tile        -1 cost model, 0 = 128x128, 1 = 256x128, 2 = 256x256, 3 = 128x96, 4 = 384x192 (NT ring only),
 *               5 = 192x256 (persistent NT ring; K a multiple of 64)
 *   splits      0 heuristic, 1 never split K, 2..16 force that many splits (plain products with a workspace)
 *   stage_mode  -1 default (0), 0 LDS-DMA for every operand, 1 registers for transposed operands
 *   ring        -1 default (1), 0 / 1: 3-stage ring kernel for NT products on the 256x128 tile off / on
 *   lean        -1 default (1), 0 / 1: lean DMA addressing off / on
 *   pring       -1 default (1), 0 / 1: persistent slot ring off / on (on: one workgroup per CU walks its tiles)
 *   band        0 default, 1..16 m-tiles per band of the persistent kernel's tile order
 *   phase_delay -1 built-in policy, else the persistent kernel's start stagger in 10 ns ticks per phase step
 *   ablation    0; timing-only modes with WRONG results (1 no MFMA, 2 no DMA, 3 no epilogue) are accepted only by
 *               a library built with -DZK_ABLATION (tools/ring_modes.py), -22 otherwise. */
typedef struct {
  int tile, splits, stage_mode, ring, lean, pring, band, phase_delay, ablation;
} zk_gemm_plan_t;
int zk_gemm_bf16_plan(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                      int64_t ldb, int64_t ldc, int transA, int transB, int epilogue, const void* bias,
                      const void* residual, int64_t ldr, void* aux_out, const void* aux_in, int64_t ldaux,
                      float alpha, const float* alpha_dev, int accumulate, void* workspace,
                      int64_t workspace_bytes, const zk_gemm_plan_t* plan, hipStream_t stream);

/* Measurement hook (no counterpart in the reference): which kernel variant this thread's last
 * zk_gemm_bf16 call launched, so a benchmark can attribute its event timings to the kernel symbols a
 * profiler reports.  kind * 1000 + tile * 100 + lean * 10 + epilogue class; kind 0 = two-stage
 * gemm_kernel, 1 = gemm_ring_kernel, 2 = gemm_ringx_kernel, 3 = gemm_pring_kernel (persistent),
 * 4 = split-K (4000 + splits); tile 0..5 = 128x128, 256x128, 256x256, 128x96, 384x192, 192x256; epilogue
 * class of the persistent kernel: 0 plain / bias, 6 the same + residual, 1 quick-GELU pair, 2 SwiGLU forward,
 * 5 SwiGLU backward, 4 erf-GELU pair. */
int zk_gemm_last_plan(void);

/* The kernel choice of zk_gemm_bf16_plan for a call of this shape WITHOUT launching anything (no counterpart in the
 * reference; host logic only, callable without a GPU: tile quantisation model, split-K model, row split, ring
 * operand limits, argument validation).  has_* say which optional operands the call would pass (alignment is taken
 * as given).  Returns the zk_gemm_last_plan code the call would leave (for a row-split product: the code of its
 * whole-round part), or the negative error the call would return. */
int zk_gemm_plan_query(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int transA,
                       int transB, int epilogue, int has_bias, int has_residual, int64_t ldr, int has_aux_out,
                       int has_aux_in, int64_t ldaux, int accumulate, int64_t workspace_bytes,
                       const zk_gemm_plan_t* plan);

/* ---- fused attention ----------------------------------------------------------------
 * softmax(Q K^T * scale [+ causal mask]) V, flash-style (no S x S matrix in HBM).
 * Replaces F.scaled_dot_product_attention as called by
 * tf:integrations/sdpa_attention.py:154-163 from tf:clip:317-330 (bidirectional, 16
 * heads, head_dim 64) and tf:llama:264-277 (is_causal, enable_gqa, head_dim 128), and its
 * backward.  q/k/v/o are [B][S][heads][head_dim] views with a row (token) stride in
 * elements, so they may alias column slices of one packed qkv buffer.  lse is
 * [B][Hq][S] fp32 (natural-log-sum-exp of the scaled scores), written by fwd, read by bwd.
 * delta is a [B][Hq][S] fp32 scratch written by bwd.  head_dim in {64, 128}.
 */
int zk_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int64_t B,
                int64_t S, int64_t Hq, int64_t Hkv, int64_t head_dim, int64_t ldq, int64_t ldk,
                int64_t ldv, int64_t ldo, float scale, int causal, hipStream_t stream);
/* The same forward (same contract, same call sites replaced) on the long-sequence structure: one wave per SIMD,
 * 64 query rows per wave as two 32-row halves that share every K / V fragment read, the softmax of one 32-key
 * sub-tile placed between the MFMAs of its neighbours, O rescaled only when a row maximum grows by more than 2^8
 * (results differ from zk_attn_fwd by rounding only).  Measured slower than zk_attn_fwd at the training step's
 * S = 577 / 768 (one workgroup per CU leaves prologue and epilogue uncovered) and within 7 % of it at S >= 3072
 * (DESIGN.md §4.2): zk_attn_fwd stays the step's kernel, this entry exists for tests and benchmarks. */
int zk_attn_fwd_w64(const void* q, const void* k, const void* v, void* o, float* lse, int64_t B,
                    int64_t S, int64_t Hq, int64_t Hkv, int64_t head_dim, int64_t ldq, int64_t ldk,
                    int64_t ldv, int64_t ldo, float scale, int causal, hipStream_t stream);
int zk_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                const float* lse, float* delta, void* dq, void* dk, void* dv, int64_t B,
                int64_t S, int64_t Hq, int64_t Hkv, int64_t head_dim, int64_t ldq, int64_t ldk,
                int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv, float scale,
                int causal, hipStream_t stream);
/* zk_attn_bwd with the rotary backward folded into its store epilogues: dq and dk leave as the
 * gradients of the UN-rotated q / k, bit for bit what zk_rope(..., inverse=1) over the dq / dk
 * column blocks would leave after zk_attn_bwd (tf:llama:140-160 apply_rotary_pos_emb's backward,
 * called from tf:llama:248-262).  cos_t / sin_t are the [S][head_dim/2] fp32 tables of zk_rope;
 * both NULL = zk_attn_bwd. */
int zk_attn_bwd_rope(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                     const float* lse, float* delta, void* dq, void* dk, void* dv, int64_t B,
                     int64_t S, int64_t Hq, int64_t Hkv, int64_t head_dim, int64_t ldq,
                     int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk,
                     int64_t lddv, float scale, int causal, const float* cos_t,
                     const float* sin_t, hipStream_t stream);

/* Single-query attention over a KV cache: one new token per sequence against `lens[b]` cached keys
 * (the current token's k/v already appended), as tf generate(use_cache=True) does for the reference's
 * greedy decoding (ref:inference/vlm_infer.py:259-270; tf:llama:264-277 with a DynamicCache).
 * q [B][Hq*hd] (row stride ldq); caches [B][max rows][Hkv*hd] with row strides ldk / ldv and batch
 * strides in elements; lens int32 [B] on the device; out [B][Hq*hd].  head_dim 64 or 128,
 * max_len (the largest lens[b]) <= 15000. */
int zk_attn_decode(const void* q, const void* k_cache, const void* v_cache, void* out,
                   const int* lens, int64_t B, int64_t Hq, int64_t Hkv, int64_t head_dim,
                   int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t batch_stride_k,
                   int64_t batch_stride_v, int64_t max_len, float scale, hipStream_t stream);

/* The same with the new token's RoPE and cache append folded in and the cache split over workgroups
 * (flash-decoding): replaces zk_rope + the k/v append + zk_attn_decode of one decoding step by a
 * partial pass (one workgroup per 64 cached keys, kv head and sequence; the q heads of a GQA group share
 * the K / V rows) and a merge pass.  qkv [B][ldqkv]: the new token's packed q|k|v BEFORE RoPE
 * (tf:llama:254-259), left untouched; cos_row / sin_row: the fp32 table rows of position `pos`
 * ([head_dim / 2]); caches [B][cache_rows][Hkv*hd]: row `pos` is written (rotated k, v), rows 0..pos are
 * attended; every sequence is at the same position.  ws: caller-owned fp32 workspace of at least
 * B * Hq * (pos / 64 + 1) * (head_dim + 2) floats.  Hq / Hkv <= 16, head_dim 64 or 128. */
int zk_attn_decode_fused(const void* qkv, const float* cos_row, const float* sin_row, void* k_cache,
                         void* v_cache, void* out, float* ws, int64_t ws_floats, int64_t B, int64_t Hq,
                         int64_t Hkv, int64_t head_dim, int64_t ldqkv, int64_t ldo, int64_t cache_rows,
                         int64_t pos, float scale, hipStream_t stream);

/* zk_attn_decode_fused for callers that cannot allocate or carry a workspace — code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  NO counterpart in the reference.  Same arguments minus
 * ws / ws_floats, same kernels, same results bit for bit; the fp32 partials live in a device array inside the
 * library's code object (ZK_ATTN_DECODE_STATIC_FLOATS floats = 8 MiB, held on every GPU the library runs on until
 * the process ends).  Calls hipMalloc / hipFree / no allocating routine, ever; the first call on each device looks
 * the array's address up (hipGetSymbolAddress; make that first call outside a capture).
 * Limit: B * Hq * (pos / 64 + 1) * (head_dim + 2) <= ZK_ATTN_DECODE_STATIC_FLOATS, otherwise -7 (E2BIG) and nothing
 * is launched (e.g. 8 q heads x hd 128, B = 1: 129,000 cached keys; 32 heads, B = 4: 8,064).
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (different streams) race.
 * The current device must be the one `stream` belongs to (each GPU has its own copy of the array at its own
 * address): -22 otherwise.  Argument errors are reported (-22) before the size limit (-7). */
#define ZK_ATTN_DECODE_STATIC_FLOATS (2 * 1024 * 1024)
int zk_attn_decode_fused_static(const void* qkv, const float* cos_row, const float* sin_row, void* k_cache,
                                void* v_cache, void* out, int64_t B, int64_t Hq, int64_t Hkv, int64_t head_dim,
                                int64_t ldqkv, int64_t ldo, int64_t cache_rows, int64_t pos, float scale,
                                hipStream_t stream);

/* One Llama decoder layer on a single-token decoding step, enqueued natively in two halves so a
 * tensor-parallel host can all-reduce the partial sums between them (tf:llama:295-324 with a KV cache;
 * ref:inference/vlm_infer.py:259-270).  Head counts and F are the LOCAL (sharded) ones; wqkv / wgu are
 * the packed q|k|v and gate|up weights ([rows][d], k contiguous); the caches are
 * [B][cache_rows][nkv*hd]; lens (device int32 [B]) must hold pos + 1; h / qkv / ao / gu / hm are
 * caller-owned scratch rows ([B][d], [B][(nh+2nkv)hd], [B][nh*hd], [B][2F], [B][F]); attn_ws
 * (optional, attn_ws_floats fp32 values, sized as zk_attn_decode_fused asks for cache_rows keys) switches
 * the attention to the split-cache kernels, otherwise zk_rope + append + zk_attn_decode run.
 *   attn: out = (add_residual ? x : 0) + o_proj(attention(rope(qkv(rmsnorm(x)))))   and appends k, v
 *   mlp : out = (add_residual ? xa : 0) + down(silu(gate(rmsnorm(xa))) * up(rmsnorm(xa)))
 * B <= 4.  Allocates nothing, never synchronises. */
typedef struct {
  int64_t B, d, nh, nkv, hd, F, cache_rows, pos;
  float eps;
  const void *ln1_w, *wqkv, *wo, *ln2_w, *wgu, *wdown;
  const float *cos_t, *sin_t;
  void *k_cache, *v_cache;
  const int* lens;
  void *h, *qkv, *ao, *gu, *hm;
  float* attn_ws;
  int64_t attn_ws_floats;
} zk_decode_layer_t;
int zk_llama_decode_attn(const zk_decode_layer_t* L, const void* x, void* out, int add_residual,
                         hipStream_t stream);
int zk_llama_decode_mlp(const zk_decode_layer_t* L, const void* xa, void* out, int add_residual,
                        hipStream_t stream);

/* ---- LayerNorm (tf:clip:358-360,370,379,605,642; ref:adapter.py:94) ------------------
 * y = (x - mean) * rstd * gamma + beta over the last dim D (<= 8192, multiple of 8),
 * fp32 statistics.  mean/rstd [rows] fp32 are optional outputs of fwd (needed by bwd).
 * bwd: dx = dres + LN'(dy).  The affine gradients come out as PARTIAL sums: dgamma / dbeta are fp32
 * [nparts][D] buffers that the call OVERWRITES (part p = the rows workgroup p walked; 1 <= nparts <=
 * ZK_NORM_MAX_PARTS, about rows / 16 is a good choice); the caller adds the parts in index order
 * (zk_f32_to_bf16_multi does while converting).  No atomics: results are bit-reproducible.  Both may be
 * null when the affine parameters are frozen (nparts is then ignored).
 */
#define ZK_NORM_MAX_PARTS 512
int zk_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean,
                     float* rstd, int64_t rows, int64_t D, float eps, hipStream_t stream);
int zk_layernorm_bwd(const void* dy, const void* x, const void* gamma, const float* mean,
                     const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta,
                     int64_t nparts, int64_t rows, int64_t D, hipStream_t stream);

/* ---- RMSNorm (tf:llama:62-67) ----------------------------------------------------------
 * y = w * bf16(x * rsqrt(mean(x^2) + eps)); bwd gives dx = dres + RMS'(dy); the weight gradient
 * sum_rows dy[r][c] * bf16(x[r][c] * rstd[r]) is a separate call, needed only where decoder layers train
 * (stage "pretrain": the translator layer, ref:configs/zero/train_llama.py:281-307); like the LayerNorm
 * affine gradients it is written as fp32 partial sums dweight[nparts][D] (overwritten, no atomics).
 */
int zk_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t rows, int64_t D,
                   float eps, hipStream_t stream);
int zk_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd,
                   const void* dres, void* dx, int64_t rows, int64_t D, hipStream_t stream);
int zk_rmsnorm_dweight(const void* dy, const void* x, const float* rstd, float* dweight,
                       int64_t nparts, int64_t rows, int64_t D, hipStream_t stream);

/* ---- RoPE (tf:llama:130-160; cos/sin table tf:llama:113-127) ---------------------------
 * In place on `heads` consecutive heads of width head_dim inside rows of stride ld:
 * x' = x*cos + rotate_half(x)*sin, half-split layout.  cos/sin are [S][head_dim/2] fp32
 * holding the bf16-rounded table values; row r of x is position r % S.  inverse != 0
 * applies the transposed rotation (backward).
 */
int zk_rope(void* x, const float* cos_t, const float* sin_t, int64_t rows, int64_t S,
            int64_t heads, int64_t head_dim, int64_t ld, int inverse, hipStream_t stream);

/* ---- SwiGLU (tf:llama:175) --------------------------------------------------------------
 * gu is [rows][2F] = [gate | up]; h = silu(gate) * up.  bwd writes d_gu [rows][2F].
 */
int zk_swiglu_fwd(const void* gu, void* h, int64_t rows, int64_t F, hipStream_t stream);
int zk_swiglu_bwd(const void* dh, const void* gu, void* dgu, int64_t rows, int64_t F,
                  hipStream_t stream);

/* ---- column sum: bias gradients (autograd of the biased Linears) -----------------------
 * out[p][n] = sum over the rows of part p of dy[m][n]: fp32 [nparts][N], OVERWRITTEN (rows are dealt to the
 * parts in groups of 8; no atomics, bit-reproducible); the caller adds the parts in index order.
 */
int zk_colsum(const void* dy, float* out, int64_t rows, int64_t N, int64_t ld, int64_t nparts,
              hipStream_t stream);

/* ---- activation backward (autograd of tf:activations.py:117-123 / nn.GELU) -------------
 * dx = dy * act'(pre), kind = ZK_EPI_QUICK_GELU | ZK_EPI_GELU_ERF; in place allowed.
 */
int zk_act_bwd(const void* dy, const void* pre, void* dx, int64_t n, int kind,
               hipStream_t stream);

/* ---- patch embedding front end (tf:clip:202-218) -----------------------------------------
 * im2col: pixels [B][3][H][W] bf16 -> cols [B*G*G][Kpad] (k = c*P*P + i*P + j, zero padded).
 * embed_finish: out[b][0] = cls + pos[0]; out[b][1+p] = bf16(patch[b*G*G+p]) + pos[1+p].
 */
int zk_patch_im2col(const void* pixels, void* cols, int64_t B, int64_t H, int64_t W, int64_t P,
                    int64_t Kpad, hipStream_t stream);
int zk_embed_finish(const void* patch, const void* cls, const void* pos, void* out, int64_t B,
                    int64_t T, int64_t D, hipStream_t stream);

/* ---- row gather / splice (ref:methods/engine_ar_instruction_llama.py:186-236) -----------
 * dst[r] = src_table_k[idx] by a per-row source code:  code >= 0 : table0[code]
 * (token embedding, wte); code == -1 : zeros (right padding); code <= -2 : table1[-2-code]
 * (row of the image-embedding block).  Rows are D bf16.  gather_rows_add: dst[r] += ...
 * is used for the backward scatter through the inverse map.
 */
/* dst[C][R] (row stride ld_dst) = transpose of src[R][C] (row stride ld_src); R, C multiples of 8.  Replaces
 * `weight.t().contiguous()` (ATen strided copy) where a k-contiguous copy of a weight is prepared for the dgrad
 * GEMMs (no counterpart in the reference: it reads W through F.linear's backward). */
int zk_transpose_bf16(const void* src, void* dst, int64_t R, int64_t C, int64_t ld_src, int64_t ld_dst,
                      hipStream_t stream);

int zk_gather_rows(const void* table0, const void* table1, const int32_t* codes, void* dst,
                   int64_t rows, int64_t D, int accumulate, hipStream_t stream);

/* ---- cross entropy on bf16 logits (ref:engine:280-305) -----------------------------------
 * For each row: fp32 log-softmax over V, loss = -logp[target] unless target == -100.
 * Adds the row losses into loss_sum[0] (fp32) and the number of non-ignored rows into
 * loss_sum[1]; when dlogits != null writes (softmax - onehot) (0 for ignored rows) as bf16
 * (may alias logits).  row_loss (optional fp32 [rows] scratch): when given, every row's loss is stored there
 * and a second one-block kernel adds them in index order (bit-reproducible loss); when null the rows add
 * themselves with fp32 atomics (order-dependent last bits).
 */
int zk_cross_entropy(const void* logits, const int32_t* targets, float* loss_sum, float* row_loss,
                     void* dlogits, int64_t rows, int64_t V, int64_t ld, hipStream_t stream);

/* ---- optimizer tail (ref:train.py:570-580: clip_grad_norm_ + AdamW.step) -----------------
 * sumsq: out[0] += sum(g^2) over a flat bf16 buffer; parts (optional fp32 [ZK_SUMSQ_PARTS] scratch): per-block
 * sums stored there and added in index order by a second one-block kernel (bit-reproducible clip coefficient),
 * otherwise fp32 atomics.  adamw_step: flat bf16 params/grads,
 * fp32 moments; the clip coefficient min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)) is
 * computed on the device from sumsq (max_norm <= 0: no clipping); grad_scale multiplies
 * every gradient first (1/world_size for the DDP average).
 */
#define ZK_SUMSQ_PARTS 1024
int zk_sumsq(const void* g, float* out, float* parts, int64_t n, float scale, hipStream_t stream);
int zk_adamw_step(void* p, const void* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int64_t step, float max_norm,
                  const float* sumsq, float grad_scale, hipStream_t stream);

/* ---- fp32 -> bf16 finalize for accumulated parameter gradients ---------------------------
 * dst = bf16(src * scale) or dst = bf16(float(dst) + src * scale) when accumulate.
 */
int zk_f32_to_bf16(const float* src, void* dst, int64_t n, float scale, int accumulate,
                   hipStream_t stream);
/* The same for up to 16 independent segments in one launch (host arrays of `count` entries): the
 * bias / norm-affine gradients of one layer, each converted from its fp32 partial sums (nparts[i] rows of n[i]
 * values, part_stride[i] elements apart, added in part order; nparts null = 1 each, part_stride null = n[i])
 * into its own slot of the flat gradient buffer. */
int zk_f32_to_bf16_multi(const float* const* src, void* const* dst, const int64_t* n,
                         const int* accumulate, const int* nparts, const int64_t* part_stride, int count,
                         float scale, hipStream_t stream);

/* ---- image data path -------------------------------------------------
 * What ref:loader.py:171-195 does per sample with PIL + CLIPImageProcessor (PIL backend):
 * expand2square (ref:loader.py:609-620), Image.resize (bicubic, antialiased; ref:loader.py:677,759
 * and tf:image_transforms.py resize), centre crop, anyres canvas + tiling (ref:loader.py:731-789),
 * rescale + normalize (tf:image_transforms.py:118-122, normalize).  u8 RGB images are [h][w][3]
 * with a row stride in BYTES.  The resampler is Pillow's 8-bit one (Pillow 12.2.0
 * src/libImaging/Resample.c): `bounds` = (first source index, tap count) per output index, `kk` =
 * ksize int32 coefficients per output index in 22-bit fixed point, both made by the host exactly as
 * precompute_coeffs / normalize_coeffs_8bpc make them.  Results are bit-identical to PIL.
 *   resample_rows : horizontal pass over `rows` rows of the VIRTUAL image (src pasted at
 *                   (paste_x, paste_y) on a canvas of colour bg_rgb = r | g<<8 | b<<16; negative
 *                   paste = crop), out [rows][out_w][3] packed.
 *   resample_cols : vertical pass over byte columns, in [in_h][row_bytes] -> out [out_h][row_bytes].
 *   patches_normalize : out [n_patches][3][patch][patch] (fp32, or bf16 when out_bf16) where patch p
 *                   is canvas rows (p / grid_x) * patch.., columns (p % grid_x) * patch.. of the
 *                   virtual image, mapped through lut[3][256] (the host's rescale+normalize of every
 *                   byte value per channel).
 */
int zk_u8_resample_rows(const uint8_t* src, int64_t src_h, int64_t src_w, int64_t src_ld, int64_t rows,
                        int64_t paste_x, int64_t paste_y, uint32_t bg_rgb, const int32_t* bounds,
                        const int32_t* kk, int ksize, int64_t out_w, uint8_t* out, hipStream_t stream);
int zk_u8_resample_cols(const uint8_t* in, int64_t in_h, int64_t row_bytes, const int32_t* bounds,
                        const int32_t* kk, int ksize, int64_t out_h, uint8_t* out, hipStream_t stream);
int zk_u8_patches_normalize(const uint8_t* img, int64_t img_h, int64_t img_w, int64_t img_ld,
                            int64_t paste_x, int64_t paste_y, uint32_t bg_rgb, int64_t grid_x,
                            int64_t patch, int64_t n_patches, const float* lut, void* out, int out_bf16,
                            hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ZERO_HIP_H_ */
