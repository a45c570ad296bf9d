// Image data path on the device: what ref:loader.py:171-195,609-688 and the
// CLIPImageProcessor (PIL backend) do per sample on CPU workers — pad to square with the mean colour,
// antialiased bicubic resize, centre crop / anyres tiling, 1/255 rescale, mean/std normalisation —
// as three byte kernels.  The arithmetic is Pillow's 8-bit resampler (third-party, not in the
// reference tree: Pillow 12.2.0 src/libImaging/Resample.c, ImagingResampleHorizontal_8bpc /
// ImagingResampleVertical_8bpc): two passes, int32 accumulation of u8 * 22-bit fixed-point
// coefficients, rounding to u8 between the passes.  Coefficients and window bounds are made on the
// host in double precision exactly as precompute_coeffs / normalize_coeffs_8bpc do (zero_amd/loader.py)
// so the result is BIT-IDENTICAL to PIL; the final float comes from a 3x256 table the host fills with
// numpy's own roundings (float32(float64(v) * (1/255)) - mean) / std.
//
// HBM-bound byte work (a 336x336 RGB output is 339 KB in, 1.35 MB out as fp32): one thread per output
// pixel / byte, consecutive threads on consecutive bytes; nothing here is shaped into a GEMM.
#include "common.h"
#include "../../include/zero_hip.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// The image the resampler reads: src [sh][sw][3] (row stride ld bytes) pasted at (px, py) into a
// background canvas; px / py may be negative (crop).  Anything outside src is the background colour.
struct VSrc {
  const uint8_t* p;
  int sh, sw;
  long long ld;
  int px, py;
  unsigned int bg;   // r | g << 8 | b << 16
};

// horizontal pass: out [rows][out_w][3]
__global__ __launch_bounds__(256) void resample_rows_kernel(VSrc s, const int* __restrict__ bounds,
                                                            const int* __restrict__ kk, int ksize, int out_w,
                                                            uint8_t* __restrict__ out) {
  const int ox = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (ox >= out_w) return;
  const int xmin = bounds[2 * ox], n = min(bounds[2 * ox + 1], ksize);
  const int* k = kk + (size_t)ox * ksize;
  const int iy = y - s.py;
  const bool row_in = iy >= 0 && iy < s.sh;
  const uint8_t* row = s.p + (long long)(row_in ? iy : 0) * s.ld;
  const int b0 = s.bg & 255, b1 = (s.bg >> 8) & 255, b2 = (s.bg >> 16) & 255;
  int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
  for (int i = 0; i < n; ++i) {
    const int x = xmin + i - s.px;
    int r = b0, g = b1, b = b2;
    if (row_in && x >= 0 && x < s.sw) {
      r = row[3 * x];
      g = row[3 * x + 1];
      b = row[3 * x + 2];
    }
    const int w = k[i];
    a0 += r * w;
    a1 += g * w;
    a2 += b * w;
  }
  uint8_t* o = out + ((size_t)y * out_w + ox) * 3;
  o[0] = (uint8_t)clip8(a0);
  o[1] = (uint8_t)clip8(a1);
  o[2] = (uint8_t)clip8(a2);
}

// vertical pass over byte columns: in [in_h][row_bytes] -> out [out_h][row_bytes]
__global__ __launch_bounds__(256) void resample_cols_kernel(const uint8_t* __restrict__ in, int in_h, int row_bytes,
                                                            const int* __restrict__ bounds,
                                                            const int* __restrict__ kk, int ksize,
                                                            uint8_t* __restrict__ out) {
  const int col = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
  if (col >= row_bytes) return;
  const int ymin = bounds[2 * oy], n = min(bounds[2 * oy + 1], ksize);
  const int* k = kk + (size_t)oy * ksize;
  int a = 1 << (PRECISION_BITS - 1);
  for (int i = 0; i < n; ++i) {
    const int y = ymin + i;
    if (y >= 0 && y < in_h) a += (int)in[(size_t)y * row_bytes + col] * k[i];
  }
  out[(size_t)oy * row_bytes + col] = (uint8_t)clip8(a);
}

// crop / tile / rescale / normalise: out [P][3][ps][ps]; patch p covers canvas rows (p / gx) * ps ..,
// columns (p % gx) * ps ..; the canvas is the virtual image (background + pasted src).
template <bool BF16>
__global__ __launch_bounds__(256) void patches_normalize_kernel(VSrc s, int gx, int ps, long long total,
                                                                const float* __restrict__ lut,
                                                                void* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int x = (int)(i % ps);
  long long t = i / ps;
  const int y = (int)(t % ps);
  t /= ps;
  const int c = (int)(t % 3), p = (int)(t / 3);
  const int ix = (p % gx) * ps + x - s.px, iy = (p / gx) * ps + y - s.py;
  int v = (s.bg >> (8 * c)) & 255;
  if (ix >= 0 && ix < s.sw && iy >= 0 && iy < s.sh) v = s.p[(long long)iy * s.ld + 3 * ix + c];
  const float f = lut[c * 256 + v];
  if (BF16)
    reinterpret_cast<bf16_t*>(out)[i] = f2bf(f);
  else
    reinterpret_cast<float*>(out)[i] = f;
}

bool fits_int(int64_t v) { return v >= -0x3FFFFFFFLL && v <= 0x3FFFFFFFLL; }

}  // namespace

extern "C" int zk_u8_resample_rows(const uint8_t* src, int64_t src_h, int64_t src_w, int64_t src_ld, int64_t rows,
                                   int64_t paste_x, int64_t paste_y, uint32_t bg_rgb, const int32_t* bounds,
                                   const int32_t* kk, int ksize, int64_t out_w, uint8_t* out, hipStream_t stream) {
  if (!src || !bounds || !kk || !out || src_h <= 0 || src_w <= 0 || src_ld < 3 * src_w || rows <= 0 || out_w <= 0 ||
      ksize <= 0)
    return ZK_EINVAL;
  if (!fits_int(src_h) || !fits_int(src_w) || !fits_int(rows) || !fits_int(out_w) || !fits_int(paste_x) ||
      !fits_int(paste_y) || rows > 65535)
    return ZK_EINVAL;
  VSrc s{src, (int)src_h, (int)src_w, (long long)src_ld, (int)paste_x, (int)paste_y, bg_rgb};
  dim3 grid((unsigned)((out_w + 255) / 256), (unsigned)rows);
  hipLaunchKernelGGL(resample_rows_kernel, grid, dim3(256), 0, stream, s, (const int*)bounds, (const int*)kk, ksize,
                     (int)out_w, out);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

extern "C" int zk_u8_resample_cols(const uint8_t* in, int64_t in_h, int64_t row_bytes, const int32_t* bounds,
                                   const int32_t* kk, int ksize, int64_t out_h, uint8_t* out, hipStream_t stream) {
  if (!in || !bounds || !kk || !out || in_h <= 0 || row_bytes <= 0 || out_h <= 0 || ksize <= 0) return ZK_EINVAL;
  if (!fits_int(in_h) || !fits_int(row_bytes) || out_h > 65535) return ZK_EINVAL;
  dim3 grid((unsigned)((row_bytes + 255) / 256), (unsigned)out_h);
  hipLaunchKernelGGL(resample_cols_kernel, grid, dim3(256), 0, stream, in, (int)in_h, (int)row_bytes,
                     (const int*)bounds, (const int*)kk, ksize, out);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}

extern "C" int zk_u8_patches_normalize(const uint8_t* img, int64_t img_h, int64_t img_w, int64_t img_ld,
                                       int64_t paste_x, int64_t paste_y, uint32_t bg_rgb, int64_t grid_x,
                                       int64_t patch, int64_t n_patches, const float* lut, void* out, int out_bf16,
                                       hipStream_t stream) {
  if (!img || !lut || !out || img_h <= 0 || img_w <= 0 || img_ld < 3 * img_w || grid_x <= 0 || patch <= 0 ||
      n_patches <= 0)
    return ZK_EINVAL;
  if (!fits_int(img_h) || !fits_int(img_w) || !fits_int(paste_x) || !fits_int(paste_y) || !fits_int(grid_x * patch) ||
      !fits_int(n_patches / grid_x * patch + patch))
    return ZK_EINVAL;
  const long long total = (long long)n_patches * 3 * patch * patch;
  if (total > 0x7FFFFFFFLL * 256) return ZK_EINVAL;
  VSrc s{img, (int)img_h, (int)img_w, (long long)img_ld, (int)paste_x, (int)paste_y, bg_rgb};
  dim3 grid((unsigned)((total + 255) / 256));
  if (out_bf16)
    hipLaunchKernelGGL(patches_normalize_kernel<true>, grid, dim3(256), 0, stream, s, (int)grid_x, (int)patch, total,
                       lut, out);
  else
    hipLaunchKernelGGL(patches_normalize_kernel<false>, grid, dim3(256), 0, stream, s, (int)grid_x, (int)patch, total,
                       lut, out);
  ZK_CHECK_LAUNCH();
  return ZK_OK;
}
