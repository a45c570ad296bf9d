// Calibration micro-benchmark: sustained bf16 MFMA rate on this MI355X (random operands), with and
// without the GEMM's LDS read traffic.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>  // 0: 16x16x32 regs only, 1: 32x32x16 regs only, 2: 16x16x32 + 8 ds_read_b128 per 16 MFMA
__global__ __launch_bounds__(512) void k(const unsigned short* in, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 65536 / 2; i += 512) ((unsigned short*)lds)[i] = in[(i * 7 + blockIdx.x) & 65535];
  __syncthreads();
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = *(const bf16x8*)(lds + ((tid * 16 + i * 8192) & 65535)); b[i] = *(const bf16x8*)(lds + ((tid * 16 + i * 8192 + 4096) & 65535)); }
  if (MODE == 1) {
    f32x16 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + r) & 3], acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 512 + tid] = s;
  } else {
    f32x4 acc[4][4] = {};
    for (int it = 0; it < iters; ++it) {
      if (MODE == 2) {
        int base = ((it & 7) * 8192 + (lane & 15) * 128 + ((lane >> 4) ^ (lane & 7)) * 16) & 65535;
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = *(const bf16x8*)(lds + ((base + i * 2048) & 65535)); b[i] = *(const bf16x8*)(lds + ((base + i * 2048 + 32768) & 65535)); }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    out[blockIdx.x * 512 + tid] = s;
  }
}

int main() {
  unsigned short* in; float* out;
  hipMalloc(&in, 65536 * 2); hipMalloc(&out, 4096 * 512 * 4);
  unsigned short* h = (unsigned short*)malloc(65536 * 2);
  srand(1);
  for (int i = 0; i < 65536; ++i) { float f = (rand() / (float)RAND_MAX) * 2 - 1; unsigned int u; memcpy(&u, &f, 4); h[i] = u >> 16; }
  hipMemcpy(in, h, 65536 * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int mode = 0; mode < 3; ++mode) {
    for (int blocks : {256, 512}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, in, out, iters);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, in, out, iters);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * 8 * iters * 16 * (mode == 1 ? 32.0 * 32 * 16 * 2 : 16.0 * 16 * 32 * 2);
        if (rep) printf("mode %d blocks %d: %.2f ms  %.0f TFLOP/s\n", mode, blocks, ms, fl / ms / 1e9);
      }
    }
  }
  return 0;
}
