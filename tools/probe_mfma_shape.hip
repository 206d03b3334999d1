// Probe (GPU, standalone; not a product path): sustained issue rate of the two fp16 MFMA shapes and of the two
// block-scaled fp4 shapes on the whole chip, one or two waves per SIMD, random operands (the clock under MFMA load
// depends on the data).  Same FLOPs per loop pass in every variant: 128 accumulator registers per wave.
//   build: hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_shape.hip -o /tmp/probe_mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// VARIANT 0: 32 x v_mfma_f32_16x16x32_f16 per pass (4 A x 8 B fragments); 1: 16 x v_mfma_f32_32x32x16_f16 (2 A x 4 B
// fragments x 2 K halves); 2: 32 x 16x16x128 fp4; 3: 16 x 32x32x64 fp4 (2 x 4 x 2)
template <int VARIANT>
__global__ __launch_bounds__(512) void k_rate(float* out, long long* cyc, int iters, const int* seed) {
  const int l = threadIdx.x;
  f16x8 a[8], b[8];
  i32x8 a4[8], b4[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a[i][j] = (_Float16)(0.013f * (float)((seed[(l + i) & 63] >> j & 31) - 16));
      b[i][j] = (_Float16)(0.017f * (float)((seed[(l + 3 * i + 1) & 63] >> j & 31) - 16));
    }
    a4[i] = i32x8{seed[(l + i) & 63], seed[(l + 2 * i) & 63], seed[(l + 3 * i) & 63], seed[(l + 5 * i) & 63], 0, 0, 0, 0};
    b4[i] = i32x8{seed[(l + 7 * i) & 63], seed[(l + 11 * i) & 63], seed[(l + 13 * i) & 63], seed[(l + 17 * i) & 63], 0, 0, 0, 0};
  }
  f32x4 acc[32];
  f32x16 acc32[8];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (VARIANT == 0) {
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[p * 8 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[q], a[p], acc[p * 8 + q], 0, 0, 0);
    } else if constexpr (VARIANT == 1) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc32[p * 4 + q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[q * 2 + k], a[p * 2 + k], acc32[p * 4 + q], 0, 0, 0);
    } else if constexpr (VARIANT == 2) {
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 8; ++q)
          acc[p * 8 + q] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b4[q], a4[p], acc[p * 8 + q], 4, 4, 0, 127, 0, 127);
    } else {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc32[p * 4 + q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b4[q * 2 + k], a4[p * 2 + k], acc32[p * 4 + q], 4, 4, 0, 127, 0, 127);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc32[i][0] + acc32[i][15];
  out[blockIdx.x * blockDim.x + l] = s;
  if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
static void run(const char* name, int threads, int iters, float* out, long long* cyc, const int* seed, double flop_per_pass) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = 256;
  hipLaunchKernelGGL(k_rate<V>, dim3(grid), dim3(threads), 0, 0, out, cyc, iters / 10, seed);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_rate<V>, dim3(grid), dim3(threads), 0, 0, out, cyc, iters, seed);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  long long c[256];
  CK(hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost));
  double avg = 0;
  for (int i = 0; i < grid; ++i) avg += (double)c[i] / grid;
  const int waves = grid * threads / 64;
  const double tf = flop_per_pass * iters * waves / (ms * 1e-3) / 1e12;
  const int per_simd = threads / 256;
  printf("%-28s %d wave/SIMD: %8.3f ms  %8.1f TFLOP/s  shader cycles per pass per wave %.1f (x%d waves = %.1f per SIMD)  clock %.2f GHz\n",
         name, per_simd, ms, tf, avg / iters, per_simd, avg / iters, avg / (ms * 1e-3) / 1e9);
}

int main() {
  float* out;
  long long* cyc;
  int* seed;
  CK(hipMalloc(&out, 256 * 512 * 4));
  CK(hipMalloc(&cyc, 256 * 8));
  CK(hipMalloc(&seed, 64 * 4));
  int h[64];
  srand(7);
  for (int i = 0; i < 64; ++i) h[i] = rand() ^ (rand() << 11);
  CK(hipMemcpy(seed, h, sizeof h, hipMemcpyHostToDevice));
  const int iters = 20000;
  const double f16 = 32.0 * 16 * 16 * 32 * 2, f4 = 32.0 * 16 * 16 * 128 * 2;
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("f16 16x16x32 x32", 256, iters, out, cyc, seed, f16);
    run<1>("f16 32x32x16 x16", 256, iters, out, cyc, seed, f16);
    run<0>("f16 16x16x32 x32", 512, iters, out, cyc, seed, f16);
    run<1>("f16 32x32x16 x16", 512, iters, out, cyc, seed, f16);
    run<2>("fp4 16x16x128 x32", 512, iters, out, cyc, seed, f4);
    run<3>("fp4 32x32x64 x16", 512, iters, out, cyc, seed, f4);
  }
  return 0;
}
