// Probe (GPU, standalone; not a product path): operand layout, scale semantics and issue rate of the block-scaled
// v_mfma_scale_f32_16x16x128_f8f6f4 with fp4 (e2m1) operands, and of v_cvt_scalef32_pk_fp4_f16 - the facts the
// "fp16 + MX-fp4 correction pass" GEMM mode is built on.
//   build: hipcc --offload-arch=gfx950 -O3 tools/probe_mx.hip -o build/probe_mx
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static const float kFp4[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
static float dec4(int c) { return (c & 8 ? -1.f : 1.f) * kFp4[c & 7]; }

__global__ void k_layout(const int* a, const int* b, float* d, int sa, int sb, int variant) {
  const int l = threadIdx.x;
  i32x8 A = {a[l * 4], a[l * 4 + 1], a[l * 4 + 2], a[l * 4 + 3], 0, 0, 0, 0};
  i32x8 B = {b[l * 4], b[l * 4 + 1], b[l * 4 + 2], b[l * 4 + 3], 0, 0, 0, 0};
  f32x4 c = {0, 0, 0, 0};
  if (variant == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 4, 4, 0, sa, 0, sb);
  if (variant == 1) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 4, 4, 1, sa, 2, sb);
  if (variant == 2) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 4, 4, 3, sa, 3, sb);
  *(f32x4*)(d + l * 4) = c;
}

__global__ void k_cvt(const _Float16* x, unsigned* o, float scale) {
  const int l = threadIdx.x;
  h2 v0 = {x[l * 8], x[l * 8 + 1]}, v1 = {x[l * 8 + 2], x[l * 8 + 3]}, v2 = {x[l * 8 + 4], x[l * 8 + 5]},
     v3 = {x[l * 8 + 6], x[l * 8 + 7]};
  unsigned r = 0xdeadbeefu;
  r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, v0, scale, 0);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, v1, scale, 1);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, v2, scale, 2);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, v3, scale, 3);
  o[l] = r;
}

// Issue-rate probe: `iters` rounds of (NF f16 MFMAs + NX MX-fp4 MFMAs) over 32 independent accumulators, the
// accumulator pattern of a wave's 128 x 64 tile.  mode bit 0: also convert 8 fragments (32 cvt) per round.
template <int NF, int NX, int CVT>
__global__ __launch_bounds__(512) void k_rate(float* out, int iters, const int* seed) {
  const int l = threadIdx.x;
  f32x4 acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = f32x4{0, 0, 0, 0};
  f16x8 xa[8], wb[4];
  i32x8 x4[8], w4[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    for (int j = 0; j < 8; ++j) xa[i][j] = (_Float16)(0.01f * (float)((seed[l & 63] + i * 8 + j) & 15));
    x4[i] = i32x8{seed[(l + i) & 63], seed[(l + 2 * i) & 63], seed[(l + 3 * i) & 63], seed[(l + 5 * i) & 63], 0, 0, 0, 0};
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < 8; ++j) wb[i][j] = (_Float16)(0.02f * (float)((seed[(l + 7) & 63] + i * 8 + j) & 15));
    w4[i] = i32x8{seed[(l + 11 * i) & 63], seed[(l + 13 * i) & 63], seed[(l + 17 * i) & 63], seed[(l + 19 * i) & 63], 0, 0, 0, 0};
  }
  const int sa = 127, sb = 127;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < NF; ++rep) {
#pragma unroll
      for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[p * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[q], xa[p], acc[p * 4 + q], 0, 0, 0);
      if (CVT) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          unsigned r = (unsigned)x4[p][rep & 3];
          r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, h2{xa[p][0], xa[p][1]}, 1.0f, 0);
          r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, h2{xa[p][2], xa[p][3]}, 1.0f, 1);
          r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, h2{xa[p][4], xa[p][5]}, 1.0f, 2);
          r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, h2{xa[p][6], xa[p][7]}, 1.0f, 3);
          x4[p][rep & 3] = (int)r;
        }
      }
    }
#pragma unroll
    for (int rep = 0; rep < NX; ++rep) {
#pragma unroll
      for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          acc[p * 4 + q] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w4[q], x4[p], acc[p * 4 + q], 4, 4, 0, sa, 0, sb);
    }
  }
  f32x4 s = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 32; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + l] = s[0] + s[1] + s[2] + s[3];
}

template <int NF, int NX, int CVT>
static void rate(const char* name, int threads) {
  float* d;
  int* seed;
  CK(hipMalloc(&d, 256 * 512 * 4 * sizeof(float)));
  CK(hipMalloc(&seed, 64 * sizeof(int)));
  int hs[64];
  for (int i = 0; i < 64; ++i) hs[i] = rand() & 0x77777777;
  CK(hipMemcpy(seed, hs, sizeof hs, hipMemcpyHostToDevice));
  const int iters = 2000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  k_rate<NF, NX, CVT><<<256, threads>>>(d, 10, seed);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_rate<NF, NX, CVT><<<256, threads>>>(d, iters, seed);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double f16_equiv = (double)iters * (NF * 32);   // 16x16x32 f16 MFMAs per wave
  const double mx = (double)iters * (NX * 32);
  const double waves = 256.0 * threads / 64;
  const double flops = waves * (f16_equiv * 16 * 16 * 32 * 2 + mx * 16 * 16 * 128 * 2);
  printf("%-34s %d thr  %.3f ms  %.1f ns/round/wave  executed %.0f TFLOP/s (f16-only part %.0f)\n", name, threads, ms,
         ms * 1e6 / iters, flops / ms / 1e9, waves * f16_equiv * 16 * 16 * 32 * 2 / ms / 1e9);
  CK(hipFree(d));
  CK(hipFree(seed));
}

int main() {
  // ---- 1. layout under the hypothesis: lane l = (i = l & 15, g = l >> 4) holds K = 32 g + e, e = 0..31, nibble e of
  //         its 16 bytes (dword e / 8, bits 4 (e % 8)); A[i][k], B[k][j] with j = l & 15; D as the other 16x16 MFMAs.
  std::vector<int> ca(16 * 128), cb(16 * 128);
  for (auto& c : ca) c = rand() & 15;
  for (auto& c : cb) c = rand() & 15;
  std::vector<int> pa(64 * 4, 0), pb(64 * 4, 0);
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 32; ++e) {
      const int i = l & 15, g = l >> 4, k = 32 * g + e;
      pa[l * 4 + e / 8] |= ca[i * 128 + k] << (4 * (e % 8));
      pb[l * 4 + e / 8] |= cb[i * 128 + k] << (4 * (e % 8));
    }
  int *da, *db;
  float* dd;
  CK(hipMalloc(&da, 64 * 4 * 4));
  CK(hipMalloc(&db, 64 * 4 * 4));
  CK(hipMalloc(&dd, 64 * 4 * 4));
  CK(hipMemcpy(da, pa.data(), 64 * 4 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, pb.data(), 64 * 4 * 4, hipMemcpyHostToDevice));
  const int scales[][3] = {{127, 127, 0}, {128, 127, 0}, {127, 125, 0}, {0x7f807f7f, 0x7f7f7e7f, 1}, {(int)0x817f7f7fu, 0x7c7f7f7f, 2}};
  for (auto& sc : scales) {
    k_layout<<<1, 64>>>(da, db, dd, sc[0], sc[1], sc[2]);
    CK(hipDeviceSynchronize());
    float hd[256];
    CK(hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost));
    // expected scale factor: byte selected by opsel of the variant
    const int oa = sc[2] == 0 ? 0 : sc[2] == 1 ? 1 : 3, ob = sc[2] == 0 ? 0 : sc[2] == 1 ? 2 : 3;
    const int ea = ((unsigned)sc[0] >> (8 * oa)) & 255, eb = ((unsigned)sc[1] >> (8 * ob)) & 255;
    const double f = ldexp(1.0, ea - 127) * ldexp(1.0, eb - 127);
    double maxerr = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int col = l & 15, row = (l >> 4) * 4 + r;   // D[row][col] = sum_k A[row][k] B[k][col]
        double ref = 0;
        for (int k = 0; k < 128; ++k) ref += dec4(ca[row * 128 + k]) * dec4(cb[col * 128 + k]);
        maxerr = fmax(maxerr, fabs(hd[l * 4 + r] - ref * f));
      }
    printf("layout hypothesis, scale_a %08x scale_b %08x opsel variant %d (expected factor %g): max |D - ref| = %g  (D[0]=%g)\n",
           sc[0], sc[1], sc[2], f, maxerr, hd[0]);
  }
  // ---- 2. v_cvt_scalef32_pk_fp4_f16
  {
    const float vals[64 * 8 / 8][8] = {
        {0.f, 0.24f, 0.25f, 0.26f, 0.5f, 0.74f, 0.75f, 0.76f},
        {1.f, 1.24f, 1.25f, 1.26f, 1.5f, 1.74f, 1.75f, 1.76f},
        {2.f, 2.4f, 2.5f, 2.6f, 3.f, 3.4f, 3.5f, 3.6f},
        {4.f, 4.9f, 5.f, 5.1f, 6.f, 7.f, 100.f, 65504.f},
        {-0.24f, -0.26f, -1.25f, -2.5f, -5.f, -7.f, 1e-5f, -1e-5f}};
    std::vector<_Float16> hx(64 * 8, (_Float16)0.f);
    for (int l = 0; l < 5; ++l)
      for (int j = 0; j < 8; ++j) hx[l * 8 + j] = (_Float16)vals[l][j];
    _Float16* dx;
    unsigned* dox;
    CK(hipMalloc(&dx, 64 * 8 * 2));
    CK(hipMalloc(&dox, 64 * 4));
    CK(hipMemcpy(dx, hx.data(), 64 * 8 * 2, hipMemcpyHostToDevice));
    for (float scale : {1.f, 2.f, 0.5f}) {
      k_cvt<<<1, 64>>>(dx, dox, scale);
      CK(hipDeviceSynchronize());
      unsigned ho[64];
      CK(hipMemcpy(ho, dox, sizeof ho, hipMemcpyDeviceToHost));
      printf("cvt_scalef32_pk_fp4_f16, scale %g:\n", scale);
      for (int l = 0; l < 5; ++l) {
        printf("   word %08x :", ho[l]);
        for (int j = 0; j < 8; ++j) printf("  %g->%g", (float)hx[l * 8 + j], dec4((ho[l] >> (4 * j)) & 15));
        printf("\n");
      }
    }
  }
  // ---- 3. issue rate
  rate<4, 0, 0>("f16 only (4 x 32 per round)", 256);
  rate<4, 0, 0>("f16 only (4 x 32 per round)", 512);
  rate<0, 1, 0>("mx-fp4 only (32 per round)", 256);
  rate<0, 1, 0>("mx-fp4 only (32 per round)", 512);
  rate<4, 1, 0>("4 x f16 + 1 x mx", 256);
  rate<4, 1, 0>("4 x f16 + 1 x mx", 512);
  rate<4, 1, 1>("4 x f16 + 1 x mx + 32 cvt / step", 256);
  rate<4, 1, 1>("4 x f16 + 1 x mx + 32 cvt / step", 512);
  rate<8, 0, 0>("8 x f16 (= two-pass fp16x2)", 512);
  return 0;
}
