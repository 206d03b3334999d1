// Probe (GPU, standalone; not a product path): inner loop of a 256 x 256 x 32 workgroup tile run by FOUR waves (one
// per SIMD, 128 x 128 outputs each, the 64 accumulator fragments in AGPRs), software-pipelined inside each wave: the
// ds_reads of step j+1 and the 4-bit conversions of step j are issued in the shadow of the MFMAs of step j, one
// barrier per step, LDS-DMA three steps ahead.  Question: how close to the MFMA issue rate does this get compared with
// the two-wave-group ping-pong of tdnn_gemm_kernel_sk (2200 cycles per 512 x 128 x 32 step = 58 % of the pipe)?
// Answer (round 2, MI355X): not closer.  64 back-to-back MFMAs of one wave take 560-590 ns (0.73-0.77 of the 2.5 PF
// peak: the clock under MFMA load), + the 16 ds_reads 575-650 ns, + one barrier: free.  But with ONE wave per SIMD every
// other instruction costs issue time the MFMA pipe idles through: the LDS-DMA pieces of a step with their scalar
// address code +300-400 ns (not their latency or bandwidth: the same with L2-hot tiles and without the vmcnt wait; the
// scaffolding alone, DMA replaced by s_nop, +200 ns), the 32 quarter-rate v_cvt_scalef32_pk_fp4_f16 of a step +170-250
// ns, two taken branches +40 ns.  Full fp16mx step: 1340-1500 ns for 256 x 256 x 32 products, against ~1100 ns for the
// same products in the shipped kernel, whose second wave per SIMD absorbs exactly this non-MFMA issue work.  The
// variants (template bits) are kept as the record of that decomposition.
//   build: hipcc --offload-arch=gfx950 -O3 tools/probe_q.hip -o build/probe_q
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define AS3 __attribute__((address_space(3)))

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ void glds16_sbase(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory");
}
__device__ __forceinline__ void glds16_nom0(const void* sbase, unsigned voff) {
  asm volatile("global_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void mfma_f16(const s16x8& a, const s16x8& b, f32x4& c) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <int OA, int OB, class TA, class TB>
__device__ __forceinline__ void mfma_mx4(const TA& a, const TB& b, f32x4& c, int sa, int sb) {
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[%5,%6,0] op_sel_hi:[%7,%8,0] cbsz:4 blgp:4"
               : "+a"(c)
               : "v"(a), "v"(b), "v"(sa), "v"(sb), "n"(OA & 1), "n"(OB & 1), "n"(OA >> 1), "n"(OB >> 1));
}
template <int N>
__device__ __forceinline__ void wait_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

constexpr int XT = 272 * 64, WT = 256 * 64, W4T = 256 * 64;
constexpr int XB = 0, WB = 3 * XT, W4B = WB + 3 * WT, LDS_BYTES = W4B + 2 * W4T;

// V bit 0: ds_reads, bit 1: DMA + barriers, bit 2: conversions + MX MFMAs
template <int V>
__global__ __launch_bounds__(256) void probe_q(const uint16_t* x, const uint16_t* w, const uint8_t* w4, float* out, int ldx,
                                               int ldw, int ldw4, int nblk, int tiles, int m_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool RD = V & 1, DMA = V & 2, MX = V & 4, SPL = V & 8, BAR = V & 16, NOWAIT = V & 32, HOT = V & 64, M0ONCE = V & 128, NODMA = V & 256, UNCOND = V & 512, TAKEN = V & 1024, NOPS = V & 2048, NOPM = V & 4096, VMW = V & 8192;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave & 1, wave_n = wave >> 1;
  const int fr_i = lane & 15, fr_g = lane >> 4;
  const int ld_row = lane >> 2, ld_chunk = (lane & 3) ^ ((lane >> 3) & 3);
  const unsigned lds_base = (unsigned)(size_t)(AS3 char*)smem;
  const int xrow = wave_m * 128 + fr_i;
  const int x_rd = xrow * 64 + (fr_g ^ ((xrow >> 1) & 3)) * 16;
  const int wrow = wave_n * 128 + fr_i;
  const int w_rd = wrow * 64 + (fr_g ^ ((wrow >> 1) & 3)) * 16;

  f32x4 acc[8][8];
  struct Buf {
    s16x8 x[8], w[8];
  };
  Buf bA, bB;
  i32x4 x4[8];
  int ws_v[2] = {0x7f7f7f7f, 0x7f7f7f7f}, xs_b[2] = {0x7f7f7f7f, 0x7f7f7f7f};
  float xs_f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    xs_f[i] = 1.0f;
    x4[i] = i32x4{0, 0, 0, 0};
  }
  float total = 0.f;

  for (int t = 0; t < tiles; ++t) {
    // workgroups b and b + 8 (same XCD, neighbouring slots) take the two column tiles of the same row tile
    const int pair = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7);
    const int m0 = HOT ? (pair & 7) * 256 : ((pair * tiles + t) % m_tiles) * 256, n0 = ((blockIdx.x >> 3) & 1) * 256;
    const char* xbase = (const char*)(x + (long)(m0 + wave * 16) * ldx);
    const char* wbase = (const char*)(w + (long)(n0 + wave * 16) * ldw);
    const char* w4base = (const char*)(w4 + (long)(n0 + wave * 16) * ldw4);
    const unsigned xoff = (unsigned)(ld_row * ldx + ld_chunk * 8) * 2u;
    const unsigned woff = (unsigned)(ld_row * ldw + ld_chunk * 8) * 2u;
    const unsigned woff4 = (unsigned)(ld_row * ldw4 + ld_chunk * 16);
    const long x64 = ldx * 128, w64 = ldw * 128, w464 = ldw4 * 64;
    int istep = 0;
    const int ns = nblk * 4;
    // DMA of one step, cut into pieces that the stage spreads between its MFMAs (a burst of 8-12 of them at the top of
    // the stage backs up the memory pipe and the MFMA pipe idles behind the wave's blocked issue)
    const char *ixs = nullptr, *iws = nullptr, *iw4 = nullptr;
    unsigned isx = 0, isw = 0, is4 = 0;
    bool i_x = false, i_w4 = false, i_on = false;
    auto issue_begin = [&](bool on) __attribute__((always_inline)) {
      i_on = (on || UNCOND) && DMA;
      if (i_on) {
        const int slot = istep % 3;
        isx = __builtin_amdgcn_readfirstlane(lds_base + XB + slot * XT + wave * 1024);
        isw = __builtin_amdgcn_readfirstlane(lds_base + WB + slot * WT + wave * 1024);
        ixs = xbase + istep * 64;
        iws = wbase + istep * 64;
        i_x = !SPL || (istep % 3) == 0;
        i_w4 = MX && (istep & 3) == 1;
        if (UNCOND || i_w4) {
          is4 = __builtin_amdgcn_readfirstlane(lds_base + W4B + ((istep >> 2) & 1) * W4T + wave * 1024);
          iw4 = w4base + (istep >> 2) * 64;
        }
        ++istep;
        if constexpr (M0ONCE) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(isx) : "memory");
      }
    };
    auto issue_piece = [&](auto P, auto W4) __attribute__((always_inline)) {   // piece p of 8
      constexpr int p = decltype(P)::value;
      constexpr bool w4_static = decltype(W4)::value;
      if constexpr (DMA) {
        if (UNCOND || i_on) {
          if constexpr (NODMA) {
            asm volatile("s_nop 0" ::: "memory");
          } else if constexpr (M0ONCE) {
            if ((p & 1) == 0) glds16_nom0(ixs + (p >> 1) * x64, xoff);
            else glds16_nom0(iws + (p >> 1) * w64, woff);
          } else if ((p & 1) == 0) {
            if (i_x) glds16_sbase(ixs + (p >> 1) * x64, xoff, isx + (p >> 1) * 4096);
          } else {
            glds16_sbase(iws + (p >> 1) * w64, woff, isw + (p >> 1) * 4096);
            if (UNCOND ? w4_static : i_w4) glds16_sbase(iw4 + (p >> 1) * w464, woff4, is4 + (p >> 1) * 4096);
          }
        }
      }
    };
    auto issue = [&](auto W4) __attribute__((always_inline)) {
      issue_begin(true);
      static_for<0, 8>([&](auto P) { issue_piece(P, W4); });
    };
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue(std::false_type{});
    issue(std::integral_constant<bool, MX>{});
    issue(std::false_type{});
    if constexpr (DMA) {
      if constexpr (MX) wait_barrier<20>();   // steps 1 (8 + 4) and 2 (8) may still be in flight
      else wait_barrier<16>();
    }
    {
      const char* xs = smem + XB + x_rd;
      const char* ws = smem + WB + w_rd;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        bA.x[i] = *(const s16x8*)(xs + i * 1024);
        bA.w[i] = *(const s16x8*)(ws + i * 1024);
      }
    }
    int rslot = 1;   // ring slot of the step whose fragments are read next
    int rblk = 0;
    auto stage = [&](auto S, Buf& cur, Buf& nxt, const bool more, const bool two_ahead) __attribute__((always_inline)) {
      constexpr int s = decltype(S)::value;
      // step j+1 has landed everywhere, every wave holds its fragments of step j: slot j % 3 is free
      if constexpr (DMA) {
        if constexpr (NOWAIT) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (!two_ahead) wait_barrier<0>();
        else if (MX && s == 3) wait_barrier<12>();   // in flight: step j+2 = position 1 of the next block (x, w, w4)
        else wait_barrier<8>();
        issue_begin(more);
      } else if constexpr (BAR) {
        if constexpr (VMW) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if constexpr (TAKEN) {   // two taken branches per stage (the skipped blocks are laid out inline)
          if (__builtin_expect(tiles > 1000, 1)) asm volatile("s_nop 3\n\ts_nop 3" ::: "memory");
          asm volatile("s_nop 0" ::: "memory");
          if (__builtin_expect(tiles > 2000, 1)) asm volatile("s_nop 4\n\ts_nop 4" ::: "memory");
        }
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      const char* xs = smem + XB + rslot * XT + x_rd;
      const char* ws = smem + WB + rslot * WT + w_rd;
      const char* w4s = smem + W4B + (rblk & 1) * W4T + w_rd;
      static_for<0, 8>([&](auto P) {
        constexpr int p = decltype(P)::value;
        unsigned r = (unsigned)x4[p][s];
        const f16x8 h = __builtin_bit_cast(f16x8, cur.x[p]);
        mfma_f16(cur.w[p], cur.x[0], acc[p][0]);
        if constexpr (RD) nxt.w[p] = *(const s16x8*)(ws + p * 1024);
        mfma_f16(cur.w[p], cur.x[1], acc[p][1]);
        if constexpr (RD) nxt.x[p] = *(const s16x8*)(xs + p * 1024);
        mfma_f16(cur.w[p], cur.x[2], acc[p][2]);
        if constexpr (MX) {
          asm volatile("" : "+v"(r));
          r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, f16x2{h[0], h[1]}, xs_f[p], 0);
          asm volatile("" : "+v"(r));
        }
        mfma_f16(cur.w[p], cur.x[3], acc[p][3]);
        if constexpr (MX) {
          r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, f16x2{h[2], h[3]}, xs_f[p], 1);
          asm volatile("" : "+v"(r));
        }
        mfma_f16(cur.w[p], cur.x[4], acc[p][4]);
        if constexpr (MX) {
          r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, f16x2{h[4], h[5]}, xs_f[p], 2);
          asm volatile("" : "+v"(r));
        }
        mfma_f16(cur.w[p], cur.x[5], acc[p][5]);
        if constexpr (MX) {
          r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, f16x2{h[6], h[7]}, xs_f[p], 3);
          asm volatile("" : "+v"(r));
          x4[p][s] = (int)r;
        }
        mfma_f16(cur.w[p], cur.x[6], acc[p][6]);
        issue_piece(P, std::integral_constant<bool, (MX && s == 2)>{});
        if constexpr (NOPS) asm volatile("s_nop 0");
        if constexpr (NOPM) asm volatile("s_nop 0" ::: "memory");
        mfma_f16(cur.w[p], cur.x[7], acc[p][7]);
        // the block's 4-bit weight fragment p replaces the fp16 one, which has just had its last use
        if constexpr (MX && s == 3) cur.w[p] = *(const s16x8*)(w4s + p * 1024);
      });
      if constexpr (MX && s == 3) {
        static_for<0, 8>([&](auto P) {
          static_for<0, 8>([&](auto Q) {
            constexpr int p = decltype(P)::value, q = decltype(Q)::value;
            mfma_mx4<(p & 3), (q & 3)>(cur.w[p], x4[q], acc[p][q], ws_v[p >> 2], xs_b[q >> 2]);
          });
        });
        ++rblk;
      }
      rslot = rslot == 2 ? 0 : rslot + 1;
    };
#pragma nounroll
    for (int j = 0; j < ns; j += 4) {
      stage(std::integral_constant<int, 0>{}, bA, bB, j + 3 < ns, j + 2 < ns);
      stage(std::integral_constant<int, 1>{}, bB, bA, j + 4 < ns, j + 3 < ns);
      stage(std::integral_constant<int, 2>{}, bA, bB, j + 5 < ns, j + 4 < ns);
      stage(std::integral_constant<int, 3>{}, bB, bA, j + 6 < ns, j + 5 < ns);
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
      for (int q = 0; q < 8; ++q) total += acc[p][q][0] + acc[p][q][1] + acc[p][q][2] + acc[p][q][3];
  }
  out[blockIdx.x * 256 + tid] = total;
}

template <int V>
static void run(const char* name, const uint16_t* x, const uint16_t* w, const uint8_t* w4, float* out, int ldx, int ldw, int ldw4,
                int nblk, int tiles, int m_tiles) {
  CK(hipFuncSetAttribute((const void*)probe_q<V>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(probe_q<V>, dim3(256), dim3(256), LDS_BYTES, 0, x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double steps = (double)tiles * nblk * 4;
    const double flop = 256.0 * steps * 256 * 256 * 32 * 2;
    printf("%-28s %8.3f ms  %7.1f ns/step  %7.1f TFLOP/s (fp16 products)  = %.3f of 2.5 PF\n", name, ms, ms * 1e6 / steps,
           flop / ms * 1e-9, flop / ms * 1e-9 / 2500.0);
  }
}

int main() {
  const int M = 102400 + 512, K = 1536, N = 512;
  const int ldx = K, ldw = K, ldw4 = K / 2;
  std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K);
  std::vector<uint8_t> hw4((size_t)N * K / 2);
  srand(1);
  for (auto& v : hx) {
    _Float16 h = (_Float16)((rand() % 2001 - 1000) * 1e-3f);
    v = __builtin_bit_cast(uint16_t, h);
  }
  for (auto& v : hw) {
    _Float16 h = (_Float16)((rand() % 2001 - 1000) * 1e-4f);
    v = __builtin_bit_cast(uint16_t, h);
  }
  for (auto& v : hw4) v = (uint8_t)rand();
  uint16_t *x, *w;
  uint8_t* w4;
  float* out;
  CK(hipMalloc(&x, hx.size() * 2));
  CK(hipMalloc(&w, hw.size() * 2));
  CK(hipMalloc(&w4, hw4.size()));
  CK(hipMalloc(&out, 256 * 256 * 4));
  CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(w4, hw4.data(), hw4.size(), hipMemcpyHostToDevice));
  const int nblk = K / 128, tiles = 6, m_tiles = 400;
  run<0>("mfma only", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  run<1>("+ ds_read", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  run<17>("+ ds_read + barrier", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  run<17 + 1024>("+ ds_read + barrier + 2 taken br", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  run<17 + 2048>("+ ds_read + barrier + 8 nop", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  run<17 + 4096>("+ ds_read + barrier + 8 nop(mem)", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  run<17 + 8192>("+ ds_read + barrier w/ vmcnt", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  run<3>("+ dma + barrier", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  run<11>("+ dma(x every 3rd) + barrier", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  run<3 + 128>("+ dma, m0 set once (wrong dst)", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  run<3 + 256>("+ dma scaffolding, no dma", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  run<3 + 256 + 512>("+ scaffolding, no dma, no branches", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  run<3 + 512>("+ dma, no branches", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  run<7 + 512>("full, no branches", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  run<3 + 32>("+ dma, no vmcnt wait", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  run<3 + 64>("+ dma, L2-hot tiles", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  run<4>("mfma + cvt + mx (no lds)", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  run<7>("full (fp16mx)", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  run<15>("full, x every 3rd", x, w, w4, out, ldx, ldw, ldw4, nblk, tiles, m_tiles);
  return 0;
}
