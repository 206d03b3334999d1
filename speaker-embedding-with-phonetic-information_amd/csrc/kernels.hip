// Hand-written gfx950 (CDNA4 / MI355X) kernels for the x-vector / c-vector forward pass.
//
// What replaces what (the reference runs these through Kaldi's nnet3 on CPU; the arithmetic
// is upstream Kaldi):
//   tdnn_gemm_kernel   <- Descriptor Append(Offset(..)) gather + NaturalGradientAffineComponent
//                         ::Propagate + RectifiedLinearComponent + BatchNormComponent(test mode)
//                         [+ StatisticsExtractionComponent in the kEpiStats epilogue]
//                         graphs: egs/sre/v2/local/nnet3/xvector/run_xvector_new.sh:95-99
//   pool_finalise      <- StatisticsPoolingComponent (mean+stddev(0:1:1:10000), run_xvector_new.sh:106)
//   prep_input         <- the host->"CuMatrix" copy of one chunk of features
//
// Kernel design:
//   * 128x128x32 workgroup tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 fragments;
//   * both operands are K-contiguous ([rows][K] activations, [N][K] weights), staged with
//     global_load_lds_dwordx4 straight into LDS (no VGPR round trip), double buffered, one
//     workgroup barrier per K step;
//   * LDS rows are 64 B; the 16-byte chunk index is XOR-swizzled with the row (phys = chunk ^ ((row>>1)&3)) so
//     that every ds_read_b128 lane group touches 16 distinct 16-B slots for ANY starting row (the spliced layers
//     read the activation tile displaced by their time offsets; brute-forced over all 16 displacements).  Because
//     LDS-DMA writes are lane-linear the swizzle is applied to the per-lane *global source* address;
//   * the splice (Append of time offsets) is a row shift of the activation tile per K segment;
//   * split precision (hi, lo planes): three MFMAs per product give fp32-grade results (bf16x3, fp16x3); fp16
//     activations x split-fp16 weights (fp16x2, two MFMAs) remove the weight rounding error, which the statistics
//     pooling cannot average out, and are what long chunks run by default;
//   * three variants of the same pipeline: tdnn_gemm_kernel (128x128, small / odd launches and split-K slices),
//     tdnn_gemm_kernel_v2 (256x128, ping-pong wave groups), tdnn_gemm_kernel_sk (persistent stream-K grid, 512x128
//     tiles, two-pass mode) - bit-identical results, chosen per launch;
//   * kEpiAct / kEpiF32 feed the weight tile as the MFMA *A* operand, with the weight rows of a
//     wave permuted at load time, so each lane ends up owning 16 contiguous output columns of
//     one frame (32-byte stores per plane); kEpiStats feeds activations as A so the 16-row
//     reduction is 3 in-register adds + 2 cross-lane adds.
#include "kernels.h"
#include "knobs.h"

#include <hip/hip_ext.h>

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

namespace xv {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define XV_AS1 __attribute__((address_space(1)))
#define XV_AS3 __attribute__((address_space(3)))

// Per-launch timing without extra packets on the stream: when the engine has armed a (start, stop) event pair
// (set_launch_events), the kernels of the next launch_* call are dispatched with hipExtLaunchKernelGGL, which stamps
// the events from the dispatch itself - the first kernel takes the start event, every kernel re-records the stop
// event (the last one wins, so a GEMM + split-K reduction pair is timed as one).
static thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
void set_launch_events(hipEvent_t start, hipEvent_t stop) {
  g_ev_start = start;
  g_ev_stop = stop;
}
static thread_local char g_last_kernel[96] = "";
const char* last_gemm_kernel() { return g_last_kernel; }
static const char* prec_name(int p) {
  static const char* n[] = {"bf16x3", "bf16", "fp16", "fp16x3", "fp16x2", "auto", "fp16mx", "fp16mx2", "fp16x3e", "fp16mxe"};
  return (p >= 0 && p <= 9) ? n[p] : "?";
}
static const char* epi_name(int e) { return e == kEpiAct ? "act" : e == kEpiF32 ? "f32" : e == kEpiStats ? "stats" : "splitk"; }
static void note_kernel(const char* variant, int prec, int epi, int mf) {
  if (mf) snprintf(g_last_kernel, sizeof g_last_kernel, "tdnn_gemm_kernel%s<%s,%s,%d>", variant, prec_name(prec), epi_name(epi), mf);
  else snprintf(g_last_kernel, sizeof g_last_kernel, "tdnn_gemm_kernel%s<%s,%s>", variant, prec_name(prec), epi_name(epi));
}
#define XV_LAUNCH(kern, grid, block, lds, stream, ...)                                           \
  do {                                                                                           \
    hipEvent_t _st = g_ev_start, _sp = g_ev_stop;                                                \
    g_ev_start = nullptr;                                                                        \
    if (_st || _sp) hipExtLaunchKernelGGL(kern, grid, block, lds, stream, _st, _sp, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                        \
  } while (0)

__device__ __forceinline__ void glds16(const void* g, void* l) {
  // 16 bytes per lane, LDS destination = wave-uniform base + lane*16
  __builtin_amdgcn_global_load_lds((const XV_AS1 void*)g, (XV_AS3 void*)l, 16, 0, 0);
}

template <bool F16>
__device__ __forceinline__ f32x4 mfma16(s16x8 a, s16x8 b, f32x4 c) {
  if constexpr (F16) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
}

template <bool F16>
__device__ __forceinline__ uint16_t to16(float x) {
  if constexpr (F16) {
    _Float16 h = (_Float16)x;
    return __builtin_bit_cast(uint16_t, h);
  } else {
    __bf16 h = (__bf16)x;
    return __builtin_bit_cast(uint16_t, h);
  }
}

template <bool F16>
__device__ __forceinline__ float from16(uint16_t u) {
  if constexpr (F16) {
    return (float)__builtin_bit_cast(_Float16, u);
  } else {
    return __builtin_bit_cast(float, ((unsigned int)u) << 16);
  }
}

// ---------------------------------------------------------------------------------------------
// Schedule fuzzing (verification build only: `make fuzz` -> -DXVEC_SCHED_FUZZ, a second library next to the product one;
// tools/fuzz_schedule.sh).  The GEMM kernels order their LDS traffic with counted waits and one barrier per phase, and their two
// wave groups run a barrier interval apart: a missing wait shows only when one wave gets far enough ahead of another, which
// under load happens once in thousands of launches (both races of rounds 4 / 5 were found that way).  The fuzz build makes
// it happen all the time: every XV_FUZZ() point sleeps the WAVE for a pseudo-random time - mostly 0-3 x 64 cycles, one time in
// four up to 63 x 64 cycles (a whole K tile) - from a per-wave generator seeded with the clock at kernel start, so the waves
// of a workgroup drift against each other and against the DMA engine differently in every launch.  Every output element is
// summed in one fixed order, so ANY result that differs from the emulation / from the previous launch is a hazard, not noise.
// Scalar instructions only (s_sleep, s_mul, s_add): no wait counter is touched, no vector register used.
#ifdef XVEC_SCHED_FUZZ
__device__ __forceinline__ unsigned sched_fuzz_seed() {
  const unsigned t = (unsigned)__builtin_amdgcn_s_memtime();
  return __builtin_amdgcn_readfirstlane((t * 2654435761u) ^ ((threadIdx.x >> 6) * 0x9E3779B9u) ^ (blockIdx.x * 0x85EBCA6Bu));
}
__device__ __forceinline__ void sched_fuzz(unsigned& st) {
  st = __builtin_amdgcn_readfirstlane(st * 1664525u + 1013904223u);
  const unsigned r = st >> 24;
  unsigned n = r < 192u ? (r & 3u) : (r - 192u);
  __builtin_amdgcn_sched_barrier(0);
  for (; n; --n) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_sched_barrier(0);
}
#define XV_FUZZ_INIT() unsigned xv_fuzz_state = sched_fuzz_seed()
#define XV_FUZZ() sched_fuzz(xv_fuzz_state)
#else
#define XV_FUZZ_INIT() do { } while (0)
#define XV_FUZZ() do { } while (0)
#endif

// Kaldi's ApplyFloor: x < floor ? floor : x, i.e. a NaN stays a NaN (fmaxf / v_max_f32 would return the floor).  gfx950's
// v_maximum3_f32 is the IEEE-754-2019 maximum - NaN-propagating - in one instruction instead of compare + select; the only
// difference to the comparison is the sign of an exact zero (maximum(-0, +0) = +0), which no later sum or product can see.
// A floor of -inf = no ReLU.
__device__ __forceinline__ float apply_floor(float x, float floor) {
  float r;
  asm("v_maximum3_f32 %0, %1, %2, %2" : "=v"(r) : "v"(x), "v"(floor));
  return r;
}
// Two fp32 values -> one dword of two 16-bit floats, round to nearest even (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32).
template <bool F16>
__device__ __forceinline__ unsigned int to16x2(f32x2 x) {
  if constexpr (F16) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(x, h2));
  } else {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(x, b2));
  }
}

// Weight-row permutation of the "weights as MFMA A operand" orientation.  LDS row rho = p*16 + g*4 + r of a wave's
// 64-row slice (p = 16-row fragment, g = lane>>4, r = accumulator register) holds weight row
//   n = (p>>1)*32 + g*8 + (p&1)*4 + r,
// so the lane with lane>>4 == g ends up with output columns g*8..g*8+7 of each 32-column half: 16 contiguous bytes
// per plane per store, and the four lane groups of a store instruction cover 64 contiguous bytes of a row.
__device__ __forceinline__ int swap_fields(int rho) {
  const int p = (rho >> 4) & 3, g = (rho >> 2) & 3, r = rho & 3;
  return ((p >> 1) << 5) | (g << 3) | ((p & 1) << 2) | r;
}

// Cross-lane adds of the statistics epilogue on the gfx950 lane-swap instructions (plain VALU) instead of
// ds_bpermute (LDS crossbar, ~100 cycles each).  v_permlane16_swap a, b exchanges the odd 16-lane rows of a with the
// even rows of b: a = [a0 b0 a2 b2], b = [a1 b1 a3 b3] (rows of 16 lanes); v_permlane32_swap does the same with the
// 32-lane halves.  (Inline asm: the builtins of this hipcc mis-assign the second result register.)
// sum4_rows_scatter: the 16-row sums of FOUR values at once (the fragments p = 0..3 of one column; a lane holds the
// sum over its four rows, the 16 rows of a fragment are spread over the four 16-lane rows of the wave).  One swap and
// one add per butterfly level and PAIR of values: after the first level the even rows hold the pair sums of the
// first value of a pair and the odd rows those of the second; after the second, 16-lane row g holds the total of
// value g - the distribution the epilogue stores (lane group g writes fragment g).  Per element the additions are
// (r0 + r1) + (r2 + r3) with the lower row first, as in the xor-16 / xor-32 butterfly this replaces: same bits.
__device__ __forceinline__ float sum4_rows_scatter(float v0, float v1, float v2, float v3) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v0), "+v"(v1));
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v2), "+v"(v3));
  float u = v0 + v1, w = v2 + v3;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(w));
  return u + w;
}

// Maximum over the 64 lanes of a wave, in every lane: four DPP steps inside the 16-lane rows, then the lane-swap
// instructions across rows.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));   // row_half_mirror
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));   // row_mirror
  unsigned a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  v = max(a, b);
  a = v;
  b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return max(a, b);
}

constexpr int kTileBytes = kBM * kBK * 2;  // one 128x32 16-bit tile = 8 KiB

// ---- kPrecFp16Mx helpers ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// 8 fp16 values -> 8 e2m1 codes (x / scale, round to nearest even, saturating at +-6), element j in nibble j.
// `old` only provides the destination register (every byte is rewritten).
__device__ __forceinline__ int cvt8_fp4(s16x8 x, float scale, int old) {
  const f16x8 h = __builtin_bit_cast(f16x8, x);
  unsigned r = (unsigned)old;
  r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, f16x2{h[0], h[1]}, scale, 0);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, f16x2{h[2], h[3]}, scale, 1);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, f16x2{h[4], h[5]}, scale, 2);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp4_f16(r, f16x2{h[6], h[7]}, scale, 3);
  return (int)r;
}
__device__ __forceinline__ i32x8 w4_frag(s16x8 w) {
  const u32x4 u = __builtin_bit_cast(u32x4, w);
  return i32x8{(int)u[0], (int)u[1], (int)u[2], (int)u[3], 0, 0, 0, 0};
}
__device__ __forceinline__ i32x8 x4_frag(const int (&x)[4]) { return i32x8{x[0], x[1], x[2], x[3], 0, 0, 0, 0}; }
// D = A . B (16x16x128, both operands e2m1) * 2^(byte OA of sa - 127) * 2^(byte OB of sb - 127) + C
template <int OA, int OB>
__device__ __forceinline__ f32x4 mfma_mx4(i32x8 a, i32x8 b, f32x4 c, int sa, int sb) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 4, OA, sa, OB, sb);
}
// The same and the fp16 MFMA with the accumulator tied in place (see the stream-K kernel for why).  A and B are any
// 128-bit register tuples (fp16 fragments or e2m1 fragments).  Byte selector of a scale word = op_sel bit + 2 *
// op_sel_hi bit of the operand (what hipcc emits for the builtin above).
typedef __attribute__((ext_vector_type(4))) int i32x4;
template <int OA, int OB, class TA, class TB>
__device__ __forceinline__ void mfma_mx4_inplace(const TA& a, const TB& b, f32x4& c, int sa, int sb) {
  static_assert(sizeof(TA) == 16 && sizeof(TB) == 16, "e2m1 16x16x128 operands are four dwords per lane");
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[%5,%6,0] op_sel_hi:[%7,%8,0] cbsz:4 blgp:4"
      : "+v"(c)
      : "v"(a), "v"(b), "v"(sa), "v"(sb), "n"(OA & 1), "n"(OB & 1), "n"(OA >> 1), "n"(OB >> 1));
}
__device__ __forceinline__ void mfma16_f16_inplace(s16x8 a, s16x8 b, f32x4& c) {
  asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// The eight [first, last) bytes of a 64-row block's four 16-row groups (wave-uniform address) through the scalar cache.
// As the vector load hipcc makes of a plain read (it cannot prove the table read-only), the value was waited for with
// vmcnt(0) right where it was requested - and that counter also holds every LDS-DMA / prefetch load the kernel has issued
// for its NEXT part or unit to hide behind this epilogue: the epilogue began with a full memory latency.
__device__ __forceinline__ void load_group_ranges(const uint8_t* p, unsigned (&rng)[2]) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  u32x2 v;
  asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  rng[0] = v[0];
  rng[1] = v[1];
}

// Per-lane epilogue parameters (bias / scale / offset, and for the statistics epilogue the valid-row table), fetched
// by epilogue_prefetch: before the K loop where the registers are affordable (see variant 2), else right before use.
struct EpiRegs {
  float bs[16], sc[16], of[16];   // act / f32: index p*4 + r;  stats: index q (one column per 16-column fragment)
  int first[4], last[4];          // stats: valid rows [first, last) of the four 16-row fragments p
  unsigned rng[2];                // planes epilogue with group maxima: the same for the four groups q, bytes first / last of
                                  // group q in bits 16 (q & 1) .. of word q >> 1 (one 8-byte load, two registers)
};

template <int EPI>
__device__ __forceinline__ void epilogue_prefetch(const GemmArgs& a, const int mbase, const int nbase, const int lane,
                                                  EpiRegs& e) {
  const int fr_i = lane & 15;
  const int fr_g = lane >> 4;
  if constexpr (EPI == kEpiAct || EPI == kEpiF32) {
    // lane owns frames q*16 + fr_i (q = 0..3) x two groups of 8 contiguous columns: nbase + h*32 + fr_g*8 + (0..7),
    // h = p>>1, position inside the group (p&1)*4 + r
    const int ncol = nbase + fr_g * 8;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int c = ncol + (p >> 1) * 32 + (p & 1) * 4;
      const f32x4 b4 = *(const f32x4*)(a.bias + c);
      f32x4 s4 = {1.f, 1.f, 1.f, 1.f}, o4 = {0.f, 0.f, 0.f, 0.f};
      if (a.bn) {
        s4 = *(const f32x4*)(a.scale + c);
        o4 = *(const f32x4*)(a.offset + c);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        e.bs[p * 4 + r] = b4[r];
        e.sc[p * 4 + r] = s4[r];
        e.of[p * 4 + r] = o4[r];
      }
    }
    if (EPI == kEpiAct && a.gmax_out) {
      // rows of the four 16-row groups q that count for the group maxima: [grp][first, last) bytes, 8 contiguous ones
      // (mbase is a multiple of 64 rows, the table 8-byte aligned)
      e.rng[0] = e.rng[1] = 0x10001000u;
      if (a.out_range) load_group_ranges((const uint8_t*)a.out_range + 2 * (mbase >> 4), e.rng);
    }
  } else if constexpr (EPI == kEpiStats) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = nbase + q * 16 + fr_i;
      e.bs[q] = a.bias[col];
      e.sc[q] = a.bn ? a.scale[col] : 1.f;
      e.of[q] = a.bn ? a.offset[col] : 0.f;
    }
    // valid rows of the block's four 16-row groups: 8 contiguous bytes at a wave-uniform, 8-byte aligned address (mbase is
    // a multiple of 64) -> one scalar load (see load_group_ranges: as byte loads they queued behind the next part's LDS-DMA)
    unsigned gr[2];
    load_group_ranges((const uint8_t*)a.grp_range + 2 * (mbase >> 4), gr);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      e.first[p] = (int)(int8_t)(gr[p >> 1] >> (16 * (p & 1)));
      e.last[p] = (int)(int8_t)(gr[p >> 1] >> (16 * (p & 1) + 8));
    }
  }
}

// The same from a copy of the tile's parameters in LDS (stream-K kernel): par = bias[128] scale[128] offset[128] of the
// workgroup tile's columns, ncol0 = first column of the 64 x 64 block inside the tile.  (Fetched from global memory
// right before use - the stream-K kernel has no registers to hold them through the K loop - every block's epilogue
// started with a full load latency behind the next part's DMA: -6 % on the whole forward pass.)
template <int EPI, bool LAZY = false>
__device__ __forceinline__ void epilogue_prefetch_lds(const GemmArgs& a, const float* par, const int mbase, const int ncol0,
                                                      const int lane, EpiRegs& e) {
  const int fr_i = lane & 15;
  const int fr_g = lane >> 4;
  if constexpr (EPI == kEpiAct || EPI == kEpiF32) {
    const int ncol = ncol0 + fr_g * 8;
#pragma unroll
    for (int p = 0; p < (LAZY ? 0 : 4); ++p) {
      const int c = ncol + (p >> 1) * 32 + (p & 1) * 4;
      const f32x4 b4 = *(const f32x4*)(par + c);
      f32x4 s4 = {1.f, 1.f, 1.f, 1.f}, o4 = {0.f, 0.f, 0.f, 0.f};
      if (a.bn) {
        s4 = *(const f32x4*)(par + 128 + c);
        o4 = *(const f32x4*)(par + 256 + c);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        e.bs[p * 4 + r] = b4[r];
        e.sc[p * 4 + r] = s4[r];
        e.of[p * 4 + r] = o4[r];
      }
    }
    if (EPI == kEpiAct && a.gmax_out) {
      // rows of the four 16-row groups q that count for the group maxima: [grp][first, last) bytes, 8 contiguous ones
      // (mbase is a multiple of 64 rows, the table 8-byte aligned)
      e.rng[0] = e.rng[1] = 0x10001000u;
      if (a.out_range) load_group_ranges((const uint8_t*)a.out_range + 2 * (mbase >> 4), e.rng);
    }
  } else if constexpr (EPI == kEpiStats) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = ncol0 + q * 16 + fr_i;
      e.bs[q] = par[c];
      e.sc[q] = a.bn ? par[128 + c] : 1.f;
      e.of[q] = a.bn ? par[256 + c] : 0.f;
    }
    // valid rows of the block's four 16-row groups: 8 contiguous bytes at a wave-uniform, 8-byte aligned address (mbase is
    // a multiple of 64) -> one scalar load (see load_group_ranges: as byte loads they queued behind the next part's LDS-DMA)
    unsigned gr[2];
    load_group_ranges((const uint8_t*)a.grp_range + 2 * (mbase >> 4), gr);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      e.first[p] = (int)(int8_t)(gr[p >> 1] >> (16 * (p & 1)));
      e.last[p] = (int)(int8_t)(gr[p >> 1] >> (16 * (p & 1) + 8));
    }
  }
}

// Shared epilogue of the GEMM kernels.  acc[p][q] is the 16x16 fragment (P-tile fragment p) x (Q-tile fragment q)
// of one wave's 64x64 tile whose first frame is mbase and first output column nbase.
// gm / gm_phase (planes epilogue with a.gmax_out): lane-local maxima of |y| per 16-row group q carried between calls that
// cover the same rows - phase 0: this call stands alone; 1: first of two (accumulate only); 2: second (accumulate, then
// reduce over the wave and publish).
// LAZY (planes / f32 epilogues of the stream-K kernel): bias / scale / offset are not taken from e but read from the copy
// of the tile's parameters in LDS (par, pcol = first column of the block inside the tile) where they are used, four
// columns at a time, again for every 16-row group: 48 registers less next to the 128 accumulators of a 64 x 128 wave tile.
template <int PREC, int EPI, bool LAZY = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x4 (&acc)[4][4], const int mbase, const int nbase,
                                              const int lane, const EpiRegs& e, float (&gm)[4], const int gm_phase,
                                              const float* par = nullptr, const int pcol = 0) {
  constexpr bool SPLIT = PrecXPlanes(PREC) == 2;   // activations carry a residual plane
  constexpr bool F16 = PrecF16(PREC);
  const int fr_i = lane & 15;
  const int fr_g = lane >> 4;
  // The per-element work is branch-free: no ReLU = a floor of -inf, no BatchNorm = scale 1 / offset 0 (set by
  // epilogue_prefetch).  With the flags tested per element the statistics epilogue of a 512 x 128 tile was ~3500
  // instructions per wave, 15 us, and tdnn5 was bound by it rather than by its MFMAs.
  const float relu_floor = a.relu ? 0.f : -__builtin_inff();
  // ---- epilogues ---------------------------------------------------------------------------
  if constexpr (EPI == kEpiSplitK) {
    // raw accumulators of this K slice (bias / ReLU / BatchNorm are applied by splitk_reduce_kernel)
    float* ws = a.splitk_ws + (long)blockIdx.y * ((long)a.m_tiles * kBM) * ((long)a.n_tiles * kBN);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float* dst = ws + (long)(mbase + q * 16 + fr_i) * ((long)a.n_tiles * kBN) + nbase + fr_g * 8;
#pragma unroll
      for (int p = 0; p < 4; ++p) *(f32x4*)(dst + (p >> 1) * 32 + (p & 1) * 4) = acc[p][q];
    }
  } else if constexpr (EPI == kEpiAct || EPI == kEpiF32) {
    // lane owns frames q*16 + fr_i (q = 0..3) x two groups of 8 contiguous columns: nbase + h*32 + fr_g*8 + (0..7),
    // h = p>>1, position inside the group (p&1)*4 + r
    const int ncol = nbase + fr_g * 8;
    // Plane stores as (wave-uniform base in SGPRs) + (32-bit byte offset per lane): base and row step are read from the kernel
    // arguments ONCE.  Written as `a.out_hi + (long)row * a.ldo + col` per store, hipcc re-loaded a.out_hi from the kernarg
    // segment in front of every one of the 32 stores of a wave tile (s_load_dwordx2 + s_waitcnt lgkmcnt(0): a scalar-cache round
    // trip with the matrix pipe idle) and formed every address with a 64-bit multiply-add and two 64-bit adds (round 5, from the ISA).
    // A plane is far below 4 GB (131072 rows x 4096 columns x 2 bytes at most).
    XV_AS1 char* hi_base = (XV_AS1 char*)a.out_hi;   // (global address space: a generic pointer would make these flat stores)
    XV_AS1 char* lo_base = (XV_AS1 char*)a.out_lo;
    XV_AS1 char* lo4_base = (XV_AS1 char*)a.out_lo4;
    unsigned row_step_b = (unsigned)a.ldo * 32u;                                   // 16 rows, bytes
    unsigned off0_b = (unsigned)(((mbase + fr_i) * a.ldo + ncol) * 2);             // this lane's first row, first column group
    asm volatile("" : "+s"(hi_base), "+s"(lo_base), "+s"(lo4_base), "+s"(row_step_b));   // values, not re-loadable expressions
    if constexpr (LAZY && !PrecEmitsLo4(PREC)) {
      // Parameters from LDS, column pair outermost: the 24 parameter words of a lane's 8 contiguous columns are read once
      // and serve the four 16-row groups (with the rows outermost they were read again for every group: 96 LDS reads and
      // as many round trips per 64 x 128 wave tile, a third of the epilogue's time).  Same arithmetic per element, same
      // bytes per store instruction.
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        f32x4 b4[2], s4[2], o4[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = pcol + fr_g * 8 + pp * 32 + j * 4;
          b4[j] = *(const f32x4*)(par + c);
          s4[j] = *(const f32x4*)(par + 128 + c);
          o4[j] = *(const f32x4*)(par + 256 + c);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = mbase + q * 16 + fr_i;
          f32x2 y2[4];
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              f32x2 z = f32x2{acc[pp * 2 + j][q][2 * k], acc[pp * 2 + j][q][2 * k + 1]} + f32x2{b4[j][2 * k], b4[j][2 * k + 1]};
              z = f32x2{apply_floor(z[0], relu_floor), apply_floor(z[1], relu_floor)};
              y2[j * 2 + k] = __builtin_elementwise_fma(z, f32x2{s4[j][2 * k], s4[j][2 * k + 1]}, f32x2{o4[j][2 * k], o4[j][2 * k + 1]});
            }
          if constexpr (EPI == kEpiF32) {
            if (row < a.m_valid) {
              float* dst = a.out_f32 + (long)row * a.ldf + ncol + pp * 32;
              *(f32x4*)(dst) = f32x4{y2[0][0], y2[0][1], y2[1][0], y2[1][1]};
              *(f32x4*)(dst + 4) = f32x4{y2[2][0], y2[2][1], y2[3][0], y2[3][1]};
            }
          } else {
            unsigned int hw[4], lw[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              hw[v] = to16x2<F16>(y2[v]);
              if constexpr (SPLIT) {
                const f32x2 back = {from16<F16>((uint16_t)hw[v]), from16<F16>((uint16_t)(hw[v] >> 16))};
                lw[v] = to16x2<F16>(y2[v] - back);
              }
            }
            if (a.gmax_out) {
              float ymax = gm[q];
#pragma unroll
              for (int v = 0; v < 4; ++v) ymax = fmaxf(fmaxf(ymax, fabsf(y2[v][0])), fabsf(y2[v][1]));
              gm[q] = ymax;
            }
            const unsigned ob = off0_b + (unsigned)q * row_step_b + (unsigned)pp * 64u;
            *(XV_AS1 u32x4*)(hi_base + ob) = u32x4{hw[0], hw[1], hw[2], hw[3]};
            if constexpr (SPLIT) *(XV_AS1 u32x4*)(lo_base + ob) = u32x4{lw[0], lw[1], lw[2], lw[3]};
          }
        }
      }
      if constexpr (EPI == kEpiAct) {
        if (a.gmax_out && gm_phase != 1) {
          // max |y| of each 16-row group -> one atomic max per wave (see the general path below)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int first_q = (int)((e.rng[q >> 1] >> (16 * (q & 1))) & 255u), last_q = (int)((e.rng[q >> 1] >> (16 * (q & 1) + 8)) & 255u);
            float m = gm[q];
            if (fr_i < first_q || fr_i >= last_q) m = 0.f;
            const unsigned u = wave_max_u32(__builtin_bit_cast(unsigned, m));
            if (lane == 0)
              (void)__hip_atomic_fetch_max(a.gmax_out + ((mbase + q * 16) >> 4), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      return;
    }
    unsigned e4 = 0u;   // PrecEmitsLo4: scale byte of row q * 16 + fr_i in byte q (the same in the row's four lanes)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = mbase + q * 16 + fr_i;
      // y[p * 4 + r] as pairs y2[p * 2 + j] = (r = 2j, 2j + 1): bias and BatchNorm on both values of a pair at once
      // (v_pk_add_f32 / v_pk_fma_f32; the accumulators, the parameter quads and the converted words are register pairs in
      // exactly this order - left to itself hipcc paired the values across fragments and spent two moves and one
      // shift / mask / or per element on getting them there and back)
      f32x2 y2[8];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        f32x4 b4, s4 = {1.f, 1.f, 1.f, 1.f}, o4 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (LAZY) {
          int c = pcol + fr_g * 8 + (p >> 1) * 32 + (p & 1) * 4;
          asm volatile("" : "+v"(c));   // keeps the reads inside the loop over q (hoisted they are 48 live registers again)
          // no branch on a.bn here: whoever fills par stores scale 1 / offset 0 without BatchNorm.  (With the branch every
          // fragment's three reads sat in a basic block of their own, each ending in a full LDS round trip: 16 serialised
          // waits per 64 x 64 block.)
          b4 = *(const f32x4*)(par + c);
          s4 = *(const f32x4*)(par + 128 + c);
          o4 = *(const f32x4*)(par + 256 + c);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            b4[r] = e.bs[p * 4 + r];
            s4[r] = e.sc[p * 4 + r];
            o4[r] = e.of[p * 4 + r];
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x2 z = f32x2{acc[p][q][2 * j], acc[p][q][2 * j + 1]} + f32x2{b4[2 * j], b4[2 * j + 1]};
          z = f32x2{apply_floor(z[0], relu_floor), apply_floor(z[1], relu_floor)};
          // one rounding per value (fma); scale 1 / offset 0 without BatchNorm
          y2[p * 2 + j] = __builtin_elementwise_fma(z, f32x2{s4[2 * j], s4[2 * j + 1]}, f32x2{o4[2 * j], o4[2 * j + 1]});
        }
      }
      float y[16];
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        y[2 * v] = y2[v][0];
        y[2 * v + 1] = y2[v][1];
      }
      if constexpr (EPI == kEpiF32) {
        if (row < a.m_valid) {
          float* dst = a.out_f32 + (long)row * a.ldf + ncol;
#pragma unroll
          for (int p = 0; p < 4; ++p)
            *(f32x4*)(dst + (p >> 1) * 32 + (p & 1) * 4) = f32x4{y[p * 4], y[p * 4 + 1], y[p * 4 + 2], y[p * 4 + 3]};
        }
      } else {
        unsigned int hw[8], lw[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) {
          hw[v] = to16x2<F16>(y2[v]);
          if constexpr (SPLIT) {
            const f32x2 back = {from16<F16>((uint16_t)hw[v]), from16<F16>((uint16_t)(hw[v] >> 16))};
            lw[v] = to16x2<F16>(y2[v] - back);
          }
        }
        float ymax = 0.f;   // max |y| of this lane's 16 values (its share of the row's 64 columns)
        if (PrecEmitsLo4(PREC) || a.gmax_out) {
#pragma unroll
          for (int v = 0; v < 16; ++v) ymax = fmaxf(ymax, fabsf(y[v]));
        }
        if (a.gmax_out) {
          // max |y| of this 16-row group (a NaN is skipped by fmaxf, the main product carries it anyway; rows that are
          // not computable frames of the layer do not count) -> one atomic max per wave: non-negative floats order
          // like their bit patterns
          float m = fmaxf(gm[q], ymax);
          gm[q] = m;
          if (gm_phase != 1) {
            const int first_q = (int)((e.rng[q >> 1] >> (16 * (q & 1))) & 255u), last_q = (int)((e.rng[q >> 1] >> (16 * (q & 1) + 8)) & 255u);
            if (fr_i < first_q || fr_i >= last_q) m = 0.f;
            const unsigned u = wave_max_u32(__builtin_bit_cast(unsigned, m));
            if (lane == 0)
              (void)__hip_atomic_fetch_max(a.gmax_out + ((mbase + q * 16) >> 4), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        // y[0..7] = columns ncol..ncol+7, y[8..15] = columns ncol+32..ncol+39
        // (the run-time tests of out_hi / out_lo4 are always true; as branches they keep hipcc from interleaving the stores,
        // the group-maximum code and the residual block, which spilled 9-16 registers in the 512 x 128 stream-K kernel)
        const unsigned ob = off0_b + (unsigned)q * row_step_b;   // (base + 32-bit offset: see the top of this branch)
        if (!PrecEmitsLo4(PREC) || a.out_hi) {
          *(XV_AS1 u32x4*)(hi_base + ob) = u32x4{hw[0], hw[1], hw[2], hw[3]};
          *(XV_AS1 u32x4*)(hi_base + ob + 64u) = u32x4{hw[4], hw[5], hw[6], hw[7]};
        }
        if (PrecEmitsLo4(PREC) && a.out_lo4) {
          // what the fp16 rounding dropped, y - fp16(y), as e2m1 with one power-of-two scale per row and 64-column block
          // (this wave's 64 columns of the row: the four lanes fr_g of the row hold them).  The residual of a value in
          // [2^E, 2^(E+1)) is at most 2^(E-11); with E the exponent of the block's largest |y| the scale 2^(E-13) maps
          // every residual into [-4, 4]: no saturation, the largest ones in the top binades of the grid.  (A scale from
          // the residuals themselves costs a second pass over them and buys nothing: their maximum is almost always in
          // the same binade.)  Below the fp16 normal range the residual is bounded by 2^-25.
          float m = ymax;
          {
            float ma = m, mb = m;
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));
            m = fmaxf(ma, mb);
            ma = m;
            mb = m;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ma), "+v"(mb));
            m = fmaxf(ma, mb);
          }
          unsigned e = (__builtin_bit_cast(unsigned, m) >> 23) & 255u;
          e = e < 113u ? 100u : (e > 254u ? 241u : e - 13u);
          const float sc4 = __builtin_bit_cast(float, e << 23);
          unsigned c0 = 0u, c1 = 0u;
          static_for<0, 4>([&](auto V) {
            constexpr int v = decltype(V)::value;
            float r0, r1, r2, r3;
            asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hw[v]), "v"(y[2 * v]));
            asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hw[v]), "v"(y[2 * v + 1]));
            asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(hw[4 + v]), "v"(y[8 + 2 * v]));
            asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(hw[4 + v]), "v"(y[8 + 2 * v + 1]));
            c0 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(c0, r0, r1, sc4, v);
            c1 = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(c1, r2, r3, sc4, v);
          });
          // (4 bits per value: a quarter of the fp16 plane's byte offsets)
          *(XV_AS1 unsigned*)(lo4_base + (ob >> 2)) = c0;
          *(XV_AS1 unsigned*)(lo4_base + (ob >> 2) + 16u) = c1;
          e4 |= e << (8 * q);
        } else if constexpr (SPLIT) {
          *(XV_AS1 u32x4*)(lo_base + ob) = u32x4{lw[0], lw[1], lw[2], lw[3]};
          *(XV_AS1 u32x4*)(lo_base + ob + 64u) = u32x4{lw[4], lw[5], lw[6], lw[7]};
        }
      }
    }
    if constexpr (EPI == kEpiAct && PrecEmitsLo4(PREC)) {
      // the scale bytes of the block's 64 rows in one store: lane (fr_i, fr_g) writes the one of row fr_g * 16 + fr_i
      a.out_lo4s[(long)(mbase + fr_g * 16 + fr_i) * Lo4ScalePitch(a.ldo) + (nbase >> 6)] = (uint8_t)(e4 >> (8 * fr_g));
    }
  } else {
    // kEpiStats: lane owns rows p*16 + fr_g*4 + r, column q*16 + fr_i.  The cross-lane adds leave the sums of
    // fragment p in lane group fr_g = p, which stores them: two store instructions per q (4 fragments x 16 columns
    // each).
    const int grp0 = mbase >> 4;
    float s1v[4][4], s2v[4][4];   // [q][p]: this lane's sums over its four rows r of fragment p
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      // all 16 rows of the group pooled (every group but the first / last ones of a chunk): no row mask.  first / last
      // are the same in every lane, so the branch is wave-uniform; the arithmetic is that of the masked path (the
      // square is rounded on its own there, hence __fmul_rn here).
      const int first = __builtin_amdgcn_readfirstlane(e.first[p]), last = __builtin_amdgcn_readfirstlane(e.last[p]);
      // a lane's four rows r of a column as two pairs (r = 0, 1) and (2, 3): squares and sums on both values of a pair at once
      // (v_pk_*_f32); the sum of a column is ((z0 + z2) + (z1 + z3)) in both branches
      float s1[4], s2[4];
      if (first == 0 && last == 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // bias and ReLU per value (as pairs they need the column's bias / scale / offset splat into register pairs: six
          // more registers, which cost the single-pass per-tile kernel its second workgroup per CU), BatchNorm, squares and
          // sums as pairs
          float z0 = apply_floor(acc[p][q][0] + e.bs[q], relu_floor), z1 = apply_floor(acc[p][q][1] + e.bs[q], relu_floor);
          float z2 = apply_floor(acc[p][q][2] + e.bs[q], relu_floor), z3 = apply_floor(acc[p][q][3] + e.bs[q], relu_floor);
          z0 = __builtin_fmaf(z0, e.sc[q], e.of[q]);
          z1 = __builtin_fmaf(z1, e.sc[q], e.of[q]);
          z2 = __builtin_fmaf(z2, e.sc[q], e.of[q]);
          z3 = __builtin_fmaf(z3, e.sc[q], e.of[q]);
          const f32x2 za = {z0, z1}, zb = {z2, z3};
          const f32x2 t1 = za + zb;
          f32x2 t2;
          {
#pragma clang fp contract(off)   // the squares are rounded on their own, as in the masked branch (__fmul_rn)
            const f32x2 qa = za * za, qb = zb * zb;
            t2 = qa + qb;
          }
          s1[q] = t1[0] + t1[1];
          s2[q] = t2[0] + t2[1];
        }
      } else {
        bool ok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ok[r] = (fr_g * 4 + r >= first) && (fr_g * 4 + r < last);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float z[4], zz[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float t = apply_floor(acc[p][q][r] + e.bs[q], relu_floor);
            t = __builtin_fmaf(t, e.sc[q], e.of[q]);
            z[r] = ok[r] ? t : 0.f;
            zz[r] = ok[r] ? __fmul_rn(t, t) : 0.f;
          }
          s1[q] = (z[0] + z[2]) + (z[1] + z[3]);
          s2[q] = (zz[0] + zz[2]) + (zz[1] + zz[3]);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        s1v[q][p] = s1[q];
        s2v[q][p] = s2[q];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = nbase + q * 16 + fr_i;
      // cross-lane part of the 16-row sums; lane group fr_g ends up with the totals of fragment p = fr_g
      const float v1 = sum4_rows_scatter(s1v[q][0], s1v[q][1], s1v[q][2], s1v[q][3]);
      const float v2 = sum4_rows_scatter(s2v[q][0], s2v[q][1], s2v[q][2], s2v[q][3]);
      float* dst = a.partial + (long)(grp0 + fr_g) * 2 * a.ldp + col;
      dst[0] = v1;
      dst[a.ldp] = v2;
    }
  }
}

// (defined with the v2 kernel below: LDS-DMA from inline asm, invisible to hipcc's wait-count scoreboard)
__device__ __forceinline__ void glds16_asm(const void* g, unsigned lds_addr);
constexpr int kSplitKStages = 4;   // LDS ring of the split-K instantiation of tdnn_gemm_kernel

template <int PREC, int EPI>
__global__ __launch_bounds__(256, 2) void tdnn_gemm_kernel(const GemmArgs a) {
  constexpr bool SPLIT = PrecXPlanes(PREC) == 2;   // activations carry a residual plane
  constexpr bool WSPLIT = PrecWPlanes(PREC) == 2;  // weights carry a residual plane
  constexpr bool F16 = PrecF16(PREC);
  constexpr bool SWAP = (EPI != kEpiStats);  // weight tile is the MFMA A operand
  constexpr int NPX = SPLIT ? 2 : 1, NPW = WSPLIT ? 2 : 1;
  constexpr int STAGE = kTileBytes * (NPX + NPW);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave & 1;
  const int wave_n = wave >> 1;

  // XCD-aware tile order: block b lands on XCD b%8 (observed dispatch); the column tiles of one
  // row tile are consecutive on the same XCD so the activation rows are re-read from its L2.
  const int bid = blockIdx.x;
  int nt, mt;
  if constexpr (EPI == kEpiSplitK) {
    // a segment-level layer has one or two row tiles: with the mapping below all its workgroups sat on XCD 0 and 1 (64 of
    // them on 64 CUs behind two L2s, and with more K slices they queued there: 24 slices took 32 us against 21 for 8).
    // Plain order instead: consecutive workgroups go to consecutive XCDs.
    nt = bid % a.n_tiles;
    mt = bid / a.n_tiles;
  } else {
    const int xcd = bid & 7;
    const int slot = bid >> 3;
    nt = slot % a.n_tiles;
    mt = (slot / a.n_tiles) * 8 + xcd;
  }
  if (mt >= a.m_tiles) return;
  const int m0 = mt * kBM;
  const int n0 = nt * kBN;
  XV_FUZZ_INIT();

  // ---- per-lane staging geometry -------------------------------------------------------
  // one global_load_lds_dwordx4 per wave = 16 LDS rows x 64 B; lane l -> row l>>2, phys chunk l&3
  const int ld_row = lane >> 2;
  const int ld_chunk = (lane & 3) ^ ((lane >> 3) & 3);  // logical 16-B chunk fetched
  const int c0 = wave * 2;                                           // this wave's two 16-row chunks

  const uint16_t* wp_hi[2];
  const uint16_t* wp_lo[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int rho = (c0 + u) * 16 + ld_row;  // LDS row inside the 128-row weight tile
    const int wrow = SWAP ? ((rho & 64) | swap_fields(rho & 63)) : rho;
    const long off = (long)(n0 + wrow) * a.ldw + ld_chunk * 8;
    wp_hi[u] = a.w_hi + off;
    wp_lo[u] = WSPLIT ? a.w_lo + off : nullptr;
  }

  // activation-side staging pointers of the current K segment (one Append() term); they are
  // re-derived only when the K loop crosses into the next segment.
  const uint16_t* xp_hi[2];
  const uint16_t* xp_lo[2];
  int seg_left = 0;  // K steps left in the current segment (wave uniform)
  int ld_seg = 0;
  auto open_seg = [&](int j) {
    const Seg& sg = a.seg[j];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long off = (long)(m0 + (c0 + u) * 16 + ld_row + sg.row_shift) * sg.ld + ld_chunk * 8;
      xp_hi[u] = sg.hi + off;
      xp_lo[u] = SPLIT ? sg.lo + off : nullptr;
    }
    seg_left = sg.ksteps;
  };
  open_seg(0);

  auto stage_loads = [&](int stage) {
    char* st = smem + stage * STAGE;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int c = c0 + u;
      glds16(xp_hi[u], st + c * 1024);
      if constexpr (SPLIT) glds16(xp_lo[u], st + kTileBytes + c * 1024);
      glds16(wp_hi[u], st + NPX * kTileBytes + c * 1024);
      if constexpr (WSPLIT) glds16(wp_lo[u], st + (NPX + 1) * kTileBytes + c * 1024);
      xp_hi[u] += kBK;
      wp_hi[u] += kBK;
      if constexpr (SPLIT) xp_lo[u] += kBK;
      if constexpr (WSPLIT) wp_lo[u] += kBK;
    }
    if (--seg_left == 0 && ld_seg + 1 < a.nseg) open_seg(++ld_seg);
  };

  // ---- per-lane fragment read geometry ---------------------------------------------------
  const int fr_i = lane & 15;
  const int fr_g = lane >> 4;
  const int fr_chunk = fr_g ^ ((fr_i >> 1) & 3);
  const int x_rd = (wave_m * 64 + fr_i) * 64 + fr_chunk * 16;                     // + f*1024
  const int w_rd = NPX * kTileBytes + (wave_n * 64 + fr_i) * 64 + fr_chunk * 16;  // + f*1024

  f32x4 acc[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  int S = a.total_ksteps;
  if constexpr (EPI == kEpiSplitK) {
    // this block's K slice: skip s_begin steps (weights advance linearly, activations segment by segment)
    const int s_begin = blockIdx.y * a.ksteps_per_slice;
    const int s_end = min(S, s_begin + a.ksteps_per_slice);
    for (int t = 0; t < s_begin; ++t) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        xp_hi[u] += kBK;
        wp_hi[u] += kBK;
        if constexpr (SPLIT) xp_lo[u] += kBK;
        if constexpr (WSPLIT) wp_lo[u] += kBK;
      }
      if (--seg_left == 0 && ld_seg + 1 < a.nseg) open_seg(++ld_seg);
    }
    S = s_end - s_begin;
  }
  if constexpr (EPI == kEpiSplitK) {
    // A K slice of a segment-level layer (the embedding affine: 256 rows, K = 3000, 24 slices of 4 steps on 192 workgroups)
    // is a short chain of memory latencies: its weights are cold (the frame-level layers have turned the caches over) and
    // with one step of lookahead every step waits for its tile.  Here the ring is kSplitKStages deep and three steps are in
    // flight; the DMA is issued from inline asm and counted by hand (left to hipcc, every fragment read that may alias a
    // pending LDS-DMA gets a vmcnt(0)).
    constexpr int NST = kSplitKStages;
    constexpr int PER = 2 * (NPX + NPW);   // DMA instructions per wave and stage
    static_assert(PER * (NST - 2) < 64, "vmcnt is six bits");
    const unsigned lds0 = (unsigned)(size_t)(XV_AS3 char*)smem;
    auto stage_loads_asm = [&](int stage) __attribute__((always_inline)) {
      const unsigned st = lds0 + stage * STAGE + wave * 2048;   // chunk c = 2 * wave + u at c * 1024
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        glds16_asm(xp_hi[u], st + u * 1024);
        if constexpr (SPLIT) glds16_asm(xp_lo[u], st + kTileBytes + u * 1024);
        glds16_asm(wp_hi[u], st + NPX * kTileBytes + u * 1024);
        if constexpr (WSPLIT) glds16_asm(wp_lo[u], st + (NPX + 1) * kTileBytes + u * 1024);
        xp_hi[u] += kBK;
        wp_hi[u] += kBK;
        if constexpr (SPLIT) xp_lo[u] += kBK;
        if constexpr (WSPLIT) wp_lo[u] += kBK;
      }
      if (--seg_left == 0 && ld_seg + 1 < a.nseg) open_seg(++ld_seg);
    };
    for (int t = 0; t < NST - 1 && t < S; ++t) stage_loads_asm(t);
    for (int s = 0; s < S; ++s) {
      // stage s has landed once at most the stages issued after it are outstanding; everyone is past its reads of stage
      // s - 1, whose slot the loads issued below overwrite
      XV_FUZZ();
      const int later = min(S - 1 - s, NST - 2);
      if (later >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * PER) : "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const char* st = smem + (s % NST) * STAGE;
      s16x8 xh[4], xl[4], wh[4], wl[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        xh[f] = *(const s16x8*)(st + x_rd + f * 1024);
        wh[f] = *(const s16x8*)(st + w_rd + f * 1024);
        if constexpr (SPLIT) xl[f] = *(const s16x8*)(st + x_rd + kTileBytes + f * 1024);
        if constexpr (WSPLIT) wl[f] = *(const s16x8*)(st + w_rd + kTileBytes + f * 1024);
      }
      XV_FUZZ();
      if (s + NST - 1 < S) stage_loads_asm((s + NST - 1) % NST);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // the products in the order of the two-stage loop below: same bits
          if constexpr (WSPLIT) acc[p][q] = mfma16<F16>(wl[p], xh[q], acc[p][q]);
          if constexpr (SPLIT) acc[p][q] = mfma16<F16>(wh[p], xl[q], acc[p][q]);
          acc[p][q] = mfma16<F16>(wh[p], xh[q], acc[p][q]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    EpiRegs er0;
    float gm0[4] = {0.f, 0.f, 0.f, 0.f};
    gemm_epilogue<PREC, EPI>(a, acc, m0 + wave_m * 64, n0 + wave_n * 64, lane, er0, gm0, 0);
    return;
  }
  stage_loads(0);
  __syncthreads();  // drains the LDS-DMA queue (vmcnt(0)) and orders the LDS writes

  for (int s = 0; s < S; ++s) {
    XV_FUZZ();
    const char* st = smem + (s & 1) * STAGE;
    s16x8 xh[4], xl[4], wh[4], wl[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      xh[f] = *(const s16x8*)(st + x_rd + f * 1024);
      wh[f] = *(const s16x8*)(st + w_rd + f * 1024);
      if constexpr (SPLIT) xl[f] = *(const s16x8*)(st + x_rd + kTileBytes + f * 1024);
      if constexpr (WSPLIT) wl[f] = *(const s16x8*)(st + w_rd + kTileBytes + f * 1024);
    }
    // stage the next K step into the other buffer: every wave finished reading it before the
    // barrier that ended the previous iteration.
    XV_FUZZ();
    if (s + 1 < S) stage_loads((s + 1) & 1);

#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if constexpr (SWAP) {
          if constexpr (WSPLIT) acc[p][q] = mfma16<F16>(wl[p], xh[q], acc[p][q]);
          if constexpr (SPLIT) acc[p][q] = mfma16<F16>(wh[p], xl[q], acc[p][q]);
          acc[p][q] = mfma16<F16>(wh[p], xh[q], acc[p][q]);
        } else {
          if constexpr (SPLIT) acc[p][q] = mfma16<F16>(xl[p], wh[q], acc[p][q]);
          if constexpr (WSPLIT) acc[p][q] = mfma16<F16>(xh[p], wl[q], acc[p][q]);
          acc[p][q] = mfma16<F16>(xh[p], wh[q], acc[p][q]);
        }
      }
    }
    // keep the MFMAs above the wait: they are register-only and hipcc would otherwise sink them
    // below the barrier, serialising the LDS-DMA flight with the matrix work.
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // next stage landed (vmcnt(0)) and everyone is done with this one
  }

  EpiRegs er;
  epilogue_prefetch<EPI>(a, m0 + wave_m * 64, n0 + wave_n * 64, lane, er);
  float gm[4] = {0.f, 0.f, 0.f, 0.f};
  gemm_epilogue<PREC, EPI>(a, acc, m0 + wave_m * 64, n0 + wave_n * 64, lane, er, gm, 0);
}

// ---------------------------------------------------------------------------------------------
// v2: 256x128x32 tile, 8 waves (4 along frames x 2 along columns, 64x64 each).
//
//  * Splice staged in LDS.  The K segments of a layer are grouped: consecutive Append() terms that read the same
//    source at uniformly spaced time offsets (tdnn2: -2,0,2; tdnn3: -3,0,3; tdnn1: -2..2) form one group.  For a
//    group the activation tile is fetched ONCE per 32-column K chunk as 272 rows (256 + 16 halo) and every time
//    offset of the group reads its MFMA fragments from that tile at a row displacement - the spliced matrix never
//    exists, and the L2->LDS traffic of the activations drops by the number of offsets (measured before this
//    change: DMA-only 0.37 ms vs MFMA-only 0.33 ms per tdnn2 launch, i.e. the DMA path co-limited the kernel).
//    The XOR swizzle of the 16-byte chunks is a function of the LDS row, so displaced reads stay conflict free.
//  * LDS rings: activations 3 slots x 272 rows x 64 B x planes, weights 3 slots x 128 rows x 64 B x planes
//    (150 KiB in split mode).  The LDS-DMA is issued from inline asm so that hipcc does not track it (it would
//    put a vmcnt(0) in front of every ds_read that may alias a pending DMA); completion is counted by hand.
//  * Ping-pong schedule: see the comment inside the kernel.
__device__ __forceinline__ void glds16_asm(const void* g, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(g), "s"(lds_addr)
      : "memory");
}

// Same, scalar base + 32-bit per-lane byte offset: the wave-uniform part of the address lives in SGPRs and moves with
// scalar adds; a lane keeps one offset register for all its DMA instead of a 64-bit pointer per instruction.
__device__ __forceinline__ void glds16_sbase(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_addr)
      : "memory");
}

// The same without saving and restoring m0 around the instruction (three scalar instructions less per DMA; the LOAD part of a
// phase of tdnn_gemm_kernel_p8 is what its barrier intervals wait for: -0.9 % on that kernel's launches, profiles/r05_p8_priority.md).
// m0 is a reserved register to hipcc - it does not allocate it and cannot be told about the write (a clobber is refused with a
// warning) - and it only ever reads m0 in instructions it sets it up for itself (LDS-DMA builtins, s_movrel, GWS, sendmsg), none
// of which a kernel that uses this function may contain: tdnn_gemm_kernel_p8 issues every LDS-DMA through these asm forms, each of
// which writes m0 first (tests/test_kernel_resources.py checks the ISA: no m0 read outside them).
__device__ __forceinline__ void glds16_sbase_m0(const void* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

// 4 bytes per lane (LDS destination = base + lane * 4)
__device__ __forceinline__ void glds4_sbase(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_addr)
      : "memory");
}

// the same for lanes 0..15 only (the others neither read nor write)
__device__ __forceinline__ void glds4_sbase_lanes16(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  unsigned long long ex;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %4\n\ts_mov_b64 exec, 0xffff\n\t"
      "global_load_lds_dword %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
      : "=&s"(keep), "=&s"(ex)
      : "v"(voff), "s"(sbase), "s"(lds_addr)
      : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm_lgkm0_barrier() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ void wait_vm_lgkm0_barrier_n(int n) {   // n wave-uniform, 0..11 (else: everything)
  switch (n) {
    case 1: wait_vm_lgkm0_barrier<1>(); break;
    case 2: wait_vm_lgkm0_barrier<2>(); break;
    case 3: wait_vm_lgkm0_barrier<3>(); break;
    case 4: wait_vm_lgkm0_barrier<4>(); break;
    case 5: wait_vm_lgkm0_barrier<5>(); break;
    case 6: wait_vm_lgkm0_barrier<6>(); break;
    case 7: wait_vm_lgkm0_barrier<7>(); break;
    case 8: wait_vm_lgkm0_barrier<8>(); break;
    case 9: wait_vm_lgkm0_barrier<9>(); break;
    case 10: wait_vm_lgkm0_barrier<10>(); break;
    case 11: wait_vm_lgkm0_barrier<11>(); break;
    default: wait_vm_lgkm0_barrier<0>(); break;
  }
}

// The single-pass instantiations run two workgroups per CU (4 waves per SIMD): 128 registers.  hipcc lands on 128-130 for
// them depending on epilogue details, and at 130 the second workgroup does not fit (tdnn5, fp16: 0.19 -> 0.28 ms): the
// minimum is stated rather than hoped for.
template <int PREC, int EPI>
__global__ __launch_bounds__(512) void tdnn_gemm_kernel_v2(const GemmArgs a) {
  constexpr bool SPLIT = PrecXPlanes(PREC) == 2;   // activations carry a residual plane
  constexpr bool WSPLIT = PrecWPlanes(PREC) == 2;  // weights carry a residual plane
  constexpr bool MX = PrecMx(PREC);            // see the stream-K kernel: same arithmetic, same order, MF = 4
  constexpr bool F16 = PrecF16(PREC);
  constexpr bool SWAP = (EPI != kEpiStats);
  constexpr int NPX = SPLIT ? 2 : 1, NPW = WSPLIT ? 2 : 1;
  constexpr int XROWS = 256 + 16;              // tile rows + halo for the time offsets of a group
  constexpr int XT = XROWS * kBK * 2;          // 17 KiB: one activation plane slot
  constexpr int WT = kTileBytes;               //  8 KiB: one 128-row weight plane slot
  constexpr int XSLOT = NPX * XT;
  constexpr int WSLOT = NPW * WT;
  constexpr int WBASE = 3 * XSLOT;             // weight ring behind the activation ring
  constexpr int SCB = WBASE + 3 * WSLOT;       // kPrecFp16Mx: scales of the residual tiles, three blocks x 128 rows x 4 lane groups
  constexpr bool MX2 = PrecMx2(PREC);          // second K walk over the 4-bit planes: see the stream-K kernel
  constexpr int SLB = SCB + 1536;
  constexpr int SLAB_ROWS = XROWS;
  constexpr int SLAB_BYTES = SLAB_ROWS * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave & 3;
  const int wave_n = wave >> 2;

  const int m_tiles = a.m_tiles >> 1;  // 256-row tiles
  const int bid = blockIdx.x;
  const int xcd = bid & 7;
  const int bslot = bid >> 3;
  const int nt = bslot % a.n_tiles;
  const int mt = (bslot / a.n_tiles) * 8 + xcd;
  if (mt >= m_tiles) return;
  const int m0 = mt * 256;
  const int n0 = nt * kBN;
  XV_FUZZ_INIT();
  // Start stagger: the first workgroup of every CU waits a different fraction of one tile time.  All tiles take the
  // same time, so without it every CU reaches its epilogue in the same microsecond and the 32 MB store burst of a
  // round (128 KiB per CU) runs at the HBM write rate with nothing to overlap; staggered, the stores of one CU drain
  // while the others multiply.  The late starters simply take one tile less (6.25 tiles per CU -> 6 or 7).
  if (a.stagger_units > 0 && bid < a.stagger_wgs) {
    const int n = (((bid >> 3) * 37 + (bid & 7) * 5) & 31) * a.stagger_units >> 5;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(32);   // 2048 cycles each
  }

  // epilogue parameters: the statistics epilogue's few registers are fetched now and are in flight during the K
  // loop; the 48 of the activation epilogue are fetched after it (held across the loop they push the single-pass
  // kernels from 120 to 162 VGPRs, i.e. from two co-resident workgroups per CU to one: -20 % measured)
  EpiRegs er;
  if constexpr (EPI == kEpiStats) epilogue_prefetch<EPI>(a, m0 + wave_m * 64, n0 + wave_n * 64, lane, er);

  const int ld_row = lane >> 2;
  const int ld_chunk = (lane & 3) ^ ((lane >> 3) & 3);

  // weights: this wave stages 16-row chunk `wave` of the 128-row tile; per lane the row start is fixed and the
  // K column of a step is a wave-uniform offset
  const uint16_t* wrow_hi;
  const uint16_t* wrow_lo;
  const uint8_t* wrow_4;
  const uint8_t* wrow_b;
  {
    const int rho = wave * 16 + ld_row;
    const int wrow = SWAP ? ((rho & 64) | swap_fields(rho & 63)) : rho;
    const long off = (long)(n0 + wrow) * a.ldw + ld_chunk * 8;
    wrow_hi = a.w_hi + off;
    wrow_lo = (WSPLIT && !MX) ? a.w_lo + off : nullptr;
    wrow_4 = MX ? a.w4 + (long)(n0 + wrow) * a.ldw4 + ld_chunk * 16 : nullptr;
    wrow_b = MX2 ? a.w4b + (long)(n0 + wrow) * a.ldw4b + ld_chunk * 16 : nullptr;
  }
  const int SH = a.total_ksteps;
  const int NG = a.ngrp + (MX2 ? a.ngrp_lo : 0);
  const unsigned lds_base = (unsigned)(size_t)(XV_AS3 char*)smem;

  // ---- issue side: walks the K steps in the order  group -> K chunk -> time offset ------------------------------
  int ig = 0, ikk = 0, ij = 0;        // group, 32-column chunk inside the group, offset index inside the group
  int ixslot = 0, iwslot = 0;         // ring slots the next activation stage / weight stage go to
  int istep = 0;
  int islab = 0, rslab = 0;   // kPrecFp16Mx2: slabs of activation-residual scales issued / begun by the read side
  Grp gi = a.grp[0];
  auto issue_step = [&]() __attribute__((always_inline)) -> int {    // returns the number of DMA instructions this wave issued
    int n = 0;
    if (ij == 0) {
      const unsigned st = __builtin_amdgcn_readfirstlane(lds_base + ixslot * XSLOT);
      const long col = (long)ikk * kBK + ld_chunk * 8;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c = wave + 8 * u;
        const long off = (long)(m0 + gi.shift0 + c * 16 + ld_row) * gi.ld + col;
        glds16_asm(gi.hi + off, st + c * 1024);
        if constexpr (SPLIT) glds16_asm(gi.lo + off, st + XT + c * 1024);
      }
      n += 2 * NPX;
      if (wave == 0 && gi.nshift > 1) {  // halo rows 256..271 (only read by displaced offsets)
        const long off = (long)(m0 + gi.shift0 + 256 + ld_row) * gi.ld + col;
        glds16_asm(gi.hi + off, st + 16 * 1024);
        if constexpr (SPLIT) glds16_asm(gi.lo + off, st + XT + 16 * 1024);
        n += NPX;
      }
      ixslot = ixslot == 2 ? 0 : ixslot + 1;
    }
    const bool lo_step = MX2 && istep >= SH;
    if constexpr (MX2) {
      if (lo_step && ij == 0 && (ikk & 1) == 0) {   // scales of the activation residuals of this and the next chunk
        const unsigned sl = lds_base + SLB + (islab % 3) * SLAB_BYTES;
        ++islab;
        const unsigned voff = (unsigned)(lane * gi.ld4s);
        if (wave < 4) {
          glds4_sbase(gi.lo4s + (long)(m0 + gi.shift0 + wave * 64) * gi.ld4s + (ikk >> 1) * 4, voff, sl + wave * 256);
          n += 1;
        } else if (wave == 4) {
          glds4_sbase_lanes16(gi.lo4s + (long)(m0 + gi.shift0 + 256) * gi.ld4s + (ikk >> 1) * 4, voff, sl + 256 * 4);
          n += 1;
        }
      }
    }
    if (lo_step) {
      if constexpr (MX2) {
        const unsigned st = __builtin_amdgcn_readfirstlane(lds_base + WBASE + iwslot * WSLOT + wave * 1024);
        const int t_lo = istep - SH;
        glds16_asm(wrow_b + t_lo * 64, st);
        n += 1;
        if (wave < 2) {
          glds4_sbase(a.w4b_scale + ((long)nt * a.lo_ksteps + t_lo) * 512 + wave * 256, (unsigned)lane * 4u,
                      lds_base + SCB + ((t_lo + (SH >> 2)) % 3) * 512 + wave * 256);
          n += 1;
        }
      }
      iwslot = iwslot == 2 ? 0 : iwslot + 1;
    } else {
      const unsigned st = __builtin_amdgcn_readfirstlane(lds_base + WBASE + iwslot * WSLOT + wave * 1024);
      const int wcol = gi.wcol0 + ij * gi.wstride + ikk * kBK;
      glds16_asm(wrow_hi + wcol, st);
      n += 1;
      if constexpr (MX) {
        if ((istep & 3) == 1) {   // 4-bit residual tile of block istep / 4 -> its own ring (residual halves of the slots)
          const unsigned st4 = __builtin_amdgcn_readfirstlane(lds_base + WBASE + ((istep >> 2) % 3) * WSLOT + WT + wave * 1024);
          glds16_asm(wrow_4 + (istep >> 2) * 64, st4);
          n += 1;
          if (wave < 2) {   // and its scales: 512 contiguous bytes per (tile, block), already in reading order
            glds4_sbase(a.w4_scale + ((long)nt * (a.total_ksteps >> 2) + (istep >> 2)) * 512 + wave * 256, (unsigned)lane * 4u,
                        lds_base + SCB + ((istep >> 2) % 3) * 512 + wave * 256);
            n += 1;
          }
        }
      } else if constexpr (WSPLIT) {
        glds16_asm(wrow_lo + wcol, st + WT);
        n += 1;
      }
      iwslot = iwslot == 2 ? 0 : iwslot + 1;
    }
    ++istep;
    if (++ij == gi.nshift) {
      ij = 0;
      if (++ikk == gi.ksteps) {
        ikk = 0;
        if (++ig < NG) gi = a.grp[ig];
      }
    }
    return n;
  };

  // kPrecFp16Mx scales (see the stream-K kernel)
  int xs_b = 0, ws_v = 0;
  auto load_xscales = [&](const unsigned* gmax) __attribute__((always_inline)) {
    if constexpr (MX) {
      typedef __attribute__((ext_vector_type(4))) unsigned uvec;
      const unsigned* p = gmax + ((m0 + wave_m * 64) >> 4);
      uvec g;
      asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(g) : "s"(p) : "memory");
      xs_b = 0;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        unsigned e = (g[f] >> 23) & 255u;
        e = e < 16u ? 16u : (e > 200u ? 200u : e);
        xs_b |= (int)((e - 2u) << (8 * f));
      }
    }
  };
  if constexpr (MX) load_xscales(gi.gmax);

  // ---- read side ---------------------------------------------------------------------------------------------
  int rg = 0, rkk = 0, rj = 0;
  int rxslot = 0, rwslot = 0;
  int r_nshift = gi.nshift, r_ksteps = gi.ksteps, r_dstep = gi.dstep;
  const int fr_i = lane & 15;
  const int fr_g = lane >> 4;
  const int w_rd = (wave_n * 64 + fr_i) * 64 + (fr_g ^ ((fr_i >> 1) & 3)) * 16;
  struct Frags {
    s16x8 xh[4], xl[4], wh[4], wl[4];
  };
  int rblk = 0;
  bool xs_reload = false;
  auto read_step = [&](Frags& f) __attribute__((always_inline)) {
    if constexpr (MX) {
      if (xs_reload) {
        load_xscales(a.grp[rg].gmax);
        xs_reload = false;
      }
    }
    const char* xs = smem + rxslot * XSLOT;
    const char* ws = smem + WBASE + rwslot * WSLOT;
    const int row = wave_m * 64 + fr_i + rj * r_dstep;  // displaced by the time offset of this step
    const int x_rd = row * 64 + (fr_g ^ ((row >> 1) & 3)) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f.xh[i] = *(const s16x8*)(xs + x_rd + i * 1024);
      f.wh[i] = *(const s16x8*)(ws + w_rd + i * 1024);
      if constexpr (SPLIT) f.xl[i] = *(const s16x8*)(xs + x_rd + XT + i * 1024);
      if constexpr (WSPLIT && !MX) f.wl[i] = *(const s16x8*)(ws + w_rd + WT + i * 1024);
    }
    rwslot = rwslot == 2 ? 0 : rwslot + 1;
    if (++rj == r_nshift) {
      rj = 0;
      rxslot = rxslot == 2 ? 0 : rxslot + 1;
      if (++rkk == r_ksteps) {
        rkk = 0;
        if (++rg < NG) {
          r_nshift = a.grp[rg].nshift;
          r_ksteps = a.grp[rg].ksteps;
          r_dstep = a.grp[rg].dstep;
          if constexpr (MX) xs_reload = rg < a.ngrp;
        }
      }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mfmas = [&](const Frags& f) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if constexpr (MX) {
          if constexpr (SWAP) mfma16_f16_inplace(f.wh[p], f.xh[q], acc[p][q]);
          else mfma16_f16_inplace(f.xh[p], f.wh[q], acc[p][q]);
        } else if constexpr (SWAP) {
          if constexpr (WSPLIT) acc[p][q] = mfma16<F16>(f.wl[p], f.xh[q], acc[p][q]);
          if constexpr (SPLIT) acc[p][q] = mfma16<F16>(f.wh[p], f.xl[q], acc[p][q]);
          acc[p][q] = mfma16<F16>(f.wh[p], f.xh[q], acc[p][q]);
        } else {
          if constexpr (SPLIT) acc[p][q] = mfma16<F16>(f.xl[p], f.wh[q], acc[p][q]);
          if constexpr (WSPLIT) acc[p][q] = mfma16<F16>(f.xh[p], f.wl[q], acc[p][q]);
          acc[p][q] = mfma16<F16>(f.xh[p], f.wh[q], acc[p][q]);
        }
      }
    }
  };

  i32x4 x4[MX ? 4 : 1];
  auto convert = [&](const Frags& f, const int s) __attribute__((always_inline)) {
    if constexpr (MX) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        x4[i][s] = cvt8_fp4(f.xh[i], __builtin_bit_cast(float, ((unsigned)(xs_b >> (8 * i)) & 255u) << 23), x4[i][s]);
        asm volatile("" : "+v"(x4[i]));
      }
    }
  };
  auto mx_mfmas = [&](Frags& f) __attribute__((always_inline)) {
    if constexpr (MX) {
      const char* w4s = smem + WBASE + (rblk % 3) * WSLOT + WT + w_rd;
#pragma unroll
      for (int w = 0; w < 4; ++w) f.wh[w] = *(const s16x8*)(w4s + w * 1024);
      // (scales of the block: ws_v, read in the LOAD segment - read_scales)
      asm volatile("s_nop 4" ::: "memory");
      static_for<0, 4>([&](auto P) {
        static_for<0, 4>([&](auto Q) {
          constexpr int p = decltype(P)::value, q = decltype(Q)::value;
          if constexpr (SWAP) mfma_mx4_inplace<p, q>(f.wh[p], x4[q], acc[p][q], ws_v, xs_b);
          else mfma_mx4_inplace<p, q>(x4[p], f.wh[q], acc[p][q], xs_b, ws_v);
        });
      });
    }
  };

  // The weight scales of a block (first walk) / of a step (second walk) share one ring of three 512-byte slots, and the slot
  // of step j + 2 is the slot of step j - 1: they are read in the LOAD segment, like the weight tiles, never in the COMPUTE
  // segment - wave group 1's COMPUTE of step j - 1 runs beside wave group 0's LOAD of step j, which stages the scales of step
  // j + 2 into that slot (found by tools/repeat_mx_case.py with a second stream keeping the chip busy: a late wave of group 1
  // read the scales of a step three ahead; run-to-run differences of 1e-4 on the c-vector network)
  auto read_scales = [&](const int slot) __attribute__((always_inline)) {
    return *(const int*)(smem + SCB + slot * 512 + wave_n * 256 + (fr_i * 4 + fr_g) * 4);
  };
  auto lo_mfmas = [&](Frags& f, const int ws_lo, const int lg, const int lkk, const int lj, const int slab) __attribute__((always_inline)) {
    if constexpr (MX2) {
      const uint8_t* sl = (const uint8_t*)smem + SLB + slab * SLAB_BYTES + (wave_m * 64 + fr_i + lj * a.grp[lg].dstep) * 4 +
                          2 * (lkk & 1) + (fr_g >> 1);
      int xs_lo = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) xs_lo |= (int)sl[i * 64] << (8 * i);
      asm volatile("s_nop 4" ::: "memory");   // VALU-assembled scale word -> MFMA operand (see the stream-K kernel)
      static_for<0, 4>([&](auto P) {
        static_for<0, 4>([&](auto Q) {
          constexpr int p = decltype(P)::value, q = decltype(Q)::value;
          if constexpr (SWAP) mfma_mx4_inplace<p, q>(f.wh[p], f.xh[q], acc[p][q], ws_lo, xs_lo);
          else mfma_mx4_inplace<p, q>(f.xh[p], f.wh[q], acc[p][q], xs_lo, ws_lo);
        });
      });
    }
  };

  // Ping-pong schedule.  The 8 waves form two groups (waves 0-3 / 4-7: one wave of each group per SIMD).  A K
  // step of a group is a LOAD segment (ds_read its fragments of step j, issue its share of the DMA of step j+2,
  // wait until only that share is outstanding, i.e. its share of step j+1 has landed) and a COMPUTE segment (48 /
  // 16 MFMAs), separated by workgroup barriers; group 1 runs one barrier interval behind group 0, so on every
  // SIMD one wave is in its MFMA segment while its partner does the LDS / DMA issue work.
  //   RAW: a share of step j+1 is waited for at the end of the owner's LOAD_j, which is followed by a barrier that
  //        every reader passes before its LOAD_{j+1}.
  //   WAR: the slots written for step j+2 were last read for step j-1 (weights) or earlier (activations); both
  //        groups finished those reads (lgkmcnt(0) before their barrier) at least one barrier earlier.
  // n = DMA instructions this wave may leave in flight: NPW (weights only), NPW + 2*NPX (+2 activation chunks),
  // NPW + 3*NPX (wave 0, + halo chunk).
  auto wait_and_barrier = [&](int n) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0) through the builtin: hipcc's scoreboard then knows about it
    if constexpr (MX) {
      wait_vm_lgkm0_barrier_n(n);
    } else {
      if (n == 0) wait_vm_lgkm0_barrier<0>();
      else if (n == NPW) wait_vm_lgkm0_barrier<NPW>();
      else if (n == NPW + 2 * NPX) wait_vm_lgkm0_barrier<NPW + 2 * NPX>();
      else wait_vm_lgkm0_barrier<NPW + 3 * NPX>();
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto plain_barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  const int S = SH + (MX2 ? a.lo_ksteps : 0);
  const int group = wave >> 2;
  issue_step();
  const int n1 = S > 1 ? issue_step() : 0;
  if constexpr (MX) {
    // both first steps (and the weight-row scale loads hipcc tracks) complete here: see the stream-K kernel
    __builtin_amdgcn_s_waitcnt(0x0f70);
    wait_and_barrier(0);
  } else {
    wait_and_barrier(n1);            // every share of step 0 has landed
  }
  if (group == 1) plain_barrier();   // group 1 runs one barrier interval behind group 0
  Frags f;
  if constexpr (MX) {
#pragma nounroll
    for (int j = 0; j < SH; j += 4, ++rblk) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        XV_FUZZ();
        read_step(f);
        if (s == 3) ws_v = read_scales(rblk % 3);
        XV_FUZZ();
        const int n = (j + s + 2 < S) ? issue_step() : 0;
        wait_and_barrier(n);
        XV_FUZZ();
        __builtin_amdgcn_s_setprio(1);
        mfmas(f);
        convert(f, s);
        if (s == 3) mx_mfmas(f);
        __builtin_amdgcn_s_setprio(0);
        plain_barrier();
      }
    }
    if constexpr (MX2) {
#pragma nounroll
      for (int j = SH; j < S; ++j) {   // the second walk (see the stream-K kernel)
        const int lg = rg, lkk = rkk, lj = rj;
        if (lj == 0 && (lkk & 1) == 0) ++rslab;
        XV_FUZZ();
        read_step(f);
        const int ws_lo = read_scales((j - SH + (SH >> 2)) % 3);
        XV_FUZZ();
        const int n = (j + 2 < S) ? issue_step() : 0;
        wait_and_barrier(n);
        XV_FUZZ();
        __builtin_amdgcn_s_setprio(1);
        lo_mfmas(f, ws_lo, lg, lkk, lj, (rslab - 1) % 3);
        __builtin_amdgcn_s_setprio(0);
        plain_barrier();
      }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  } else {
    for (int j = 0; j < S; ++j) {
      // ---- LOAD segment of step j
      XV_FUZZ();
      read_step(f);
      XV_FUZZ();
      const int n = (j + 2 < S) ? issue_step() : 0;
      wait_and_barrier(n);
      XV_FUZZ();
      // ---- COMPUTE segment of step j
      __builtin_amdgcn_s_setprio(1);
      mfmas(f);
      __builtin_amdgcn_s_setprio(0);
      plain_barrier();
    }
  }
  if (group == 0) plain_barrier();
  if constexpr (EPI != kEpiStats) epilogue_prefetch<EPI>(a, m0 + wave_m * 64, n0 + wave_n * 64, lane, er);
  float gm[4] = {0.f, 0.f, 0.f, 0.f};
  gemm_epilogue<PREC, EPI>(a, acc, m0 + wave_m * 64, n0 + wave_n * 64, lane, er, gm, 0);
}

// Which GEMM variant to launch: 1 = 128x128 / 2-stage, 2 = 256x128 / 3-stage ring, 4 = stream-K (persistent; falls
// back to 2 / 1 when the launch has too few tiles or an odd shape), 0 = by precision and K length (default, see
// launch_one).  XVEC_DEBUG=gemm_variant=... overrides.
// (function-local static initialisers: thread-safe - the engines of a --devices job launch from a consumer thread each)
static int gemm_variant() {
  static const int v = [] {
    const int x = DebugKnobInt("gemm_variant", 0);
    return (x != 0 && x != 1 && x != 2 && x != 4) ? 0 : x;
  }();
  return v;
}
// Frame fragments per wave of the stream-K variant: 8 (512-row tiles) where the LDS allows, else 4.  XVEC_DEBUG=sk_mf=4
// forces the 256-row tiles.
static int sk_max_mf() {
  static const int v = [] {
    const int x = DebugKnobInt("sk_mf", 8);
    return (x != 4 && x != 8) ? 8 : x;
  }();
  return v;
}

// The dynamic-LDS attribute of a kernel is per device: one bit per device index, so that a process driving several
// GPUs (xv_ctx_create_broadcast) sets it on each of them.
static bool lds_attr_needed(std::atomic<unsigned long long>* done_mask, int* dev_out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  *dev_out = dev;
  return !((done_mask->load(std::memory_order_acquire) >> (dev & 63)) & 1ull);
}

// Groups consecutive K segments that read the same source at uniformly spaced time offsets (see the v2 kernel); the
// grouping rule itself is PlanWalkGroups (kernels.h), shared with the host code that packs weights in walk order.
static void build_groups(GemmArgs* g) {
  long key[kMaxSeg];
  int shift[kMaxSeg], ksteps[kMaxSeg];
  for (int j = 0; j < g->nseg; ++j) {
    key[j] = j;
    for (int k = 0; k < j; ++k)
      if (g->seg[k].hi == g->seg[j].hi && g->seg[k].lo == g->seg[j].lo && g->seg[k].ld == g->seg[j].ld) {
        key[j] = key[k];
        break;
      }
    shift[j] = g->seg[j].row_shift;
    ksteps[j] = g->seg[j].ksteps;
  }
  WalkGroup wg[kMaxSeg];
  g->ngrp = PlanWalkGroups(g->nseg, key, shift, ksteps, wg);
  for (int i = 0; i < g->ngrp; ++i) {
    const Seg& s0 = g->seg[wg[i].first_seg];
    Grp& G = g->grp[i];
    G.hi = s0.hi;
    G.lo = s0.lo;
    G.ld = s0.ld;
    G.ksteps = wg[i].ksteps;
    G.nshift = wg[i].nshift;
    G.shift0 = wg[i].shift0;
    G.dstep = wg[i].dstep;
    G.wcol0 = wg[i].wcol0;
    G.wstride = wg[i].wstride;
    G.pad_ = 0;
    G.gmax = s0.gmax;
    G.lo4s = nullptr;
    G.ld4s = 0;
    G.pad2_ = 0;
  }
  g->ngrp_lo = 0;
  g->lo_ksteps = 0;
}

// kPrecFp16Mx2: the groups of the second K walk behind the first ones - the same sources, time offsets and order, steps of
// 128 columns over the 4-bit residual planes (as 16-bit elements: a quarter of the row pitch and of the steps).
static void build_lo_groups(GemmArgs* g) {
  WalkGroup wg[kMaxSeg];
  {
    long key[kMaxSeg];
    int shift[kMaxSeg], ksteps[kMaxSeg];
    for (int j = 0; j < g->nseg; ++j) {
      key[j] = j;
      for (int k = 0; k < j; ++k)
        if (g->seg[k].hi == g->seg[j].hi && g->seg[k].lo == g->seg[j].lo && g->seg[k].ld == g->seg[j].ld) {
          key[j] = key[k];
          break;
        }
      shift[j] = g->seg[j].row_shift;
      ksteps[j] = g->seg[j].ksteps;
    }
    PlanWalkGroups(g->nseg, key, shift, ksteps, wg);
  }
  g->ngrp_lo = g->ngrp;
  g->lo_ksteps = 0;
  for (int i = 0; i < g->ngrp; ++i) {
    const Seg& s0 = g->seg[wg[i].first_seg];
    Grp& G = g->grp[g->ngrp + i];
    G = g->grp[i];
    G.hi = (const uint16_t*)s0.lo4;
    G.lo = nullptr;
    G.ld = s0.ld / 4;
    G.ksteps = g->grp[i].ksteps / 4;
    G.gmax = nullptr;
    G.lo4s = s0.lo4s;
    G.ld4s = Lo4ScalePitch(s0.ld);
    g->lo_ksteps += G.ksteps * G.nshift;
  }
}

bool gemm_mx2_applicable(const GemmArgs& a) {
  if (!gemm_mx_applicable(a) || !a.w4b || !a.w4b_scale || a.ldw4b <= 0) return false;
  GemmArgs b = a;
  build_groups(&b);
  for (int i = 0; i < b.ngrp; ++i)   // whole 128-column steps
    if (b.grp[i].ksteps % 4 || b.grp[i].ld % 128) return false;
  for (int j = 0; j < a.nseg; ++j)
    if (!a.seg[j].lo4 || !a.seg[j].lo4s) return false;
  return true;
}

bool gemm_mx_applicable(const GemmArgs& a) {
  if (!a.w4 || !a.w4_scale || a.ldw4 <= 0 || (a.m_tiles & 1) || (a.total_ksteps & 3)) return false;
  GemmArgs b = a;
  build_groups(&b);
  for (int i = 0; i < b.ngrp; ++i)
    if (!b.grp[i].gmax || (b.grp[i].ksteps * b.grp[i].nshift) % 4) return false;
  return true;
}

// CUs of the CURRENT device (a --devices job drives several; the count is kept per ordinal, filled once under a lock)
static int device_cu_count() {
  static std::mutex mu;
  static int count[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  std::lock_guard<std::mutex> lock(mu);
  if (count[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    count[dev] = n;
  }
  return count[dev];
}

// ---------------------------------------------------------------------------------------------
// v4 ("stream-K"): the v2 pipeline run by a persistent grid of one workgroup per CU.
//
//  * The K steps of all tiles of a launch (tiles x S) are dealt out evenly: the tiles are cut into 8 contiguous blocks
//    (one per XCD, whole tiles), and inside a block workgroup j of G/8 takes steps [steps*j/(G/8), steps*(j+1)/(G/8)).
//    A workgroup's range covers part of a first tile ("head": its last K steps), whole tiles, and part of a last tile
//    ("tail": its first K steps).  6.25 tiles per CU then cost 6.25 tile times instead of 7 rounds.
//  * Order inside a workgroup: tail part first - its raw accumulators go to a workspace slot and a flag is
//    released -, then the whole tiles, then the head part, which starts from the accumulators the previous workgroup
//    of the block left (its tail part, stored long before) and runs the epilogue.  The accumulation order of every
//    output element is the plain K order, so results are bit-identical to the unsplit kernels.  A workgroup only ever
//    waits for a lower-numbered one whose first action is the store it waits for: no deadlock under any dispatch.
//  * MF = 16-row fragments per wave along the frames: 4 -> 256 x 128 tile as in v2; 8 -> 512 x 128 tile, a wave owns
//    128 x 64 outputs: 16 fragment reads per 64 MFMA-pairs instead of 12 per 32 (the LOAD segment of v2's ping-pong
//    schedule, not the MFMA segment, sets the step time once a product costs two MFMAs or one), and the activation
//    tile of a time-offset group is reused by twice as many MFMAs.  147 KiB of LDS with split weights.
template <int PREC, int EPI, int MF>
__global__ __launch_bounds__(512) void tdnn_gemm_kernel_sk(const GemmArgs a) {
  constexpr bool SPLIT = PrecXPlanes(PREC) == 2;
  constexpr bool WSPLIT = PrecWPlanes(PREC) == 2;
  constexpr bool MX = PrecMx(PREC);            // the second weight plane is the 4-bit residual, used once per four K steps
  constexpr bool F16 = PrecF16(PREC);
  constexpr bool SWAP = (EPI != kEpiStats);
  constexpr int NPX = SPLIT ? 2 : 1, NPW = WSPLIT ? 2 : 1;
  constexpr int TM = 64 * MF;                  // rows of a workgroup tile (4 waves x MF fragments x 16)
  constexpr int CH = MF / 2;                   // 16-row activation chunks each wave stages per tile (TM / 16 / 8)
  constexpr int NH = MF / 4;                   // 64 x 64 blocks of a wave's tile (the epilogues work on 64 x 64)
  // Wave tile.  MF = 4: 64 rows x 64 columns (waves 4 x 2).  MF = 8 ("wide"): 64 rows x all 128 columns (waves 8 x 1):
  // 4 activation fragments against 8 weight fragments per step - the same 12 fragment reads and 32 products as
  // 128 x 64, but every activation fragment is read, and in kPrecFp16Mx converted to 4 bits, by ONE wave instead of
  // two (the conversions, 48 VALU instructions per wave and step with 128 x 64, cost 14 % of tdnn2 and 30 % of tdnn5).
  constexpr bool WIDE = (MF == 8);
  constexpr int XF = WIDE ? 4 : MF;            // activation fragments per wave
  constexpr int WF = WIDE ? 8 : 4;             // weight fragments per wave
  constexpr int XT = (TM + 16) * kBK * 2;      // one activation plane slot (tile rows + halo of a time-offset group)
  constexpr int WT = kTileBytes;
  constexpr int XSLOT = NPX * XT;
  constexpr int WSLOT = NPW * WT;
  constexpr int WBASE = 3 * XSLOT;
  constexpr int SCB = WBASE + 3 * WSLOT;       // kPrecFp16Mx: scales of the residual tiles, three blocks x 128 rows x 4 lane groups
  constexpr int PB = SCB + 1536;               // epilogue parameters of the tile's columns, two buffers of 3 x 128 floats
  // kPrecFp16Mx2: the K walk goes on, after the S steps of 32 columns, with S / 4 steps of 128 columns over the 4-bit
  // residual planes of the activations (64 bytes per row and step: the tiles have the shape of the fp16 tiles and use the
  // same rings and fragment reads) against the 4-bit image of the weights; one block-scaled MFMA per fragment pair.
  // Scales: weights - 512 bytes per step through the SCB ring like those of the residual blocks; activations - one byte
  // per row and 64 columns: the dword of a row that covers two consecutive steps (256 columns) is staged for the tile
  // rows + halo as a "slab" when the walk of a group reaches it (ring of three slabs; the issue and the read side count
  // them with the same rule).
  constexpr bool MX2 = PrecMx2(PREC);
  constexpr int SLB = PB + 2 * 1536;
  constexpr int SLAB_ROWS = TM + 16;
  constexpr int SLAB_BYTES = SLAB_ROWS * 4;
  constexpr int KQ = MX ? 4 : 1;               // K steps are dealt out in units of KQ (a 128-deep block is never cut)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row_w = WIDE ? wave * 64 : (wave & 3) * (16 * MF);   // first row / column of the wave's tile inside the
  const int col_w = WIDE ? 0 : (wave >> 2) * 64;                 // workgroup tile
  const int group = wave >> 2;
  const int bid = blockIdx.x;
  XV_FUZZ_INIT();

  // ---- this workgroup's share of the K steps ---------------------------------------------------------------------
  // XCD block `xcd` owns a contiguous range of whole ROW tiles.  Its G/8 workgroups form G/8/L groups of L "column
  // lanes": lane l walks the column tiles [l*cpl, (l+1)*cpl) of every row tile of the block, and the K steps of that
  // walk are dealt out evenly over the groups.  The L workgroups of a group therefore sit on the same row tile at the
  // same time, so its activation rows come out of HBM once and out of the XCD's L2 for the others (a contiguous
  // range of whole tiles per workgroup had 32 workgroups stream 32 different row tiles through a 4 MB L2: 1.3 GB of
  // fabric reads per tdnn2 launch against 0.38 GB for the per-tile kernel).
  const int SH = a.total_ksteps;               // steps of the first walk
  const int S = SH + (MX2 ? a.lo_ksteps : 0);  // all steps of a tile
  const int NG = a.ngrp + (MX2 ? a.ngrp_lo : 0);
  const int SQ = S / KQ;
  const int G8 = gridDim.x >> 3;
  const int xcd = bid & 7, jb = bid >> 3;
  const int L = a.sk_lanes, cpl = a.n_tiles / L, Ng = G8 / L;
  const int col_lane = jb % L, grp_j = jb / L;
  const int tb0 = (int)((long)a.sk_mtiles * xcd / 8), tb1 = (int)((long)a.sk_mtiles * (xcd + 1) / 8);   // row tiles
  long s0, s1;
  if constexpr (MX2) {
    // cuts anywhere in the second walk, at multiples of four steps in the first one
    const long all = (long)(tb1 - tb0) * cpl * S;
    auto cut = [&](long g) {
      const long s = all * g / Ng;
      int k = (int)(s % S);
      if (k < SH) k &= ~3;
      return s - s % S + k;
    };
    s0 = cut(grp_j);
    s1 = cut(grp_j + 1);
  } else {
    const long steps_b = (long)(tb1 - tb0) * cpl * SQ;
    s0 = steps_b * grp_j / Ng * KQ;
    s1 = steps_b * (grp_j + 1) / Ng * KQ;
  }
  const int k_head = (int)(s0 % S), k_tail = (int)(s1 % S);
  const int t_first = (int)((s0 + S - 1) / S), t_end = (int)(s1 / S);   // whole tiles [t_first, t_end) of the block
  const int n_parts = (k_tail ? 1 : 0) + (t_end - t_first) + (k_head ? 1 : 0);

  const int ld_row = lane >> 2;
  const int ld_chunk = (lane & 3) ^ ((lane >> 3) & 3);
  const unsigned lds_base = (unsigned)(size_t)(XV_AS3 char*)smem;
  const int fr_i = lane & 15;
  const int fr_g = lane >> 4;
  const int w_rd = (col_w + fr_i) * 64 + (fr_g ^ ((fr_i >> 1) & 3)) * 16;
  int w_rho;
  {
    const int rho = wave * 16 + ld_row;
    w_rho = SWAP ? ((rho & 64) | swap_fields(rho & 63)) : rho;
  }

  // per-part state (set at the top of the part loop)
  int m0 = 0, n0 = 0;
  const uint16_t* wtile_hi = nullptr;   // weight row n0, wave-uniform
  const uint16_t* wtile_lo = nullptr;
  const uint8_t* wtile_4 = nullptr;
  const uint8_t* wtile_s = nullptr;    // scales of the residual plane: this column tile's 512-byte pieces, one per block
  const uint8_t* wtile_b = nullptr;    // kPrecFp16Mx2: 4-bit weight image, row n0; and its scales (512 bytes per step)
  const uint8_t* wtile_bs = nullptr;
  const unsigned woff = (unsigned)(w_rho * a.ldw + ld_chunk * 8) * 2u;   // this lane's row / chunk inside the tile
  const unsigned woff4 = MX ? (unsigned)(w_rho * a.ldw4 + ld_chunk * 16) : 0u;
  const unsigned woffb = MX2 ? (unsigned)(w_rho * a.ldw4b + ld_chunk * 16) : 0u;
  bool force_slab = false;             // the next step of the second walk stages a slab whatever its position (part start)
  int islab = 0, rslab = 0;            // slabs issued / begun by the read side in this part
  int ig = 0, ikk = 0, ij = 0, ixslot = 0, iwslot = 0, istep = 0;
  bool force_x = true;
  Grp gi = a.grp[0];
  // Address state of the activation DMA of the current (part, group), kept incremental: the 16-row chunks a wave stages
  // are 128 rows apart (x128 bytes), a K chunk is 64 bytes further along the rows.  (Computed per instruction from
  // scratch - 64-bit multiplies - the scalar code of a step's issue was longer than the issue itself; with the
  // 1.25-pass arithmetic the LOAD segment, not the MFMA segment, sets the step time.)
  const char* xrow_hi = nullptr;   // plane row m0 + shift0 + wave * 16, wave-uniform
  const char* xrow_lo = nullptr;
  int x128 = 0, xhalo = 0;         // bytes: 128 rows; from this wave's first chunk to the halo rows
  unsigned xoff = 0;               // lane part of the address: (ld_row * ld + ld_chunk * 8) elements
  auto bind_group = [&]() __attribute__((always_inline)) {
    const long row0 = (long)(m0 + gi.shift0 + wave * 16) * gi.ld * 2;
    xrow_hi = (const char*)gi.hi + row0;
    xrow_lo = SPLIT ? (const char*)gi.lo + row0 : nullptr;
    x128 = gi.ld * 256;
    xhalo = (TM - wave * 16) * gi.ld * 2;
    xoff = (unsigned)(ld_row * gi.ld + ld_chunk * 8) * 2u;
  };
  auto issue_step = [&]() __attribute__((always_inline)) -> int {
    int n = 0;
    const bool lo_step = MX2 && istep >= SH;   // a step of the second walk: 4-bit planes, 128 columns
    if (ij == 0 || force_x) {
      force_x = false;
      const unsigned st = __builtin_amdgcn_readfirstlane(lds_base + ixslot * XSLOT + wave * 1024);
      const char* xh = xrow_hi + ikk * (kBK * 2);
      const char* xl = SPLIT ? xrow_lo + ikk * (kBK * 2) : nullptr;
      if (wave == 0 && gi.nshift > 1) {  // halo rows TM..TM+15 (only read by displaced offsets)
        glds16_sbase(xh + xhalo, xoff, st + (TM / 16) * 1024);
        if constexpr (SPLIT) glds16_sbase(xl + xhalo, xoff, st + XT + (TM / 16) * 1024);
        n += NPX;
      }
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        glds16_sbase(xh, xoff, st + u * 8192);
        if constexpr (SPLIT) glds16_sbase(xl, xoff, st + XT + u * 8192);
        xh += x128;
        if constexpr (SPLIT) xl += x128;
      }
      n += CH * NPX;
      ixslot = ixslot == 2 ? 0 : ixslot + 1;
    }
    if constexpr (MX2) {
      if (lo_step && ((ij == 0 && (ikk & 1) == 0) || force_slab)) {
        // scales of the activation residuals of this and the next 128-column chunk for the tile rows (+ halo)
        force_slab = false;
        const unsigned sl = lds_base + SLB + (islab % 3) * SLAB_BYTES;
        ++islab;
        const uint8_t* src = gi.lo4s + (long)(m0 + gi.shift0 + wave * 64) * gi.ld4s + (ikk >> 1) * 4;
        const unsigned voff = (unsigned)(lane * gi.ld4s);
        glds4_sbase(src, voff, sl + wave * 256);
        n += 1;
        if (wave == 0) {
          glds4_sbase_lanes16(src + (long)TM * gi.ld4s, voff, sl + TM * 4);
          n += 1;
        }
      }
    }
    {
      const unsigned st = __builtin_amdgcn_readfirstlane(lds_base + WBASE + iwslot * WSLOT + wave * 1024);
      const int wcol = gi.wcol0 + ij * gi.wstride + ikk * kBK;
      if (lo_step) {
        // 4-bit weight tile of the step (64 bytes per row, steps in walk order) into the fp16 half of the slot, and its
        // scales into the SCB ring: slots continue behind the last residual block's, which is still unread when the
        // first two steps of this walk are issued
        if constexpr (MX2) {
          const int t_lo = istep - SH;
          glds16_sbase(wtile_b + t_lo * 64, woffb, st);
          n += 1;
          if (wave < 2) {
            glds4_sbase(wtile_bs + t_lo * 512 + wave * 256, (unsigned)lane * 4u,
                        lds_base + SCB + ((t_lo + (SH >> 2)) % 3) * 512 + wave * 256);
            n += 1;
          }
        }
      } else {
        glds16_sbase(wtile_hi + wcol, woff, st);
        n += 1;
        if constexpr (MX) {
          // The 4-bit residual tile of block b = istep / 4 (8 KiB) travels with the block's second step into the
          // residual half of weight slot b % 3: a ring of its own, three blocks deep.  It is read in the COMPUTE segment
          // of the block's last step (after that step's fp16 MFMAs, into the registers of the fp16 weight fragments),
          // which the step ring could not allow - the other wave group refills a step's slot one barrier interval
          // after its LOAD segment.
          if ((istep & 3) == 1) {
            const unsigned st4 = __builtin_amdgcn_readfirstlane(lds_base + WBASE + ((istep >> 2) % 3) * WSLOT + WT + wave * 1024);
            glds16_sbase(wtile_4 + (istep >> 2) * 64, woff4, st4);
            n += 1;
            if (wave < 2) {   // and its scales: 512 contiguous bytes per (tile, block), already in reading order
              glds4_sbase(wtile_s + (istep >> 2) * 512 + wave * 256, (unsigned)lane * 4u,
                          lds_base + SCB + ((istep >> 2) % 3) * 512 + wave * 256);
              n += 1;
            }
          }
        } else if constexpr (WSPLIT) {
          glds16_sbase(wtile_lo + wcol, woff, st + WT);
          n += 1;
        }
      }
      iwslot = iwslot == 2 ? 0 : iwslot + 1;
    }
    ++istep;
    if (++ij == gi.nshift) {
      ij = 0;
      if (++ikk == gi.ksteps) {
        ikk = 0;
        if (++ig < NG) {
          gi = a.grp[ig];
          bind_group();
        }
      }
    }
    return n;
  };

  // kPrecFp16Mx: scales of the 4-bit copies.  Activations: one power of two per 16-row group of the OUTPUT rows (the
  // producing epilogue recorded max |x| of every 16-row group of the source plane; rows displaced by a time offset
  // belong to the same chunk wherever the output row is valid, and saturate mildly at worst - the term they feed
  // is 2^-11 of the product).  max |x| = 1.m x 2^e  ->  x / 2^(e-2) in [4, 8): the top binade of e2m1 ([4, 6]).
  constexpr int XW = (XF + 3) / 4, WW = WF / 4;
  int xs_b[MX ? XW : 1];          // 2^(e-2) as E8M0 bytes, fragment f in byte f & 3 of word f >> 2
  float xs_f[MX ? XF : 1];        // the same as floats: the divisor of the conversions
  int ws_v[MX ? WW : 1];          // E8M0 scales of this lane's weight-fragment rows, fragment w in byte w & 3 of word w >> 2
  typedef __attribute__((ext_vector_type(MX ? XF : 4))) unsigned xs_uvec;
  auto xs_request = [&](const unsigned* gmax, xs_uvec& g) __attribute__((always_inline)) {   // scalar load, not waited for
    if constexpr (MX) {
      const unsigned* p = gmax + ((m0 + row_w) >> 4);
      if constexpr (XF == 8) asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(g) : "s"(p) : "memory");
      else asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=&s"(g) : "s"(p) : "memory");
    }
  };
  auto xs_finish = [&](xs_uvec& g) __attribute__((always_inline)) {
    if constexpr (MX) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(g) : : "memory");
#pragma unroll
      for (int h = 0; h < XW; ++h) xs_b[h] = 0;
#pragma unroll
      for (int f = 0; f < XF; ++f) {
        unsigned e = (g[f] >> 23) & 255u;
        e = e < 16u ? 16u : (e > 200u ? 200u : e);   // empty / all-zero groups, Inf / NaN maxima
        xs_b[f >> 2] |= (int)((e - 2u) << (8 * (f & 3)));
        xs_f[f] = __builtin_bit_cast(float, (e - 2u) << 23);
      }
    }
  };
  auto load_xscales = [&](const unsigned* gmax) __attribute__((always_inline)) {
    xs_uvec g;
    xs_request(gmax, g);
    xs_finish(g);
  };

  int rg = 0, rkk = 0, rj = 0, rxslot = 0, rwslot = 0;
  int r_nshift = 1, r_ksteps = 1, r_dstep = 0;
  struct Frags {
    s16x8 xh[XF], xl[SPLIT ? XF : 1], wh[WF], wl[(WSPLIT && !MX) ? WF : 1];
  };
  int rblk = 0;   // kPrecFp16Mx: block (of four steps) the read side is in
  bool xs_reload = false;   // the previous step was the last one of its group: fetch the next group's scales
  auto read_step = [&](Frags& f) __attribute__((always_inline)) {
    if constexpr (MX) {
      if (xs_reload) {   // a block of four steps never straddles two groups
        load_xscales(a.grp[rg].gmax);
        xs_reload = false;
      }
    }
    const char* xs = smem + rxslot * XSLOT;
    const char* ws = smem + WBASE + rwslot * WSLOT;
    const int row = row_w + fr_i + rj * r_dstep;  // displaced by the time offset of this step
    const int x_rd = row * 64 + (fr_g ^ ((row >> 1) & 3)) * 16;
#pragma unroll
    for (int i = 0; i < XF; ++i) {
      f.xh[i] = *(const s16x8*)(xs + x_rd + i * 1024);
      if constexpr (SPLIT) f.xl[i] = *(const s16x8*)(xs + x_rd + XT + i * 1024);
    }
#pragma unroll
    for (int i = 0; i < WF; ++i) {
      f.wh[i] = *(const s16x8*)(ws + w_rd + i * 1024);
      if constexpr (WSPLIT && !MX) f.wl[i] = *(const s16x8*)(ws + w_rd + WT + i * 1024);
    }
    rwslot = rwslot == 2 ? 0 : rwslot + 1;
    if (++rj == r_nshift) {
      rj = 0;
      rxslot = rxslot == 2 ? 0 : rxslot + 1;
      if (++rkk == r_ksteps) {
        rkk = 0;
        if (++rg < NG) {
          r_nshift = a.grp[rg].nshift;
          r_ksteps = a.grp[rg].ksteps;
          r_dstep = a.grp[rg].dstep;
          if constexpr (MX) xs_reload = rg < a.ngrp;   // the COMPUTE segment of THIS step still converts with the old group's scales
        }
      }
    }
  };

  // [64 x 64 block][p][q] as in the other variants; block h = rows h * 64.. of a 128 x 64 wave tile, or columns h * 64..
  // of a wide one.  XI / WI: activation / weight fragment behind index (h, v)
  f32x4 acc[NH][4][4];
  auto XI = [](int h, int v) __attribute__((always_inline)) { return WIDE ? v : h * 4 + v; };
  auto WI = [](int h, int v) __attribute__((always_inline)) { return WIDE ? h * 4 + v : v; };
  // The MFMAs of a step.  kPrecFp16Mx issues them from inline asm with the accumulator tied in place: left to itself
  // hipcc (ROCm 7.2) un-ties destination and addend once the residual MFMAs sit in a branch of the loop, rotates the
  // 128 accumulator registers through copies and spills 150-400 dwords per lane.  The asm is volatile, so the MFMAs stay
  // in program order inside their COMPUTE segment (plain VALU work, the conversions, may still be scheduled between
  // them); what the compiler no longer does for these
  // instructions - wait states between an MFMA and a VALU instruction that reads its result or writes its operands - is
  // done by hand where it can occur: before the residual MFMAs (their operands come from v_cvt) and after the K loop.
  auto mfmas = [&](const Frags& f) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if constexpr (MX) {
            if constexpr (SWAP) mfma16_f16_inplace(f.wh[WI(h, p)], f.xh[XI(h, q)], acc[h][p][q]);
            else mfma16_f16_inplace(f.xh[XI(h, p)], f.wh[WI(h, q)], acc[h][p][q]);
          } else if constexpr (SWAP) {
            if constexpr (WSPLIT) acc[h][p][q] = mfma16<F16>(f.wl[WI(h, p)], f.xh[XI(h, q)], acc[h][p][q]);
            if constexpr (SPLIT) acc[h][p][q] = mfma16<F16>(f.wh[WI(h, p)], f.xl[XI(h, q)], acc[h][p][q]);
            acc[h][p][q] = mfma16<F16>(f.wh[WI(h, p)], f.xh[XI(h, q)], acc[h][p][q]);
          } else {
            if constexpr (SPLIT) acc[h][p][q] = mfma16<F16>(f.xl[XI(h, p)], f.wh[WI(h, q)], acc[h][p][q]);
            if constexpr (WSPLIT) acc[h][p][q] = mfma16<F16>(f.xh[XI(h, p)], f.wl[WI(h, q)], acc[h][p][q]);
            acc[h][p][q] = mfma16<F16>(f.xh[XI(h, p)], f.wh[WI(h, q)], acc[h][p][q]);
          }
        }
      }
    }
  };
  // kPrecFp16Mx: 4-bit copies of the activation fragments of the four steps of a block (dword s = step s), and the
  // block's residual MFMAs.  Lane (i, g) of a 16x16x128 operand holds K = 32 g .. 32 g + 31 of row i; here these are
  // the columns 8 g .. 8 g + 7 of each of the four steps - the residual plane is packed to match (kernels.h).
  i32x4 x4[MX ? XF : 1];
  auto convert = [&](const Frags& f, const int s) __attribute__((always_inline)) {   // s is a constant at every call site
    if constexpr (MX) {
#pragma unroll
      for (int i = 0; i < XF; ++i) {
        x4[i][s] = cvt8_fp4(f.xh[i], xs_f[i], x4[i][s]);
        // pin the conversion to its own step: hipcc otherwise sinks the conversions of a whole block in front of the
        // residual MFMAs and keeps the fp16 fragments of four steps (96 registers) alive for it
        asm volatile("" : "+v"(x4[i]));
      }
    }
  };
  auto mx_mfmas = [&](Frags& f) __attribute__((always_inline)) {
    if constexpr (MX) {
      // the 4-bit weight fragments of the block go into the registers of the fp16 weight fragments (dead by now)
      const char* w4s = smem + WBASE + (rblk % 3) * WSLOT + WT + w_rd;
#pragma unroll
      for (int w = 0; w < WF; ++w) f.wh[w] = *(const s16x8*)(w4s + w * 1024);
      // (scales of the block, staged with its residual tile: ws_v, read in the LOAD segment - read_scales)
      asm volatile("s_nop 4" ::: "memory");   // v_cvt results -> MFMA operands
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        static_for<0, 4>([&](auto P) {
          static_for<0, 4>([&](auto Q) {
            constexpr int p = decltype(P)::value, q = decltype(Q)::value;
            // operand A of the instruction = first argument; its selector picks the byte of the scale word
            if constexpr (SWAP) mfma_mx4_inplace<p, q>(f.wh[WI(h, p)], x4[XI(h, q)], acc[h][p][q], ws_v[WIDE ? h : 0], xs_b[WIDE ? 0 : h]);
            else mfma_mx4_inplace<p, q>(x4[XI(h, p)], f.wh[WI(h, q)], acc[h][p][q], xs_b[WIDE ? 0 : h], ws_v[WIDE ? h : 0]);
          });
        });
      }
    }
  };
  // kPrecFp16Mx2, a step of the second walk: the fragments just read ARE the 4-bit operands (16 bytes per lane = 32
  // columns of the step's 128); one block-scaled MFMA per fragment pair.  t_lo = index of the step in the walk, (lg, lkk,
  // lj) = its group / 128-column chunk / offset.  Scales: the weight rows' from the SCB ring (one dword per 64-row half),
  // the activation rows' from the group's slab - byte 2 lkk + (fr_g >> 1) of the row (one scale per 64 columns).
  // Weight scales of ring slot `slot` into ws_v: one dword per 64-row half = the bytes of its four fragments for this lane's
  // (fr_i, fr_g); fragment w in byte w & 3 of word w >> 2.  Called in the LOAD segment only: the blocks of the first walk and
  // the steps of the second share the three slots, and the slot of step j + 2 - staged by wave group 0 in its LOAD of step j,
  // beside wave group 1's COMPUTE of step j - 1 - is the slot of step j - 1 (see tdnn_gemm_kernel_v2).
  auto read_scales = [&](const int slot) __attribute__((always_inline)) {
    if constexpr (MX) {
      const char* sc = smem + SCB + slot * 512 + (col_w >> 6) * 256 + (fr_i * 4 + fr_g) * 4;
#pragma unroll
      for (int h = 0; h < WW; ++h) ws_v[h] = *(const int*)(sc + h * 256);
    }
  };
  auto lo_mfmas = [&](Frags& f, const int lg, const int lkk, const int lj, const int slab) __attribute__((always_inline)) {
    if constexpr (MX2) {
      const int (&ws_lo)[MX ? WW : 1] = ws_v;
      const uint8_t* sl = (const uint8_t*)smem + SLB + slab * SLAB_BYTES + (row_w + fr_i + lj * a.grp[lg].dstep) * 4 +
                          2 * (lkk & 1) + (fr_g >> 1);
      int xs_lo = 0;
#pragma unroll
      for (int i = 0; i < XF; ++i) xs_lo |= (int)sl[i * 64] << (8 * i);
      // xs_lo is assembled by VALU instructions and read by the (inline asm) MFMAs as their scale operand: the wait states
      // hipcc would insert for a builtin are ours to provide.  (Without them a build whose scheduling happened to put the
      // last v_or directly in front of the first MFMA lost the bit-identity with the per-tile kernel.)
      asm volatile("s_nop 4" ::: "memory");
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        static_for<0, 4>([&](auto P) {
          static_for<0, 4>([&](auto Q) {
            constexpr int p = decltype(P)::value, q = decltype(Q)::value;
            if constexpr (SWAP) mfma_mx4_inplace<p, q>(f.wh[WI(h, p)], f.xh[XI(h, q)], acc[h][p][q], ws_lo[WIDE ? h : 0], xs_lo);
            else mfma_mx4_inplace<p, q>(f.xh[XI(h, p)], f.wh[WI(h, q)], acc[h][p][q], xs_lo, ws_lo[WIDE ? h : 0]);
          });
        });
      }
    }
  };
  auto wait_and_barrier = [&](int n) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    wait_vm_lgkm0_barrier_n(n);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto plain_barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // position of K step k in the (group, chunk, offset) walk
  auto seek = [&](int k, int& g, int& kk, int& jj) __attribute__((always_inline)) {
    g = 0;
    for (;;) {
      const int n = a.grp[g].ksteps * a.grp[g].nshift;
      if (k < n || g + 1 >= NG) break;
      k -= n;
      ++g;
    }
    const int ns = a.grp[g].nshift;
    kk = k / ns;
    jj = k - kk * ns;
  };

  constexpr long kPartialFloats = (long)TM * kBN;
  constexpr int kAuxCoherent = 1 | 16;   // sc0 sc1: performed at system scope, no cache keeps a copy
   // one workspace slot: the raw accumulators of a tile
  // Sets the issue / read state to the first K step of part `part` and puts the DMA of its first two steps in flight.
  // Called for part i+1 BEFORE the epilogue of part i, so that the pipeline fill (DMA latency) of a part hides behind the
  // previous part's epilogue; the first wait of a part therefore drains everything (vmcnt(0): the epilogue's stores were
  // issued after these DMA instructions and the counter does not tell them apart).
  int kind = 0, n_steps = 0;   // kind 0: whole tile, 1: tail part (accumulators -> workspace), 2: head part
  int n_hi = 0, rstep = 0;     // steps of the part that belong to the first walk; step the read side is at
  auto open_part = [&](int part) __attribute__((always_inline)) {
    // part order: tail (first K steps of the range's last tile), whole tiles, head (last K steps of its first tile)
    int tile, kb, ke;
    if (k_tail && part == 0) {
      tile = t_end; kb = 0; ke = k_tail; kind = 1;
    } else {
      const int w = part - (k_tail ? 1 : 0);
      if (w < t_end - t_first) {
        tile = t_first + w; kb = 0; ke = S; kind = 0;
      } else {
        tile = t_first - 1; kb = k_head; ke = S; kind = 2;
      }
    }
    const int mt = tb0 + tile / cpl, nt = col_lane * cpl + tile % cpl;   // tile = index in this lane's walk
    m0 = mt * TM;
    n0 = nt * kBN;
    wtile_hi = a.w_hi + (long)n0 * a.ldw;
    wtile_lo = (WSPLIT && !MX) ? a.w_lo + (long)n0 * a.ldw : nullptr;
    seek(kb, ig, ikk, ij);
    gi = a.grp[ig];
    bind_group();
    rg = ig; rkk = ikk; rj = ij;
    r_nshift = gi.nshift; r_ksteps = gi.ksteps; r_dstep = gi.dstep;
    ixslot = iwslot = rxslot = rwslot = 0;
    istep = kb;
    rblk = kb >> 2;
    rstep = kb;
    xs_reload = false;
    force_x = true;
    force_slab = MX2 && kb >= SH;
    islab = rslab = 0;
    n_steps = ke - kb;
    n_hi = kb >= SH ? 0 : (ke < SH ? ke : SH) - kb;
    if (EPI != kEpiSplitK && kind != 1 && wave < 6) {
      // bias / scale / offset of the tile's columns -> LDS buffer part & 1, one 256-byte piece per wave; complete with the
      // first wait of the part, read by its epilogue (which runs after open_part of the NEXT part: the other buffer).
      // Without BatchNorm the scale / offset pieces are the constants 1 / 0 (the epilogues read them unconditionally).
      if (wave < 2 || a.bn) {
        const float* src = (wave < 2 ? a.bias : wave < 4 ? a.scale : a.offset) + n0 + (wave & 1) * 64;
        glds4_sbase(src, (unsigned)lane * 4u, lds_base + PB + (part & 1) * 1536 + wave * 256);
      } else {
        *(float*)(smem + PB + (part & 1) * 1536 + wave * 256 + lane * 4) = wave < 4 ? 1.f : 0.f;
      }
    }
    xs_uvec xg;
    if constexpr (MX) {
      wtile_4 = a.w4 + (long)n0 * a.ldw4;
      wtile_s = a.w4_scale + (long)nt * (SH >> 2) * 512;
      if constexpr (MX2) {
        wtile_b = a.w4b + (long)n0 * a.ldw4b;
        wtile_bs = a.w4b_scale + (long)nt * (S - SH) * 512;
      }
      // the activation scales travel (scalar cache) while the DMA of the first steps is issued
      if (n_hi) xs_request(gi.gmax, xg);
    }
    issue_step();
    if (n_steps > 1) issue_step();
    if (n_hi) xs_finish(xg);
  };

  open_part(0);
#pragma nounroll
  for (int part = 0; part < n_parts; ++part) {
    if (kind == 2) {
      // accumulators the previous workgroup of this block left for this tile (its first action).  The workspace is
      // fine-grained (coherent) device memory accessed with cache-bypassing loads / stores, the flag a relaxed
      // agent-scope atomic: acquire / release FENCES at agent scope would write back and invalidate the whole L2 of
      // the XCD once per wave (measured ~70 us per launch).  The wait is bounded: a flag that does not arrive
      // within ~2 s (a lost predecessor: the protocol only ever waits for a lower-numbered workgroup whose first
      // action is the store waited for) raises the launch's error word, which the host checks (sk_check_error), and
      // the workgroup goes on with whatever the slot holds instead of hanging the GPU.
      const int prev = bid - 8 * L;   // same lane, previous group
      if (tid == 0) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__hip_atomic_load(a.sk_flags + prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.sk_epoch) {
          __builtin_amdgcn_s_sleep(8);
          if (__builtin_readcyclecounter() - t0 > 4000000000ull) {
            __hip_atomic_store(a.sk_error, a.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      __syncthreads();
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.sk_ws + (long)prev * kPartialFloats), 0, (int)(kPartialFloats * 4), 0x00020000);
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[h][p][q] = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, ((h * 4 + p) * 4 + q) * 8192, kAuxCoherent));
    } else {
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[h][p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // Ping-pong schedule of v2 (two barrier intervals per K step; group 1 runs one interval behind group 0).  A
    // one-barrier-per-step form (group 0: LOAD_j, COMPUTE_j; group 1: COMPUTE_{j-1}, LOAD_j) was measured 6 % slower:
    // without the second barrier the groups drift into loading at the same time.
    // (open_part leaves no compiler-tracked memory load behind: a vmcnt(0) of hipcc's in front of a first use INSIDE
    // the K loop would drain the LDS-DMA queue on every pass.  The exchange loads above ARE tracked, and the wait inside
    // wait_and_barrier is inline asm hipcc does not look into: without the builtin wait here - the same instruction, but
    // one its scoreboard sees - the first COMPUTE segment of every four-step block carried vmcnt(31) ... vmcnt(0) in front
    // of the MFMAs that first touch each accumulator, i.e. a full drain of the LDS-DMA queue once per block.)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) (expcnt / lgkmcnt untouched)
    wait_and_barrier(0);               // the first two steps have landed (and the previous epilogue's stores are out)
    if (group == 1) plain_barrier();
    Frags f;
    const int ns = n_steps;
    const int ns_hi = MX2 ? n_hi : ns;
    if constexpr (MX) {
      // Blocks of four steps (unrolled: the position inside a block is a compile-time constant): every COMPUTE segment
      // converts its activation fragments into one dword of the 4-bit fragments, the last one also fetches the block's
      // 4-bit weight fragments and issues the 32 residual MFMAs.
#pragma nounroll
      for (int j = 0; j < ns_hi; j += 4, ++rblk) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          // with 1.25 passes per product the LOAD segment (12 fragment reads, 3-6 LDS-DMA instructions) is as long as
          // the COMPUTE segment of the partner wave (s_memtime stamps: ~870 vs ~890 cycles): it gets the issue
          // priority (measured -4.5 % on tdnn2 against the opposite assignment the two-pass kernels use)
          XV_FUZZ();
          __builtin_amdgcn_s_setprio(1);
          read_step(f);
          if (s == 3) read_scales(rblk % 3);
          XV_FUZZ();
          const int n = (j + s + 2 < ns) ? issue_step() : 0;
          wait_and_barrier(n);
          XV_FUZZ();
          __builtin_amdgcn_s_setprio(0);
          mfmas(f);
          convert(f, s);
          if (s == 3) mx_mfmas(f);
          plain_barrier();
        }
      }
      if constexpr (MX2) {
        rstep += ns_hi;
#pragma nounroll
        for (int j = ns_hi; j < ns; ++j, ++rstep) {   // the second walk: one 128-column step = 32 block-scaled MFMAs
          __builtin_amdgcn_s_setprio(1);
          const int lg = rg, lkk = rkk, lj = rj;
          if ((lj == 0 && (lkk & 1) == 0) || (j == ns_hi && ns_hi == 0)) ++rslab;   // the rule of issue_step
          read_step(f);
          read_scales((rstep - SH + (SH >> 2)) % 3);
          XV_FUZZ();
          const int n = (j + 2 < ns) ? issue_step() : 0;
          wait_and_barrier(n);
          XV_FUZZ();
          __builtin_amdgcn_s_setprio(0);
          lo_mfmas(f, lg, lkk, lj, (rslab - 1) % 3);
          plain_barrier();
        }
      }
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // last MFMA results -> the epilogue's VALU reads
    } else {
#pragma nounroll
      for (int j = 0; j < ns; ++j) {
        XV_FUZZ();
        read_step(f);
        XV_FUZZ();
        const int n = (j + 2 < ns) ? issue_step() : 0;
        wait_and_barrier(n);
        XV_FUZZ();
        __builtin_amdgcn_s_setprio(1);
        mfmas(f);
        __builtin_amdgcn_s_setprio(0);
        plain_barrier();
      }
    }
    if (group == 0) plain_barrier();
    XV_FUZZ();
    // every wave is past its last LDS read: the rings may be refilled for the next part while this one's results go out
    const int e_kind = kind, e_m0 = m0, e_n0 = n0;
    const float* e_par = (const float*)(smem + PB + (part & 1) * 1536);
    // Where the planes epilogue also emits the 4-bit residual (kPrecFp16Mx2) its first 64 x 64 half runs BEFORE the next
    // part is opened: with all 128 accumulators still live, the next part's walk state on top of the residual code's
    // temporaries spilled 24 registers; the pipeline fill of the next part then hides behind the second half only.
    constexpr bool SPLIT_EPI = WIDE && EPI == kEpiAct && PrecEmitsLo4(PREC);
    float gm_carry[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (SPLIT_EPI) {
      if (kind != 1) {
        EpiRegs er;
        epilogue_prefetch_lds<EPI, true>(a, e_par, e_m0 + row_w, 0, lane, er);
        gemm_epilogue<PREC, EPI, true>(a, acc[0], e_m0 + row_w, e_n0, lane, er, gm_carry, 1, e_par, 0);
      }
    }
    if (part + 1 < n_parts) open_part(part + 1);

#ifdef XVEC_CLOCK_PROBE
    const unsigned long long xp_e0 = __builtin_readcyclecounter();
#endif
    if (e_kind == 1) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.sk_ws + (long)bid * kPartialFloats), 0, (int)(kPartialFloats * 4), 0x00020000);
#pragma unroll
      for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[h][p][q]), rs, tid * 16,
                                                   ((h * 4 + p) * 4 + q) * 8192, kAuxCoherent);
      // every store of this workgroup has been performed (acknowledged) before the flag goes out
      XV_FUZZ();
      __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0), through the builtin so that hipcc's scoreboard knows
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(a.sk_flags + bid, a.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if constexpr (WIDE) {
        // two 64 x 64 blocks side by side: same rows, columns e_n0 and e_n0 + 64.  The group maxima of the planes
        // epilogue are carried across both (one reduction and one atomic per 16-row group and wave).
        float (&gm)[4] = gm_carry;
#pragma unroll
        for (int h = SPLIT_EPI ? 1 : 0; h < NH; ++h) {
          constexpr bool LAZY = (EPI == kEpiAct || EPI == kEpiF32) && MX;   // where the registers are short (spills)
          EpiRegs er;
          epilogue_prefetch_lds<EPI, LAZY>(a, e_par, e_m0 + row_w, h * 64, lane, er);
          gemm_epilogue<PREC, EPI, LAZY>(a, acc[h], e_m0 + row_w, e_n0 + h * 64, lane, er, gm, h == 0 ? 1 : 2, e_par, h * 64);
        }
      } else {
        EpiRegs er;
        epilogue_prefetch_lds<EPI>(a, e_par, e_m0 + row_w, col_w, lane, er);
        float gm[4] = {0.f, 0.f, 0.f, 0.f};
        gemm_epilogue<PREC, EPI>(a, acc[0], e_m0 + row_w, e_n0 + col_w, lane, er, gm, 0);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// v5 ("p8"): 256 x 256 output tiles, K tiles of 64 columns, phase-interleaved (shapes of
// egs/sre/v2/local/nnet3/xvector/run_xvector_new.sh:96-99).  Standalone study of the schedule: tools/probe_p8.hip.
//
//  * 8 waves as 4 (frames) x 2 (columns): a wave owns 64 frames x 128 columns = two 64 x 64 blocks side by side, the
//    accumulator layout of the wide stream-K kernel, so the epilogues are shared.  Wave group = column half (waves 0-3 / 4-7,
//    one of each per SIMD).
//  * LDS: two K-tile buffers, each an activation tile (256 frames x 128 B) and a weight tile (256 rows x 128 B, rows
//    ordered [64-column block h][column half][64 rows]), laid out [frames 0 | frames 1 | weights 0 | weights 1]: rows of 128 bytes = whole cache lines per row piece for the
//    LDS-DMA (the 64-byte row pieces of the 32-column K steps are half lines: twice the requests per byte).  16-byte chunks
//    XOR-swizzled with three row bits (chunk ^ ((row >> 1) & 7)): every ds_read_b128 lane group touches 16 distinct slots
//    (SQ_LDS_BANK_CONFLICT = 0, profiles/r04_probe_p8_pmc.txt); applied to the per-lane SOURCE address of the DMA.
//  * A K tile is four phases of 16 MFMAs; a phase = LOAD part (the phase's fragment reads, ONE staging unit of 16 KiB = two
//    LDS-DMA instructions per wave, a counted wait where data is about to be needed), barrier, MFMA part, barrier; wave
//    group 1 runs one barrier interval behind group 0, so that on every SIMD one wave multiplies while the other loads.
//      phase 0: reads weight fragments h = 0 (8) + frame fragments 0, 1 (4)   MFMAs h = 0 x frames 0, 1
//      phase 1: reads frame fragments 2, 3 (4)                               MFMAs h = 0 x frames 2, 3
//      phase 2: reads weight fragments h = 1 (8)                             MFMAs h = 1 x frames 2, 3
//      phase 3: -                                                            MFMAs h = 1 x frames 0, 1
//    Staging units and their issue (tile t, phase p): (t,0) frames 128..255 of tile t+1; (t,1) weights h = 1 of t+1;
//    (t,2) weights h = 0 of t+2; (t,3) frames 0..127 of t+2 - each at least two phases after the last read of the bytes it
//    overwrites and four to six phases before its first read.  Waits: end of LOAD(t,1) for the weights h = 1 of tile t (four
//    units stay in flight), end of LOAD(t,3) for tile t+1's phase-0 data (three units in flight): never vmcnt(0) inside a part.
//  * The time offsets of a spliced layer are part of the K-tile ADDRESS (rows m0 + offset): the fragment reads never move.
//    K walk: group -> 64-column chunk -> offset (PlanWalkSteps64); accumulation order therefore differs from the 32-column
//    kernels: a layer runs either family for ALL its launches of a mode (GemmArgs::p8), never by launch size.
//  * kPrecFp16Mx: a block = two K tiles.  The conversions of the frame fragments to e2m1 ride the MFMA parts of phases 0 / 1;
//    the block's 4-bit weight tile (16 KiB, rows of 64 B, swizzled like the 32-column tiles) and its scales (1 KiB) are
//    DMA'd in phase 1 of the block's first tile into a buffer of their own, read at the end of the block's last MFMA part into
//    the registers of the fp16 weight fragments, and followed by the 32 block-scaled MFMAs.
//  * Persistent grid, K tiles dealt out evenly in pairs, partial tiles exchanged through the stream-K workspace: the
//    partition, the exchange and its ordering guarantees are those of tdnn_gemm_kernel_sk (bit-identical whatever the cut).
constexpr int kP8Buf = 65536, kP8XW = 32768, kP8Unit = 16384;
constexpr int kP8W4 = 2 * kP8Buf;       // 4-bit residual tile of the current block: 256 rows x 64 B
constexpr int kP8SC = kP8W4 + 16384;    // its scales: two 128-column tiles x 512 B
constexpr int kP8PB = kP8SC + 1024;     // epilogue parameters: [part & 1][column half][bias | scale | offset][128 floats]
constexpr int kP8WSL = kP8PB + 2 * 3072;   // kPrecFp16Mx2, second walk: weight scales [tile & 1][128-column half of the K tile][128-column tile][512]
constexpr int kP8XSL = kP8WSL + 2 * 2048;  // and the scales of the activation residuals, [tile & 1][frame][4]: one byte per 64 columns
constexpr int kP8Lds = kP8XSL + 2 * 1024;

// s_waitcnt vmcnt(n) for a wave-uniform n (the counted waits of tdnn_gemm_kernel_p8: what is in flight behind the data a wait is
// for depends on the walk - 4-bit tiles, scale pieces, the tail of a part - and is counted as it is issued)
__device__ __forceinline__ void wait_vmcnt_n(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
  }
}

#ifdef XVEC_CLOCK_PROBE
// Measurement build only (tools/clock_probe.py; never defined in the shipped library): the shader clock the chip holds while
// tdnn_gemm_kernel_p8 runs, from inside the kernel - s_memtime ticks (shader cycles) over s_memrealtime ticks (a constant
// 100 MHz counter) between a workgroup's first and last instruction.  Slot = launch kind (0: act, <= 8 K tiles; 1: act, more;
// 2: statistics epilogue), two words per workgroup.
__device__ unsigned long long g_clock_probe[3 * 512 * 2];
// [slot][workgroup][0: exchange store (issue -> vmcnt(0) + barrier), 1: flag wait + exchange loads issued, 2: drain at the start of the
// head part (the loads landing), 3: the epilogue of the workgroup's last whole-tile part] in shader cycles, wave 0
__device__ unsigned long long g_exchange_probe[3 * 512 * 4];
// [slot][workgroup][0: cycles wave 0 spent in the wait + barrier that opens its whole-tile parts (the DMA of the part's first tiles
// landing AND the previous epilogue's stores draining: the counter does not tell them apart), summed over the launch, 1: such parts]
__device__ unsigned long long g_part_probe[3 * 512 * 2];
extern "C" int xvec_part_probe_read(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_part_probe), sizeof(g_part_probe), 0, hipMemcpyDeviceToHost);
}
extern "C" int xvec_clock_probe_read(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clock_probe), sizeof(g_clock_probe), 0, hipMemcpyDeviceToHost);
}
extern "C" int xvec_exchange_probe_read(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_exchange_probe), sizeof(g_exchange_probe), 0, hipMemcpyDeviceToHost);
}
#endif

template <int PREC, int EPI>
__global__ __launch_bounds__(512) void tdnn_gemm_kernel_p8(const GemmArgs a) {
  static_assert(PREC == kPrecFp16 || PREC == kPrecFp16Mx || PREC == kPrecFp16Mx2 || PREC == kPrecFp16MxE,
                "single-pass fp16, the 1.25-pass arithmetic (with or without the residual plane of its output), the 1.5-pass arithmetic");
  constexpr bool MX = PrecMx(PREC);
  constexpr bool MX2 = PrecMx2(PREC);
  constexpr bool SWAP = (EPI != kEpiStats);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;   // frames wm * 64.., columns wn * 128..; wave group = wn
  const int fr_i = lane & 15, fr_g = lane >> 4;
  const int bid = blockIdx.x;
  const unsigned lds_base = (unsigned)(size_t)(XV_AS3 char*)smem;
  XV_FUZZ_INIT();

  // ---- this workgroup's share of the K tiles (the partition of tdnn_gemm_kernel_sk) ----------------------------------------
  // kPrecFp16Mx2: behind the S tiles of 64 fp16 columns an output tile's walk goes on with S_lo tiles of 256 4-bit columns
  // (128 B per row: the same LDS image, DMA and fragment reads) over the activations' residual planes and the 4-bit image of the
  // weights.  Cuts: at even tiles inside the first walk (a block of the residual product is a tile pair), anywhere in the second.
  const int S = a.p8_ktiles;                   // K tiles of the first walk (even)
  const int S_lo = MX2 ? a.p8_ktiles_lo : 0;
  const int ST = S + S_lo;
  const int NG = a.ngrp + (MX2 ? a.ngrp_lo : 0);
  const int G8 = gridDim.x >> 3;
  const int xcd = bid & 7, jb = bid >> 3;
  const int L = a.sk_lanes, NT = a.n_tiles >> 1, cpl = NT / L, Ng = G8 / L;
  const int col_lane = jb % L, grp_j = jb / L;
  const int tb0 = (int)((long)a.sk_mtiles * xcd / 8), tb1 = (int)((long)a.sk_mtiles * (xcd + 1) / 8);
  long s0, s1;
  {
    const long all = (long)(tb1 - tb0) * cpl * ST;
    auto cut = [&](long g) __attribute__((always_inline)) {
      const long s = all * g / Ng;
      // whole tiles only (GemmArgs::p8_whole): no partial tile, no exchange through the workspace - the shares then differ by one tile
      if (a.p8_whole) return (s + ST / 2) / ST * ST;
      int k = (int)(s % ST);
      if (k < S) k &= ~1;
      return s - s % ST + k;
    };
    s0 = cut(grp_j);
    s1 = cut(grp_j + 1);
  }
  const int k_head = (int)(s0 % ST), k_tail = (int)(s1 % ST);
  const int t_first = (int)((s0 + ST - 1) / ST), t_end = (int)(s1 / ST);
  // (the launcher sizes the grid so that every share is at least one whole tile: a head and a tail never meet in one tile)
  const int n_parts = (k_tail ? 1 : 0) + (t_end - t_first) + (k_head ? 1 : 0);
  if (s1 <= s0) return;   // an XCD block without row tiles (launches of fewer than eight of them)
#ifdef XVEC_CLOCK_PROBE
  const unsigned long long clk_t0 = __builtin_readcyclecounter(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- staging geometry: a wave stages rows wave * 16 + piece * 8 + (lane >> 3) of every 128-row unit; the lane fetches the
  // logical 16-byte chunk (lane & 7) ^ ((piece * 4 + (lane >> 4)) & 7) of its row.  The per-lane byte offsets of the units live
  // in registers (six of them).  Only the 1.5-pass instantiation, which has none to spare, recomputes them where a unit is
  // issued (RECOMPUTE; from an opaque copy of the lane id, or hipcc hoists them back): measured on the 1.25-pass kernel that
  // costs 13 % (tdnn2 0.186 -> 0.210 ms) - the LOAD part of a phase has no room for two dozen more VALU instructions.
  constexpr bool RECOMPUTE = MX2;
  auto lane_now = [&]() __attribute__((always_inline)) {
    int l = lane;
    if constexpr (RECOMPUTE) asm volatile("" : "+v"(l));
    return l;
  };
  auto unit_offsets = [&](const int ld_elems, const bool weights, unsigned (&off)[2]) __attribute__((always_inline)) {
    const int l = lane_now();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ru = wave * 16 + j * 8 + (l >> 3);
      const int c = (l & 7) ^ ((j * 4 + (l >> 4)) & 7);
      int row = ru;
      if (weights) {
        const int rho = ru & 63;
        row = (ru >> 6) * 128 + (SWAP ? swap_fields(rho) : rho);
      }
      off[j] = (unsigned)(row * ld_elems + c * 8) * 2u;
    }
  };
  auto w4_offsets = [&](unsigned (&off)[2]) __attribute__((always_inline)) {   // 4-bit tile: rows wave * 32 + u * 16 + (lane >> 2), 64 B each
    const int l = lane_now();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r4 = wave * 32 + u * 16 + (l >> 2);
      const int rho = r4 & 63;
      const int wrow = ((r4 >> 6) & 1) * 128 + (r4 >> 7) * 64 + (SWAP ? swap_fields(rho) : rho);
      off[u] = (unsigned)(wrow * a.ldw4 + ((l & 3) ^ ((l >> 3) & 3)) * 16);
    }
  };
  unsigned woff_r[2] = {0u, 0u}, woff4_r[2] = {0u, 0u}, xoff_r[2] = {0u, 0u};
  if constexpr (!RECOMPUTE) {
    unit_offsets(a.ldw, true, woff_r);
    if constexpr (MX) w4_offsets(woff4_r);
  }
  const unsigned st_lane = lds_base + wave * 2048;

  // ---- per-part state ----------------------------------------------------------------------------------------------------
  int m0 = 0, n0 = 0, nt = 0;
  int kind = 0, n_tiles_part = 0, kb_part = 0;
  // issue side: position in the walk (tile index, group, chunk, offset) and the addresses of its K tile
  int it = 0, ig = 0, ic = 0, ij = 0;
  Grp gi = a.grp[0];
  const char* xb = nullptr;
  const char* wb = nullptr;
  const char* wtile = nullptr;          // weight row n0
  int x_next = 0, x_wrap = 0, w_next = 0, w_wrap = 0;   // byte steps of the walk (they fit 32 bits; fewer scalar registers)
  auto bind_group = [&]() __attribute__((always_inline)) {
    if constexpr (!RECOMPUTE) unit_offsets(gi.ld, false, xoff_r);
    xb = (const char*)gi.hi + ((long)(m0 + gi.shift0 + ij * gi.dstep) * gi.ld + ic * 64) * 2;
    x_next = gi.dstep * gi.ld * 2;
    x_wrap = 128 - (gi.nshift - 1) * gi.dstep * gi.ld * 2;
    // second walk: the weight tiles follow each other in walk order, 128 bytes per row and tile, in an image with the row pitch of
    // the fp16 plane (GemmArgs::ldw4b); first walk: the tile's columns of the fp16 plane
    const bool lo_i = MX2 && it >= S;
    const char* wb_hi = wtile + (long)(gi.wcol0 + ij * gi.wstride + ic * 64) * 2;
    const char* wb_lo = MX2 ? (const char*)a.w4b + (long)n0 * a.ldw * 2 + (long)(it - S) * 128 : wb_hi;
    wb = lo_i ? wb_lo : wb_hi;
    w_next = lo_i ? 128 : gi.wstride * 2;
    w_wrap = lo_i ? 128 : 128 - (gi.nshift - 1) * gi.wstride * 2;
  };
  auto adv = [&]() __attribute__((always_inline)) {
    ++it;
    if (++ij == gi.nshift) {
      ij = 0;
      xb += x_wrap;
      wb += w_wrap;
      if (++ic == (gi.ksteps >> 1)) {
        ic = 0;
        if (++ig < NG) {
          gi = a.grp[ig];
          bind_group();
        }
      }
    } else {
      xb += x_next;
      wb += w_next;
    }
  };
  // staging units: 0 = weights h = 0, 1 = frames 0..127, 2 = frames 128..255, 3 = weights h = 1
  auto issue = [&](const int unit, const int buf) __attribute__((always_inline)) {
    if (unit == 0 || unit == 3) {
      const unsigned dst = st_lane + 2 * kP8XW + buf * kP8XW + (unit == 3 ? kP8Unit : 0);
      const char* src = wb + (unit == 3 ? (long)a.ldw * 128 : 0);
      unsigned woff[2] = {woff_r[0], woff_r[1]};   // (the second walk's 4-bit weight image has the row pitch of the fp16 plane: the same offsets)
      if constexpr (RECOMPUTE) unit_offsets(a.ldw, true, woff);
      glds16_sbase_m0(src, woff[0], dst);
      glds16_sbase_m0(src, woff[1], dst + 1024);
    } else {
      const char* src = xb + (unit == 2 ? (long)gi.ld * 256 : 0);
      const unsigned dst = st_lane + buf * kP8XW + (unit == 2 ? kP8Unit : 0);
      unsigned xoff[2] = {xoff_r[0], xoff_r[1]};
      if constexpr (RECOMPUTE) unit_offsets(gi.ld, false, xoff);
      glds16_sbase_m0(src, xoff[0], dst);
      glds16_sbase_m0(src, xoff[1], dst + 1024);
    }
  };
  // the 4-bit tile of block `blk` of the tile's walk and its scales: three DMA instructions in EVERY wave (waves 4-7 repeat
  // the scale pieces of waves 0-3: the counted waits are the same in all waves)
  auto issue_w4 = [&](const int blk) __attribute__((always_inline)) {
    if constexpr (MX) {
      const unsigned d4 = lds_base + kP8W4 + wave * 2048;
      unsigned woff4[2] = {woff4_r[0], woff4_r[1]};
      if constexpr (RECOMPUTE) w4_offsets(woff4);
      const uint8_t* w4t = a.w4 + (long)n0 * a.ldw4 + blk * 64;
      glds16_sbase(w4t, woff4[0], d4);
      glds16_sbase(w4t, woff4[1], d4 + 1024);
      const int pc = wave & 3;   // piece: 128-column tile pc >> 1, half pc & 1
      glds4_sbase(a.w4_scale + ((long)(nt * 2 + (pc >> 1)) * (S >> 1) + blk) * 512 + (pc & 1) * 256, (unsigned)lane * 4u,
                  lds_base + kP8SC + pc * 256);
    }
  };
  // kPrecFp16Mx2: the scales of the second-walk tile the issue side stands on, into scale buffer sb: the weights' (2 KiB: two
  // 128-column steps x two 128-column tiles x 512 B, one 256-byte piece per wave) and the activation residuals' (one dword per
  // frame = its four 64-column blocks of the tile; waves 4-7 repeat the pieces of waves 0-3).  Two DMA instructions per wave.
  auto issue_scales = [&](const int sb) __attribute__((always_inline)) {
    if constexpr (MX2) {
      const int t_lo = it - S;
      const int t128 = (wave >> 1) & 1, ks = wave >> 2, half = wave & 1;
      glds4_sbase(a.w4b_scale + ((long)(nt * 2 + t128) * (2 * S_lo) + 2 * t_lo + ks) * 512 + half * 256, (unsigned)lane * 4u,
                  lds_base + kP8WSL + sb * 2048 + ks * 1024 + t128 * 512 + half * 256);
      const uint8_t* xs = gi.lo4s + (long)(m0 + gi.shift0 + ij * gi.dstep + (wave & 3) * 64) * gi.ld4s + ic * 4;
      glds4_sbase(xs, (unsigned)(lane * gi.ld4s), lds_base + kP8XSL + sb * 1024 + (wave & 3) * 256);
    }
  };

  // ---- fragment read geometry ------------------------------------------------------------------------------------------------
  const int sw = (fr_g ^ ((fr_i >> 1) & 7)) * 16;
  const int xrd0 = (wm * 64 + fr_i) * 128 + sw, xrd1 = xrd0 ^ 64;
  const int wrd0 = 2 * kP8XW + (wn * 64 + fr_i) * 128 + sw, wrd1 = wrd0 ^ 64;
  const int w4rd = kP8W4 + (wn * 64 + fr_i) * 64 + (fr_g ^ ((fr_i >> 1) & 3)) * 16;   // + h * 8192 + p * 1024

  // kPrecFp16Mx: scales of the 4-bit copies of the frame fragments (one power of two per 16-row group of the output rows,
  // from the group maxima of the source plane: see tdnn_gemm_kernel_sk)
  int xs_b = 0;             // 2^(e - 2) as E8M0 bytes, fragment f in byte f: the scale operand of the block-scaled MFMAs
  float xs_f[MX ? 4 : 1];   // the same as floats: the divisor of the conversions
  typedef __attribute__((ext_vector_type(4))) unsigned xs_uvec;
  auto xs_request = [&](const unsigned* gmax, xs_uvec& g) __attribute__((always_inline)) {
    if constexpr (MX) {
      const unsigned* p = gmax + ((m0 + wm * 64) >> 4);
      asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=&s"(g) : "s"(p) : "memory");
    }
  };
  auto xs_finish = [&](xs_uvec& g) __attribute__((always_inline)) {
    if constexpr (MX) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(g) : : "memory");
      xs_b = 0;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        unsigned e = (g[f] >> 23) & 255u;
        e = e < 16u ? 16u : (e > 200u ? 200u : e);
        xs_b |= (int)((e - 2u) << (8 * f));
        xs_f[f] = __builtin_bit_cast(float, (e - 2u) << 23);
      }
    }
  };
  int rg = 0, r_left = 0;   // read side: group of the current K tile, tiles left in it

  f32x4 acc[2][4][4];
  s16x8 xf[4][2], wf[4][2];
  i32x4 x4[MX ? 4 : 1];
  int ws_lo[MX2 ? 2 : 1], xs_lo[MX2 ? 2 : 1];   // second walk: scale words, [128-column half]: the weights' of the phase's 64-row block, the frames' of the tile

  auto barrier = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // Issue priority (same-box A/B of round 5 through a runtime knob, profiles/r05_p8_priority.md; compile-time again since).  A
  // barrier interval lasts as long as the LOAD part of the wave that is NOT multiplying (two LDS-DMA instructions at 100-185
  // cycles of issue each + up to twelve fragment reads), so that is the part that gets the issue slots: priority 1 around the
  // LOAD part.  No priority instruction at all: +1.7 % time on this kernel's launches; priority around the MFMA part (the 8-phase
  // template's recipe, round 4's kernel): +4 %; the late wave group at priority 1 throughout (MI355X_MICROARCH.md, two waves
  // per SIMD, item 4): +0.3 %.
  constexpr bool prio_load = true, prio_mfma = false;
  auto mfma = [&](auto HH, auto PP, auto QQ, const int k) __attribute__((always_inline)) {
    constexpr int H = decltype(HH)::value, pw = decltype(PP)::value, qx = decltype(QQ)::value;   // weight fragment pw, frame fragment qx
    // accumulators tied in place (inline asm): left to itself hipcc rotates the 128 accumulator registers through copies
    // across the phases of the loop and spills (see tdnn_gemm_kernel_sk); the waits these instructions need are explicit
    if constexpr (SWAP) mfma16_f16_inplace(wf[pw][k], xf[qx][k], acc[H][pw][qx]);
    else mfma16_f16_inplace(xf[qx][k], wf[pw][k], acc[H][qx][pw]);
  };
  // DMA instructions issued in this phase (c0) and in the three before it: what may still be in flight when a wait is for the
  // data of an older phase.  WAIT = 1 (end of LOAD(t,1)): everything issued up to phase (t-1,1) has landed - the weights h = 1 of
  // tile t, the 4-bit tile of a block that began in t-1; WAIT = 3 (end of LOAD(t,3)): everything up to phase (t,0) - tile t+1's
  // phase-0 data (and the second walk's scales, issued a tile earlier).  Never vmcnt(0) inside a part in the steady state.
  int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  // One phase.  P = phase, B = buffer of the tile, ODD = second tile of its pair (the 4-bit conversions go to dwords 2, 3; the
  // block-scaled MFMAs follow its last phase), LO = a tile of the second walk; unit / ibuf = what the LOAD part stages (unit < 0:
  // nothing), w4_blk >= 0: also the 4-bit tile of that block, sc_buf >= 0: also the second-walk scales of the issue side's tile.
  auto phase = [&](auto PP, auto BB, auto OO, auto LL, auto FF, const int unit, const int ibuf, const int w4_blk, const int sc_buf,
                   const bool steady) __attribute__((always_inline)) {
    constexpr int P = decltype(PP)::value, B = decltype(BB)::value, ODD = decltype(OO)::value;
    constexpr bool LO = decltype(LL)::value != 0;
    // FAST: a phase of the part's steady state (at least two more tiles behind this pair; first walk) - what it stages and what it
    // waits for are compile-time constants: unit {2, 3, 0, 1}[P] into buffer (P < 2 ? 1 - B : B), the block's 4-bit tile in phase 1
    // of its first tile, vmcnt(8 / 6 [+ 3]).  The LOAD part is what a barrier interval waits for (profiles/r05_p8_priority.md),
    // and the runtime form of these decisions (is there a unit? a 4-bit tile? which wait?) cost it a dozen scalar and two vector
    // instructions per phase.
    constexpr bool FAST = decltype(FF)::value != 0;
    // (only ever CALLED for the first walk of the 1- and 1.25-pass arithmetic: pair_fast)
    // LDS: [frames, buffer 0 | frames, buffer 1 | weights, buffer 0 | weights, buffer 1], 32 KiB each: every fragment read is one of
    // four per-lane bases + a 16-bit immediate
    const char* xs0 = smem + xrd0 + B * kP8XW;
    const char* xs1 = smem + xrd1 + B * kP8XW;
    const char* ws0 = smem + wrd0 + B * kP8XW;
    const char* ws1 = smem + wrd1 + B * kP8XW;
    XV_FUZZ();
    if (prio_load) __builtin_amdgcn_s_setprio(1);
    if constexpr (P == 0) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        wf[p][0] = *(const s16x8*)(ws0 + p * 2048);
        wf[p][1] = *(const s16x8*)(ws1 + p * 2048);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        xf[q][0] = *(const s16x8*)(xs0 + q * 2048);
        xf[q][1] = *(const s16x8*)(xs1 + q * 2048);
      }
      if constexpr (MX2 && LO) {
        // the tile's scales into registers for its four phases: weights - one dword per 64-row block = the bytes of its four
        // fragments for this lane's (row, lane group); activations - byte 2 k + (lane group >> 1) of a frame's dword
        const uint8_t* xsl = (const uint8_t*)smem + kP8XSL + B * 1024 + (wm * 64 + fr_i) * 4 + (fr_g >> 1);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          int x = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) x |= (int)xsl[q * 64 + 2 * k] << (8 * q);
          xs_lo[k] = x;
        }
      }
    } else if constexpr (P == 1) {
#pragma unroll
      for (int q = 2; q < 4; ++q) {
        xf[q][0] = *(const s16x8*)(xs0 + q * 2048);
        xf[q][1] = *(const s16x8*)(xs1 + q * 2048);
      }
    } else if constexpr (P == 2) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        wf[p][0] = *(const s16x8*)(ws0 + kP8Unit + p * 2048);
        wf[p][1] = *(const s16x8*)(ws1 + kP8Unit + p * 2048);
      }
    }
    if constexpr (MX2 && LO && (P == 0 || P == 2)) {
      // the weights' scale words of the 64-row block h the next two phases multiply: one dword = the bytes of its four fragments
      const char* wsl = smem + kP8WSL + B * 2048 + wn * 512 + (P >> 1) * 256 + (fr_i * 4 + fr_g) * 4;
      ws_lo[0] = *(const int*)(wsl);
      ws_lo[1] = *(const int*)(wsl + 1024);
    }
    // steady state (every phase of this pair and of the one before it staged its unit): the counts are constants - 2 per phase,
    // + 3 in phase 1 of a block's first tile (4-bit tile and scales); the tail of a part and the second walk count as they go
    constexpr int E1 = (MX && !LO && !ODD) ? 3 : 0;
    XV_FUZZ();
    if constexpr (FAST) {
      constexpr int UNIT = P == 0 ? 2 : P == 1 ? 3 : P == 2 ? 0 : 1;
      issue(UNIT, P < 2 ? 1 - B : B);
      if constexpr (MX && P == 1 && !ODD) issue_w4(w4_blk);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (P == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + E1) : "memory");
      if constexpr (P == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + E1) : "memory");
    } else {
      c3 = c2;
      c2 = c1;
      c1 = c0;
      c0 = 0;
      if (unit >= 0) {
        issue(unit, ibuf);
        c0 += 2;
      }
      if (w4_blk >= 0) {
        issue_w4(w4_blk);
        c0 += 3;
      }
      if (sc_buf >= 0) {
        issue_scales(sc_buf);
        c0 += 2;
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (P == 1) {
        if (!MX2 && steady) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + E1) : "memory");
        else wait_vmcnt_n(c0 + c1 + c2 + c3);
      }
      if constexpr (P == 3) {
        if (!MX2 && steady) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + E1) : "memory");
        else wait_vmcnt_n(c0 + c1 + c2);
      }
    }
    asm volatile("s_barrier" ::: "memory");
    XV_FUZZ();
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the fragments are in (the MFMAs are inline asm: their waits are ours)
    __builtin_amdgcn_sched_barrier(0);
    constexpr int H = P >> 1;
    constexpr int Q0 = (P == 0 || P == 3) ? 0 : 2;
    if (prio_load) __builtin_amdgcn_s_setprio(0);
    if (prio_mfma) __builtin_amdgcn_s_setprio(1);
    if constexpr (MX2 && LO) {
      // 16 block-scaled MFMAs: the fragments just read ARE the 4-bit operands (16 bytes per lane = 32 columns of a 128-column half)
      asm volatile("s_nop 4" ::: "memory");   // the scale words are assembled by VALU instructions
      static_for<0, 2>([&](auto K) {
        static_for<0, 4>([&](auto PW) {
          static_for<Q0, Q0 + 2>([&](auto QX) {
            constexpr int k = decltype(K)::value, pw = decltype(PW)::value, qx = decltype(QX)::value;
            if constexpr (SWAP) mfma_mx4_inplace<pw, qx>(wf[pw][k], xf[qx][k], acc[H][pw][qx], ws_lo[k], xs_lo[k]);
            else mfma_mx4_inplace<qx, pw>(xf[qx][k], wf[pw][k], acc[H][qx][pw], xs_lo[k], ws_lo[k]);
          });
        });
      });
    } else if constexpr (MX && P < 2) {
      // 16 MFMAs and, one behind each, the 16 conversions of the fragments just read (e2m1 dword 2 * ODD + k of the block's
      // 4-bit fragments): left to hipcc twelve of them trail the MFMAs as one chain of dependent quarter-rate instructions
      // (every fragment's four conversions write bytes of one register), ~200 cycles of an MFMA part in which the matrix pipe
      // idles.  Here the four chains (frame fragment x 32-column half) take turns, so dependent ones are eight issues apart.
      static_for<0, 16>([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr int mk = i >> 3, mp = (i >> 1) & 3, mq = Q0 + (i & 1);          // MFMA i: k-half, weight fragment, frame fragment
        mfma(std::integral_constant<int, H>{}, std::integral_constant<int, mp>{}, std::integral_constant<int, mq>{}, mk);
        constexpr int cq = Q0 + (i & 1), ck = (i >> 1) & 1, cj = i >> 2;          // conversion i: chain (cq, ck), its dword cj
        const u32x4 u = __builtin_bit_cast(u32x4, xf[cq][ck]);
        const float xsc = xs_f[cq];
        int d;
        if constexpr (cj == 0) {
          // the first of a dword's four conversions DEFINES the register (the other three bytes follow): the 4-bit fragments are
          // then dead between blocks and through the second walk instead of 16 registers carried around the loop
          asm volatile("v_cvt_scalef32_pk_fp4_f16 %0, %1, %2" : "=v"(d) : "v"(u[0]), "v"(xsc));
        } else {
          d = x4[cq][2 * ODD + ck];
          asm volatile("v_cvt_scalef32_pk_fp4_f16 %0, %1, %2 op_sel:[0,0,%3,%4]"
                       : "+v"(d)
                       : "v"(u[cj]), "v"(xsc), "n"(cj & 1), "n"(cj >> 1));
        }
        x4[cq][2 * ODD + ck] = d;
      });
    } else if constexpr (MX && ODD && P == 3) {
      // last phase of a block: its fp16 MFMAs by 32-column half, the 4-bit weight fragments of h = 0 / h = 1 read into the
      // registers of the half just finished (their latency behind the other half's MFMAs / the first 16 block-scaled ones)
      int ws_v[2];
      const char* sc = smem + kP8SC + wn * 512 + (fr_i * 4 + fr_g) * 4;
      ws_v[0] = *(const int*)(sc);
      ws_v[1] = *(const int*)(sc + 256);
      static_for<0, 4>([&](auto PW) {
        static_for<Q0, Q0 + 2>([&](auto QX) { mfma(std::integral_constant<int, H>{}, PW, QX, 0); });
      });
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int w = 0; w < 4; ++w) wf[w][0] = *(const s16x8*)(smem + w4rd + w * 1024);
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, 4>([&](auto PW) {
        static_for<Q0, Q0 + 2>([&](auto QX) { mfma(std::integral_constant<int, H>{}, PW, QX, 1); });
      });
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int w = 0; w < 4; ++w) wf[w][1] = *(const s16x8*)(smem + w4rd + 8192 + w * 1024);
      asm volatile("s_waitcnt lgkmcnt(4)\n\ts_nop 4" ::: "memory");   // the fragments of h = 0 are in; v_cvt results -> MFMA operands
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, 2>([&](auto HH) {
        constexpr int h = decltype(HH)::value;
        if constexpr (h == 1) {
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_waitcnt(0xc07f);
          __builtin_amdgcn_sched_barrier(0);
        }
        static_for<0, 4>([&](auto PW) {
          static_for<0, 4>([&](auto QX) {
            constexpr int pw = decltype(PW)::value, qx = decltype(QX)::value;
            if constexpr (SWAP) mfma_mx4_inplace<pw, qx>(wf[pw][h], x4[qx], acc[h][pw][qx], ws_v[h], xs_b);
            else mfma_mx4_inplace<qx, pw>(x4[qx], wf[pw][h], acc[h][qx][pw], xs_b, ws_v[h]);
          });
        });
      });
    } else {
      static_for<0, 2>([&](auto K) {
        static_for<0, 4>([&](auto PW) {
          static_for<Q0, Q0 + 2>([&](auto QX) { mfma(std::integral_constant<int, H>{}, PW, QX, decltype(K)::value); });
        });
      });
    }
    if (prio_mfma) __builtin_amdgcn_s_setprio(0);
    barrier();
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  typedef std::integral_constant<int, 2> I2;
  typedef std::integral_constant<int, 3> I3;
  // The eight phases of a tile pair (tile t in buffer 0, t + 1 in buffer 1) of the first (LL = 0) or the second walk.  n_left =
  // tiles of the part from t on: what a phase stages exists only while the part goes on - (t,0) (t,1): tile t+1's second half,
  // (t,2) (t,3): tile t+2's first half (+ its scales, second walk), and so on; a part's last pair may hold one tile only.
  // Second-walk scales (kPrecFp16Mx2): the scales of tile u go to scale buffer u & 1 and are read in phases (u,0) and (u,2) - by
  // wave group 1 one barrier interval after group 0.  They are therefore staged in phase (u-1,0): every wave of BOTH groups is
  // past its last read of tile u-2's scales (LOAD(u-2,2), at least two barrier intervals earlier), and the wait at the end of
  // LOAD(u-1,3) covers them.  (Round 4 staged them in phase (u-2,2), into the buffer the late group was still reading in ITS
  // phase (u-2,2): the race tools/repeat_mx_case.py showed on every launch with time offsets.)  `first`: the part's first
  // pair, whose tiles' scales open_part staged.
  auto pair = [&](auto LL, const int n_left, const int blk, const bool first) __attribute__((always_inline)) {
    constexpr int LOW = decltype(LL)::value;
    const bool t1 = n_left > 1, t2 = n_left > 2, t3 = n_left > 3;
#ifdef XVEC_FUZZ_INJECT_HAZARD
    // Self-test of the fuzzing method (`make fuzz-inject`, never the product): round 4's race put back - the scales of tile u are
    // staged in phase (u-2, 2), into the buffer the late wave group still reads in its own phase (u-2, 2).  tools/fuzz_schedule.py
    // must report the 1.5-pass cases of this build as differing.
    (void)first;
    phase(I0{}, I0{}, I0{}, LL, I0{}, t1 ? 2 : -1, 1, -1, -1, t3);
    phase(I1{}, I0{}, I0{}, LL, I0{}, t1 ? 3 : -1, 1, (MX && !LOW) ? blk : -1, -1, t3);
    if (t2) adv();
    phase(I2{}, I0{}, I0{}, LL, I0{}, t2 ? 0 : -1, 0, -1, (MX2 && t2 && it >= S) ? 0 : -1, t3);
    phase(I3{}, I0{}, I0{}, LL, I0{}, t2 ? 1 : -1, 0, -1, -1, t3);
    if (t1) {
      phase(I0{}, I1{}, I1{}, LL, I0{}, t2 ? 2 : -1, 0, -1, -1, t3);
      phase(I1{}, I1{}, I1{}, LL, I0{}, t2 ? 3 : -1, 0, -1, -1, t3);
      if (t3) adv();
      phase(I2{}, I1{}, I1{}, LL, I0{}, t3 ? 0 : -1, 1, -1, (MX2 && t3 && it >= S) ? 1 : -1, t3);
      phase(I3{}, I1{}, I1{}, LL, I0{}, t3 ? 1 : -1, 1, -1, -1, t3);
    }
#else
    phase(I0{}, I0{}, I0{}, LL, I0{}, t1 ? 2 : -1, 1, -1, (MX2 && t1 && !first && it >= S) ? 1 : -1, t3);
    phase(I1{}, I0{}, I0{}, LL, I0{}, t1 ? 3 : -1, 1, (MX && !LOW) ? blk : -1, -1, t3);
    if (t2) adv();
    phase(I2{}, I0{}, I0{}, LL, I0{}, t2 ? 0 : -1, 0, -1, -1, t3);
    phase(I3{}, I0{}, I0{}, LL, I0{}, t2 ? 1 : -1, 0, -1, -1, t3);
    if (t1) {
      phase(I0{}, I1{}, I1{}, LL, I0{}, t2 ? 2 : -1, 0, -1, (MX2 && t2 && it >= S) ? 0 : -1, t3);
      phase(I1{}, I1{}, I1{}, LL, I0{}, t2 ? 3 : -1, 0, -1, -1, t3);
      if (t3) adv();
      phase(I2{}, I1{}, I1{}, LL, I0{}, t3 ? 0 : -1, 1, -1, -1, t3);
      phase(I3{}, I1{}, I1{}, LL, I0{}, t3 ? 1 : -1, 1, -1, -1, t3);
    }
#endif
  };

  // The same eight phases in the steady state of the first walk (at least two more tiles of the part behind this pair): every
  // phase stages its unit, the block's 4-bit tile goes out in phase (t, 1), the waits are constants.  Leaves the issue counters in
  // the state the generic pair behind it expects (two instructions in each of the last four phases).
  auto pair_fast = [&](const int blk) __attribute__((always_inline)) {
    phase(I0{}, I0{}, I0{}, I0{}, I1{}, 0, 0, blk, -1, true);
    phase(I1{}, I0{}, I0{}, I0{}, I1{}, 0, 0, blk, -1, true);
    adv();
    phase(I2{}, I0{}, I0{}, I0{}, I1{}, 0, 0, blk, -1, true);
    phase(I3{}, I0{}, I0{}, I0{}, I1{}, 0, 0, blk, -1, true);
    phase(I0{}, I1{}, I1{}, I0{}, I1{}, 0, 0, blk, -1, true);
    phase(I1{}, I1{}, I1{}, I0{}, I1{}, 0, 0, blk, -1, true);
    adv();
    phase(I2{}, I1{}, I1{}, I0{}, I1{}, 0, 0, blk, -1, true);
    phase(I3{}, I1{}, I1{}, I0{}, I1{}, 0, 0, blk, -1, true);
    c0 = c1 = c2 = c3 = 2;
  };

  // position of K tile k in the walk (first-walk groups, then the second walk's)
  auto seek = [&](int k, int& g, int& c, int& j) __attribute__((always_inline)) {
    g = 0;
    for (;;) {
      const int n = (a.grp[g].ksteps >> 1) * a.grp[g].nshift;
      if (k < n || g + 1 >= NG) break;
      k -= n;
      ++g;
    }
    const int ns = a.grp[g].nshift;
    c = k / ns;
    j = k - c * ns;
  };
  constexpr long kPartialFloats = 256L * 256;
  constexpr int kAuxCoherent = 1 | 16;
  auto open_part = [&](int part) __attribute__((always_inline)) {
    int tile, kb, ke;
    if (k_tail && part == 0) {
      tile = t_end; kb = 0; ke = k_tail; kind = 1;
    } else {
      const int w = part - (k_tail ? 1 : 0);
      if (w < t_end - t_first) {
        tile = t_first + w; kb = 0; ke = ST; kind = 0;
      } else {
        tile = t_first - 1; kb = k_head; ke = ST; kind = 2;
      }
    }
    const int mt = tb0 + tile / cpl;
    nt = col_lane * cpl + tile % cpl;
    m0 = mt * 256;
    n0 = nt * 256;
    wtile = (const char*)a.w_hi + (long)n0 * a.ldw * 2;
    it = kb;
    seek(kb, ig, ic, ij);
    gi = a.grp[ig];
    bind_group();
    rg = ig;
    r_left = (gi.ksteps >> 1) * gi.nshift - (ic * gi.nshift + ij);
    kb_part = kb;
    n_tiles_part = ke - kb;
    if (EPI != kEpiSplitK && kind != 1 && wave < 6) {
      // bias / scale / offset of the tile's 256 columns -> LDS, [column half][bias | scale | offset][128]
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const unsigned dst = lds_base + kP8PB + (part & 1) * 3072 + hh * 1536 + wave * 256;
        if (wave < 2 || a.bn) {
          const float* src = (wave < 2 ? a.bias : wave < 4 ? a.scale : a.offset) + n0 + hh * 128 + (wave & 1) * 64;
          glds4_sbase(src, (unsigned)lane * 4u, dst);
        } else {
          *(float*)(smem + kP8PB + (part & 1) * 3072 + hh * 1536 + wave * 256 + lane * 4) = wave < 4 ? 1.f : 0.f;
        }
      }
    }
    xs_uvec xg;
    const bool first_walk = kb < S;
    if constexpr (MX) {
      if (first_walk) xs_request(gi.gmax, xg);
    }
    issue(0, 0);
    issue(1, 0);
    issue(2, 0);
    issue(3, 0);
    if (MX2 && it >= S) issue_scales(0);
    if (n_tiles_part > 1) {
      adv();
      issue(0, 1);
      issue(1, 1);
      if (MX2 && it >= S) issue_scales(1);
    }
    if (MX && first_walk) xs_finish(xg);
  };

  open_part(0);
#ifdef XVEC_CLOCK_PROBE
  unsigned long long pp_sum = 0, pp_n = 0;
#endif
#pragma nounroll
  for (int part = 0; part < n_parts; ++part) {
#ifdef XVEC_CLOCK_PROBE
    unsigned long long xp_t0 = __builtin_readcyclecounter(), xp_t1 = xp_t0;
#endif
    XV_FUZZ();
    if (kind == 2) {
      const int prev = bid - 8 * L;   // same lane, previous group: its first action was the store waited for here
      if (tid == 0) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__hip_atomic_load(a.sk_flags + prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.sk_epoch) {
          __builtin_amdgcn_s_sleep(8);
          if (__builtin_readcyclecounter() - t0 > 4000000000ull) {
            __hip_atomic_store(a.sk_error, a.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      __syncthreads();
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.sk_ws + (long)prev * kPartialFloats), 0, (int)(kPartialFloats * 4), 0x00020000);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[h][p][q] = __builtin_bit_cast(
                f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, ((h * 4 + p) * 4 + q) * 8192, kAuxCoherent));
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[h][p][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#ifdef XVEC_CLOCK_PROBE
    xp_t1 = __builtin_readcyclecounter();
#endif
    // everything open_part issued has landed (and the previous epilogue's stores are out: the counter does not tell them apart)
    XV_FUZZ();
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#ifdef XVEC_CLOCK_PROBE
    if (tid == 0 && bid < 512 && kind == 2) {
      const int slot = EPI == kEpiStats ? 2 : (S > 8 ? 1 : 0);
      g_exchange_probe[(slot * 512 + bid) * 4 + 1] = xp_t1 - xp_t0;
      g_exchange_probe[(slot * 512 + bid) * 4 + 2] = __builtin_readcyclecounter() - xp_t1;
    }
    if (kind == 0 && part > 0) {   // (part 0 has no epilogue in front of it)
      pp_sum += __builtin_readcyclecounter() - xp_t1;
      ++pp_n;
    }
#endif
    c0 = c1 = c2 = c3 = 0;
    if (wn == 1) barrier();
    // the pairs of the first walk, then - kPrecFp16Mx2 - those of the second: two loops one behind the other.  (As the two arms
    // of an if inside ONE loop hipcc gives the accumulator tuples different homes in the two bodies and keeps 28 of the 32
    // fragments in scratch between them: 660 bytes per lane.)
    const int ntp = n_tiles_part;
    const int n_hi = MX2 ? (kb_part >= S ? 0 : (kb_part + ntp <= S ? ntp : S - kb_part)) : ntp;   // tiles of the first walk in this part
    // entering another source of the walk: its group maxima (a pair never straddles two groups)
    auto enter_pair = [&]() __attribute__((always_inline)) {
      if constexpr (MX) {
        if (r_left == 0) {
          ++rg;
          r_left = (a.grp[rg].ksteps >> 1) * a.grp[rg].nshift;
          xs_uvec xg;
          xs_request(a.grp[rg].gmax, xg);
          xs_finish(xg);
        }
        r_left -= 2;
      }
    };
    // Two loops one behind the other, never two bodies in one loop (hipcc then splits the accumulator tuples between the bodies
    // and spills ~100 registers - seen again in round 5): first the pairs of the steady state, whose staging and waits are
    // compile-time constants (pair_fast), then the part's last two pairs in the general form.
    int t = 0;
    // (not the kernels at the register limit: the 1.5-pass one - it recomputes its staging offsets as it is - and fp16mxe,
    // which spills one register with the second loop body)
    if constexpr (!MX2 && PREC != kPrecFp16MxE) {
#pragma nounroll
      for (; ntp - t > 3; t += 2) {
        enter_pair();
        pair_fast((kb_part + t) >> 1);
      }
    }
#pragma nounroll
    for (; t < n_hi; t += 2) {
      enter_pair();
      pair(I0{}, ntp - t, (kb_part + t) >> 1, t == 0);
    }
    if constexpr (MX2) {
#pragma nounroll
      for (t = n_hi; t < ntp; t += 2) pair(I1{}, ntp - t, 0, t == 0);
    }
    if (wn == 0) barrier();
    XV_FUZZ();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // last MFMA results -> the epilogue's VALU reads
    // every wave is past its last LDS read: the buffers may be refilled for the next part while this one's results go out
    const int e_kind = kind, e_m0 = m0, e_n0 = n0;
    const float* e_par = (const float*)(smem + kP8PB + (part & 1) * 3072 + wn * 1536);
    if (part + 1 < n_parts) open_part(part + 1);
#ifdef XVEC_CLOCK_PROBE
    const unsigned long long xp_e0 = __builtin_readcyclecounter();
#endif
    if (e_kind == 1) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.sk_ws + (long)bid * kPartialFloats), 0, (int)(kPartialFloats * 4), 0x00020000);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[h][p][q]), rs, tid * 16,
                                                   ((h * 4 + p) * 4 + q) * 8192, kAuxCoherent);
      XV_FUZZ();
      __builtin_amdgcn_s_waitcnt(0x0f70);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(a.sk_flags + bid, a.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef XVEC_CLOCK_PROBE
      if (tid == 0 && bid < 512) g_exchange_probe[((EPI == kEpiStats ? 2 : (S > 8 ? 1 : 0)) * 512 + bid) * 4 + 0] = __builtin_readcyclecounter() - xp_e0;
#endif
    } else {
      float gm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        constexpr bool LAZY = (EPI == kEpiAct || EPI == kEpiF32);   // parameters read from LDS where they are used: 48 registers less
        EpiRegs er;
        epilogue_prefetch_lds<EPI, LAZY>(a, e_par, e_m0 + wm * 64, h * 64, lane, er);
        gemm_epilogue<PREC, EPI, LAZY>(a, acc[h], e_m0 + wm * 64, e_n0 + wn * 128 + h * 64, lane, er, gm, h == 0 ? 1 : 2, e_par, h * 64);
      }
#ifdef XVEC_CLOCK_PROBE
      if (tid == 0 && bid < 512 && e_kind == 0) g_exchange_probe[((EPI == kEpiStats ? 2 : (S > 8 ? 1 : 0)) * 512 + bid) * 4 + 3] = __builtin_readcyclecounter() - xp_e0;
#endif
    }
  }
#ifdef XVEC_CLOCK_PROBE
  if (tid == 0 && bid < 512) {
    const int slot = EPI == kEpiStats ? 2 : (S > 8 ? 1 : 0);
    g_part_probe[(slot * 512 + bid) * 2] = pp_sum;
    g_part_probe[(slot * 512 + bid) * 2 + 1] = pp_n;
    g_clock_probe[(slot * 512 + bid) * 2] = __builtin_readcyclecounter() - clk_t0;
    g_clock_probe[(slot * 512 + bid) * 2 + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
  }
#endif
}

// Workspace of the stream-K variant (one slot of raw accumulators + one flag per workgroup), per stream: launches on
// different streams may overlap.  The epoch makes flags of earlier launches stale without clearing them.  Entries
// belong to whoever owns the stream and are released through release_stream_workspace before the stream is destroyed.
struct SkWorkspace {
  float* ws = nullptr;
  unsigned* flags = nullptr;
  unsigned* error = nullptr;   // pinned host word, device-visible: receives the epoch of a launch on THIS stream whose flag wait timed out
  size_t ws_bytes = 0;
  int grid = 0;
};
static std::mutex g_sk_mu;
static std::map<std::pair<int, hipStream_t>, SkWorkspace> g_sk_table;
static unsigned g_sk_epoch = 0;

static void sk_free(SkWorkspace* w) {
  if (w->ws) (void)hipFree(w->ws);
  if (w->flags) (void)hipFree(w->flags);
  if (w->error) (void)hipHostFree(w->error);
  *w = SkWorkspace();
}

static hipError_t sk_workspace(hipStream_t s, int grid, size_t ws_bytes, SkWorkspace* out, unsigned* epoch, unsigned** err) {
  std::lock_guard<std::mutex> lock(g_sk_mu);
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  SkWorkspace& w = g_sk_table[std::make_pair(dev, s)];
  if (w.ws_bytes < ws_bytes || w.grid < grid) {
    // (re)allocation: only ever on the first launches of a stream; hipFree synchronises the device
    unsigned* keep_error = w.error;   // the error word of a stream survives a regrown workspace
    w.error = nullptr;
    sk_free(&w);
    w.error = keep_error;
    if (!w.error) {
      if ((e = hipHostMalloc((void**)&w.error, 64, hipHostMallocMapped | hipHostMallocPortable)) != hipSuccess) return e;
      *w.error = 0;
    }
    // fine-grained = coherent across the XCDs' L2s without cache maintenance (the exchange happens inside a kernel)
    if ((e = hipExtMallocWithFlags((void**)&w.ws, ws_bytes, hipDeviceMallocFinegrained)) != hipSuccess) {
      sk_free(&w);
      return e;
    }
    if ((e = hipExtMallocWithFlags((void**)&w.flags, (size_t)grid * sizeof(unsigned), hipDeviceMallocFinegrained)) != hipSuccess ||
        // the fill is ordered on the stream whose launches use the flags (this table is keyed by it) - no device-wide wait
        // under g_sk_mu.  A plain hipMemset would run on the null stream, which the non-blocking launch stream
        // is not ordered behind: the first launch could run beside the clearing of its own flags (round 4)
        (e = hipMemsetAsync(w.flags, 0, (size_t)grid * sizeof(unsigned), s)) != hipSuccess) {
      sk_free(&w);
      return e;
    }
    w.ws_bytes = ws_bytes;
    w.grid = grid;
  }
  if (++g_sk_epoch == 0) ++g_sk_epoch;   // 0 is the cleared state
  *epoch = g_sk_epoch;
  *err = w.error;
  *out = w;
  return hipSuccess;
}

void release_stream_workspace(hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_sk_mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  auto it = g_sk_table.find(std::make_pair(dev, s));
  if (it == g_sk_table.end()) return;
  sk_free(&it->second);
  g_sk_table.erase(it);
}

unsigned sk_take_error(hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_sk_mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0u;
  auto it = g_sk_table.find(std::make_pair(dev, s));
  if (it == g_sk_table.end() || !it->second.error) return 0u;
  volatile unsigned* w = it->second.error;
  const unsigned v = *w;
  if (v) *w = 0u;   // reported once: later launches on the stream start clean
  return v;
}

// True when the stream-K variant with MF fragments per wave can run this launch.
template <int PREC, int MF>
static bool sk_applicable(const GemmArgs& a) {
  constexpr int lds = 3 * (PrecXPlanes(PREC) * (64 * MF + 16) * kBK * 2 + PrecWPlanes(PREC) * kTileBytes) + 1536 + 2 * 1536 + (PrecMx2(PREC) ? 3 * (64 * MF + 16) * 4 : 0);
  if (lds > 160 * 1024) return false;
  const int rows = a.m_tiles * kBM;
  if (rows % (64 * MF)) return false;
  const int grid = device_cu_count() / 8 * 8;
  if (grid < 8) return false;
  // every workgroup gets at least one tile's worth of K steps (XCD blocks are whole row tiles) ...
  if ((long)(rows / (64 * MF) / 8) * a.n_tiles >= grid / 8) return true;
  // ... or, on fewer workgroups (sk_groups), launches between the two regimes: at least four row tiles per XCD block, and
  // more 256-row tiles than the per-tile kernel finishes in one round (48 and 64 chunks of 400 frames: 300 / 400 tiles of
  // tdnn2 on 256 CUs = two rounds, 82 us for both, where one round is 42)
  return MF == 8 && rows / (64 * MF) / 8 >= 4 && (long)(a.m_tiles / 2) * a.n_tiles > device_cu_count();
}

// Workgroup groups per XCD block and column lane: as many as the CUs allow, but not more than the block has tiles per lane
template <int MF>
static int sk_groups(const GemmArgs& a, int grid, int lanes) {
  const int tiles_min = std::max(1, (a.m_tiles * kBM / (64 * MF) / 8) * (a.n_tiles / lanes));
  return std::min(grid / 8 / lanes, tiles_min);
}

template <int PREC, int EPI, int MF>
static hipError_t launch_one_sk(const GemmArgs& a, hipStream_t s) {
  constexpr int lds = 3 * (PrecXPlanes(PREC) * (64 * MF + 16) * kBK * 2 + PrecWPlanes(PREC) * kTileBytes) + 1536 + 2 * 1536 + (PrecMx2(PREC) ? 3 * (64 * MF + 16) * 4 : 0);
  if constexpr (lds > 160 * 1024) {
    return hipErrorInvalidValue;
  } else {
    static std::atomic<unsigned long long> attr_done{0};
    int attr_dev = 0;
    if (lds_attr_needed(&attr_done, &attr_dev)) {
      hipError_t e = hipFuncSetAttribute((const void*)tdnn_gemm_kernel_sk<PREC, EPI, MF>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return e;
      attr_done.fetch_or(1ull << (attr_dev & 63), std::memory_order_release);
    }
    int grid = device_cu_count() / 8 * 8;
    GemmArgs b = a;
    build_groups(&b);
    if constexpr (PrecMx2(PREC)) build_lo_groups(&b);
    b.sk_mtiles = a.m_tiles * kBM / (64 * MF);
    {
      // column lanes: the largest of 4, 2, 1 that divides the column tiles and the workgroups of an XCD block
      static const int max_lanes = [] {
        const int x = DebugKnobInt("sk_lanes", 4);
        return (x != 1 && x != 2 && x != 4) ? 4 : x;
      }();
      int l = max_lanes;
      while (l > 1 && (a.n_tiles % l || (grid / 8) % l)) l >>= 1;
      b.sk_lanes = l;
      grid = 8 * l * sk_groups<MF>(a, grid, l);
    }
    SkWorkspace w;
    hipError_t e = sk_workspace(s, grid, (size_t)grid * 64 * MF * kBN * sizeof(float), &w, &b.sk_epoch, &b.sk_error);
    if (e != hipSuccess) return e;
    b.sk_ws = w.ws;
    b.sk_flags = w.flags;
    note_kernel("_sk", PREC, EPI, MF);
    XV_LAUNCH((tdnn_gemm_kernel_sk<PREC, EPI, MF>), dim3(grid), dim3(512), lds, s, b);
    return hipGetLastError();
  }
}


// tdnn_gemm_kernel_p8 can run this launch: whole 256 x 256 tiles, every K group a whole number of 64-column tiles (of
// 128-column blocks for kPrecFp16Mx), an even number of K tiles, and for kPrecFp16Mx the residual plane in ITS walk order
// (GemmArgs::p8 is the caller's statement that w4 / w4_scale are in that order)
bool gemm_p8_applicable(const GemmArgs& a, int precision) {
  if (precision == kPrecFp16MxE) precision = kPrecFp16Mx;   // the same product; its planes epilogue also writes the residual plane
  if (precision != kPrecFp16 && precision != kPrecFp16Mx && precision != kPrecFp16Mx2) return false;
  if ((a.m_tiles & 1) || (a.n_tiles & 1) || a.ksplit > 1) return false;
  const bool mx = precision != kPrecFp16, mx2 = precision == kPrecFp16Mx2;
  GemmArgs b = a;
  build_groups(&b);
  int t = 0;
  for (int i = 0; i < b.ngrp; ++i) {
    // whole K tiles: 64 columns; whole blocks of the residual product: 128; whole tiles of the second walk: 256
    if (b.grp[i].ksteps % (mx2 ? 8 : mx ? 4 : 2) || b.grp[i].ld % (mx2 ? 256 : 64)) return false;
    if (mx && !b.grp[i].gmax) return false;
    t += (b.grp[i].ksteps >> 1) * b.grp[i].nshift;
  }
  if (t < 2 || (t & 1)) return false;
  if (mx && (!a.w4 || !a.w4_scale || a.ldw4 <= 0)) return false;
  if (mx2) {
    if (!a.w4b || !a.w4b_scale || a.ldw4b <= 0) return false;
    for (int j = 0; j < a.nseg; ++j)
      if (!a.seg[j].lo4 || !a.seg[j].lo4s) return false;
  }
  return true;
}

template <int PREC, int EPI>
static hipError_t launch_one_p8(const GemmArgs& a, hipStream_t s) {
  if (!gemm_p8_applicable(a, PREC)) return hipErrorInvalidValue;
  static std::atomic<unsigned long long> attr_done{0};
  int attr_dev = 0;
  if (lds_attr_needed(&attr_done, &attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)tdnn_gemm_kernel_p8<PREC, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, kP8Lds);
    if (e != hipSuccess) return e;
    attr_done.fetch_or(1ull << (attr_dev & 63), std::memory_order_release);
  }
  GemmArgs b = a;
  build_groups(&b);
  b.p8_ktiles = 0;
  for (int i = 0; i < b.ngrp; ++i) b.p8_ktiles += (b.grp[i].ksteps >> 1) * b.grp[i].nshift;
  b.p8_ktiles_lo = 0;
  if constexpr (PrecMx2(PREC)) {
    build_lo_groups(&b);   // second walk: the 4-bit planes, steps of 128 columns (a tile = two of them)
    for (int i = 0; i < b.ngrp_lo; ++i) b.p8_ktiles_lo += (b.grp[b.ngrp + i].ksteps >> 1) * b.grp[b.ngrp + i].nshift;
  }
  b.p8 = 1;
  // GemmArgs::p8_whole as the caller's policy (Engine: XVEC_DEBUG=p8_whole): 0 = K tiles dealt out evenly over the workgroups (stream-K
  // exchange; the default), 1 = whole tiles for every launch, 2 = whole tiles for layers of at most 8 K tiles per output tile
  b.p8_whole = (a.p8_whole == 1 || (a.p8_whole == 2 && b.p8_ktiles + b.p8_ktiles_lo <= 8)) ? 1 : 0;
  b.sk_mtiles = a.m_tiles >> 1;
  const int nt = a.n_tiles >> 1;
  int grid = device_cu_count() / 8 * 8;
  if (grid < 8) return hipErrorInvalidValue;
  // column lanes: the largest of 4, 2, 1 that divides the column tiles and the workgroups of an XCD block; groups: as many
  // as the CUs allow, but every group's share must hold at least one whole tile
  int l = 4;
  while (l > 1 && (nt % l || (grid / 8) % l)) l >>= 1;
  // Column tiles in threes (1536 = 6, 768 = 3 tiles of 256: the statistics layer, the 650-wide layers of the c-vector network):
  // with two lanes - or one - every workgroup walks its row tile's frames once per column tile, 16 to 32 row tiles are in flight
  // per XCD (4 - 8 MB of frames against 4 MB of L2), and every pass after the first misses: 578 MB per tdnn5 launch for a 105 MB
  // plane (profiles/r05w_pmc_hbm.md).  As many lanes as column tiles instead: the lanes of a group walk the SAME row tile at the
  // same time, the first brings a line into the XCD's L2 and the others find it there.  The workgroups of an XCD block are
  // then a multiple of 3 (30 of 32 CUs); the partition is bit-identical whatever the cut.
  // (same box, A/B: tdnn5 of the x-vector 572 -> 245 MB per launch at the same 0.205 ms; the 768-wide layers of the c-vector
  // network 1141 -> 330 MB and 721 -> 285 MB, their launches +2.5 % on their own and the two-lane step -3.5 % - profiles/r06_lanes.md)
  if (nt % 3 == 0 && nt <= 6 && grid / 8 >= nt) l = nt;
  b.sk_lanes = l;
  const int tiles_min = std::max(1, (b.sk_mtiles / 8) * (nt / l));
  grid = 8 * l * std::min(grid / 8 / l, tiles_min);
  SkWorkspace w;
  hipError_t e = sk_workspace(s, grid, (size_t)grid * 256 * 256 * sizeof(float), &w, &b.sk_epoch, &b.sk_error);
  if (e != hipSuccess) return e;
  b.sk_ws = w.ws;
  b.sk_flags = w.flags;
  note_kernel("_p8", PREC, EPI, 0);
  XV_LAUNCH((tdnn_gemm_kernel_p8<PREC, EPI>), dim3(grid), dim3(512), kP8Lds, s, b);
  return hipGetLastError();
}

template <int PREC, int EPI>
static hipError_t launch_one_v2(const GemmArgs& a, hipStream_t s) {
  constexpr int lds = 3 * (PrecXPlanes(PREC) * (256 + 16) * kBK * 2 + PrecWPlanes(PREC) * kTileBytes) + (PrecMx(PREC) ? 1536 : 0) + (PrecMx2(PREC) ? 3 * (256 + 16) * 4 : 0);
  static std::atomic<unsigned long long> attr_done{0};
  int attr_dev = 0;
  if (lds_attr_needed(&attr_done, &attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)tdnn_gemm_kernel_v2<PREC, EPI>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_done.fetch_or(1ull << (attr_dev & 63), std::memory_order_release);
  }
  const int mt = a.m_tiles >> 1;
  const int mt8 = (mt + 7) / 8 * 8;
  dim3 grid(mt8 * a.n_tiles), block(512);
  GemmArgs b = a;
  build_groups(&b);
  if constexpr (PrecMx2(PREC)) build_lo_groups(&b);
  {
    // stagger window = gemm_stagger (XVEC_DEBUG) percent of the modelled tile time (K steps x ~2200 cycles in split mode,
    // ~1000 single pass, + ~14000 fixed)
    static const int pct = DebugKnobInt("gemm_stagger", 85);
    const int cus = device_cu_count();
    b.stagger_wgs = cus;
    b.stagger_units = ((int)grid.x > 2 * cus) ? (int)(((long)b.total_ksteps * (400 + 600 * PrecPasses(PREC)) + 14000) * pct / 100 / 2048) : 0;
  }
  note_kernel("_v2", PREC, EPI, 0);
  XV_LAUNCH((tdnn_gemm_kernel_v2<PREC, EPI>), grid, block, lds, s, b);
  return hipGetLastError();
}

template <int PREC, int EPI>
static hipError_t launch_one(const GemmArgs& a, hipStream_t s) {
  if (a.p8) {   // the caller packed / chose this layer for the 64-column K walk: no other kernel accumulates in that order
    if constexpr (((PREC == kPrecFp16 || PREC == kPrecFp16Mx || PREC == kPrecFp16Mx2) && (EPI == kEpiAct || EPI == kEpiStats)) ||
                  (PREC == kPrecFp16MxE && EPI == kEpiAct))
      return launch_one_p8<PREC, EPI>(a, s);
    else return hipErrorInvalidValue;
  }
  const int variant = gemm_variant();
  // default policy (same box, 256 x 400 workload): two-pass mode - stream-K for every layer; three-pass and single-pass
  // modes - stream-K only for the long-K layers (tdnn2 / tdnn3, 48 K steps: -5 % / -9 %), the short-K ones are faster
  // on the per-tile kernel there (tdnn5 single-pass: 0.190 vs 0.229 ms)
  const bool sk_default = PREC == kPrecFp16x2 || PREC == kPrecFp16Mx || a.total_ksteps >= 32;
  if constexpr (PrecMx(PREC)) {
    // 512-row stream-K tiles or the 256-row per-tile kernel (bit-identical); nothing else implements the mode
    if (PrecMx2(PREC) ? !gemm_mx2_applicable(a) : !gemm_mx_applicable(a)) return hipErrorInvalidValue;
    if (variant != 2 && variant != 1 && sk_max_mf() == 8 && sk_applicable<PREC, 8>(a)) return launch_one_sk<PREC, EPI, 8>(a, s);
    return launch_one_v2<PREC, EPI>(a, s);
  } else {
    if (variant == 4 || (variant == 0 && sk_default)) {
      if (sk_max_mf() == 8 && sk_applicable<PREC, 8>(a)) return launch_one_sk<PREC, EPI, 8>(a, s);
      if ((variant == 4 || PrecXPlanes(PREC) == 2) && sk_applicable<PREC, 4>(a)) return launch_one_sk<PREC, EPI, 4>(a, s);
    }
    if (variant != 1 && (a.m_tiles & 1) == 0) return launch_one_v2<PREC, EPI>(a, s);
    constexpr int lds = kTileBytes * (PrecXPlanes(PREC) + PrecWPlanes(PREC)) * 2;
    static std::atomic<unsigned long long> attr_done{0};
    int attr_dev = 0;
    if (lds_attr_needed(&attr_done, &attr_dev)) {
      hipError_t e = hipFuncSetAttribute((const void*)tdnn_gemm_kernel<PREC, EPI>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return e;
      attr_done.fetch_or(1ull << (attr_dev & 63), std::memory_order_release);
    }
    const int mt8 = (a.m_tiles + 7) / 8 * 8;
    dim3 grid(mt8 * a.n_tiles), block(256);
    note_kernel("", PREC, EPI, 0);
    XV_LAUNCH((tdnn_gemm_kernel<PREC, EPI>), grid, block, lds, s, a);
    return hipGetLastError();
  }
}

// ---------------------------------------------------------------------------------------------
// Split-K reduction: out = epilogue(sum over K slices, in slice order, + bias).  One thread = 4 columns of a row.
template <int PREC, int EPI>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs a) {
  constexpr bool SPLIT = PrecXPlanes(PREC) == 2;   // activations carry a residual plane
  constexpr bool F16 = PrecF16(PREC);
  const int n_pad = a.n_tiles * kBN;
  const int rows = a.m_tiles * kBM;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_row = n_pad >> 2;
  if (idx >= (long)rows * per_row) return;
  const int row = (int)(idx / per_row);
  const int col = (int)(idx - (long)row * per_row) * 4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  const long slab = (long)rows * n_pad;
  // the slices are added in slice order (one fixed order: same bits for any number of resident workgroups), but fetched
  // eight at a time: one load per pass of a loop of unknown length is a chain of memory latencies (0.6-0.9 us per slice)
  const float* src = a.splitk_ws + (long)row * n_pad + col;
  int k = 0;
  for (; k + 8 <= a.ksplit; k += 8) {
    f32x4 t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = *(const f32x4*)(src + (k + j) * slab);
#pragma unroll
    for (int j = 0; j < 8; ++j) v += t[j];
  }
  {
    f32x4 t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = (k + j < a.ksplit) ? *(const f32x4*)(src + (k + j) * slab) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (k + j < a.ksplit) v += t[j];
  }
  const f32x4 b = *(const f32x4*)(a.bias + col);
  float y[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float z = v[r] + b[r];
    if (a.relu) z = (z < 0.f) ? 0.f : z;  // Kaldi ApplyFloor(0): a NaN stays a NaN (fmaxf would turn it into 0)
    if (a.bn) z = __builtin_fmaf(z, a.scale[col + r], a.offset[col + r]);   // one rounding, like the GEMM epilogues
    y[r] = z;
  }
  if constexpr (EPI == kEpiF32) {
    if (row < a.m_valid) *(f32x4*)(a.out_f32 + (long)row * a.ldf + col) = f32x4{y[0], y[1], y[2], y[3]};
  } else {
    const uint16_t h0 = to16<F16>(y[0]), h1 = to16<F16>(y[1]), h2 = to16<F16>(y[2]), h3 = to16<F16>(y[3]);
    unsigned int* dh = (unsigned int*)(a.out_hi + (long)row * a.ldo + col);
    dh[0] = (unsigned int)h0 | ((unsigned int)h1 << 16);
    dh[1] = (unsigned int)h2 | ((unsigned int)h3 << 16);
    if constexpr (SPLIT) {
      const uint16_t l0 = to16<F16>(y[0] - from16<F16>(h0)), l1 = to16<F16>(y[1] - from16<F16>(h1));
      const uint16_t l2 = to16<F16>(y[2] - from16<F16>(h2)), l3 = to16<F16>(y[3] - from16<F16>(h3));
      unsigned int* dl = (unsigned int*)(a.out_lo + (long)row * a.ldo + col);
      dl[0] = (unsigned int)l0 | ((unsigned int)l1 << 16);
      dl[1] = (unsigned int)l2 | ((unsigned int)l3 << 16);
    }
  }
}

template <int PREC, int EPI>
static hipError_t launch_splitk(const GemmArgs& a, hipStream_t s) {
  constexpr int lds = kTileBytes * (PrecXPlanes(PREC) + PrecWPlanes(PREC)) * kSplitKStages;
  static std::atomic<unsigned long long> attr_done{0};
  int attr_dev = 0;
  if (lds_attr_needed(&attr_done, &attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)tdnn_gemm_kernel<PREC, kEpiSplitK>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_done.fetch_or(1ull << (attr_dev & 63), std::memory_order_release);
  }
  dim3 grid(a.m_tiles * a.n_tiles, a.ksplit), block(256);
  note_kernel("", PREC, kEpiSplitK, 0);
  XV_LAUNCH((tdnn_gemm_kernel<PREC, kEpiSplitK>), grid, block, lds, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const long total = (long)a.m_tiles * kBM * (a.n_tiles * kBN / 4);
  XV_LAUNCH((splitk_reduce_kernel<PREC, EPI>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

template <int PREC>
static hipError_t launch_prec(const GemmArgs& a, int epi, hipStream_t s) {
  if constexpr (PREC == kPrecFp16x3E) {   // planes out only (the layers in front of a kPrecFp16Mx2 consumer)
    if (a.ksplit > 1 || epi != kEpiAct || !a.out_lo4 || !a.out_lo4s) return hipErrorInvalidValue;
    return launch_one<PREC, kEpiAct>(a, s);
  } else if constexpr (PREC == kPrecFp16MxE) {   // planes out only: the 1.25-pass product in front of a kPrecFp16Mx2 consumer
    if (a.ksplit > 1 || epi != kEpiAct || !a.out_lo4 || !a.out_lo4s) return hipErrorInvalidValue;
    return launch_one<PREC, kEpiAct>(a, s);
  } else if constexpr (PrecMx(PREC)) {   // frame-level layers only: planes out or pooled statistics
    if (a.ksplit > 1) return hipErrorInvalidValue;
    if (epi == kEpiAct) return launch_one<PREC, kEpiAct>(a, s);
    if (epi == kEpiStats) return launch_one<PREC, kEpiStats>(a, s);
    return hipErrorInvalidValue;
  } else {
    if (a.ksplit > 1 && a.splitk_ws) {
      if (epi == kEpiAct) return launch_splitk<PREC, kEpiAct>(a, s);
      if (epi == kEpiF32) return launch_splitk<PREC, kEpiF32>(a, s);
      return hipErrorInvalidValue;
    }
    switch (epi) {
      case kEpiAct: return launch_one<PREC, kEpiAct>(a, s);
      case kEpiF32: return launch_one<PREC, kEpiF32>(a, s);
      case kEpiStats: return launch_one<PREC, kEpiStats>(a, s);
      default: return hipErrorInvalidValue;
    }
  }
}

hipError_t launch_tdnn_gemm(const GemmArgs& a, int precision, int epilogue, hipStream_t s) {
  if (a.nseg < 1 || a.nseg > kMaxSeg || a.m_tiles < 1 || a.n_tiles < 1) return hipErrorInvalidValue;
  switch (precision) {
    case kPrecBf16x3: return launch_prec<kPrecBf16x3>(a, epilogue, s);
    case kPrecBf16: return launch_prec<kPrecBf16>(a, epilogue, s);
    case kPrecFp16: return launch_prec<kPrecFp16>(a, epilogue, s);
    case kPrecFp16x3: return launch_prec<kPrecFp16x3>(a, epilogue, s);
    case kPrecFp16x2: return launch_prec<kPrecFp16x2>(a, epilogue, s);
    case kPrecFp16Mx: return launch_prec<kPrecFp16Mx>(a, epilogue, s);
    case kPrecFp16Mx2: return launch_prec<kPrecFp16Mx2>(a, epilogue, s);
    case kPrecFp16x3E: return launch_prec<kPrecFp16x3E>(a, epilogue, s);
    case kPrecFp16MxE: return launch_prec<kPrecFp16MxE>(a, epilogue, s);
    default: return hipErrorInvalidValue;
  }
}

// ---------------------------------------------------------------------------------------------
// tdnn_first: the layers that read the network input (see kernels.h, FirstArgs).
//  * Work unit = 64 frames x one group of up to 512 output columns; a persistent grid (one 512-thread workgroup per CU)
//    takes contiguous ranges of units.  Wave w owns the 64 columns w * 64.. of the group.
//  * Weights stay on the CU for the whole launch: the wave's 64 rows of the compact hi plane [n][128] in REGISTERS (4
//    fragments x 4 K steps = 64 VGPRs), the lo plane of the group's 512 rows in LDS (128 KiB, 16-byte chunks XOR-swizzled
//    with four row bits so that a fragment read is conflict free; both planes in registers spill), loaded once (again only
//    when a workgroup's range crosses into the next column group), in the "weights as MFMA A operand" row order of the
//    other kernels (swap_fields), so the generic planes epilogue applies unchanged.
//  * Features: the unit's frames + halo, split into fp16 hi / lo, in LDS as [frame][dp] (dp = dim rounded up to 8: 24
//    halves = 48 bytes per frame).  Column k' = j * dp + d of the compact K axis is element d of frame t + off[j]: a lane's
//    16-byte fragment chunk (8 consecutive k') never straddles two frames, so the splice costs one address per lane and
//    step (xo[]), computed once.  Double buffered: the next unit's rows are fetched into registers before the MFMAs of the
//    current one and written to the other buffer after them; one workgroup barrier per unit.
//  * 192 MFMAs per wave and unit against 32 fragment reads; the kernel is bound by the planes it writes.
template <int EPREC>
__global__ __launch_bounds__(512) void tdnn_first_kernel(const FirstArgs fa) {
  const GemmArgs& a = fa.g;
  constexpr bool F16 = PrecF16(EPREC);
  constexpr int XR = kFirstRows + 32;   // frames per buffer: unit + halo (offset span <= 30)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* par_all = (float*)smem;        // [4 tiles][bias 128 | scale 128 | offset 128]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr_i = lane & 15, fr_g = lane >> 4;
  const int dp = (fa.dim + 7) & ~7;
  int min_off = fa.off[0], max_off = fa.off[0];
  for (int j = 1; j < fa.noff; ++j) {
    min_off = min(min_off, fa.off[j]);
    max_off = max(max_off, fa.off[j]);
  }
  const int span = max_off - min_off;
  const int plane_bytes = XR * dp * 2;
  char* wlo_s = smem + 4 * 384 * 4;     // [512 rows][16 chunks of 16 bytes], chunk c of row r at c ^ key(r)
  char* xbuf = wlo_s + 512 * 256;       // [buffer][hi | lo][XR * dp halves]
  auto wkey = [](int r) __attribute__((always_inline)) { return ((r >> 1) & 12) | (r & 3); };

  const int n_pad = a.n_tiles * kBN;
  // workgroup b works on column group b % ncg (one group for layers of up to 512 columns) and on a contiguous range of its
  // row blocks
  const int nrb = fa.nrows / kFirstRows, ncg = (n_pad + 511) / 512;
  const int cg = blockIdx.x % ncg, wg = blockIdx.x / ncg, nwg = gridDim.x / ncg;
  const int rb0 = (int)((long)nrb * wg / nwg), rb1 = (int)((long)nrb * (wg + 1) / nwg);
  if (rb0 >= rb1 || wg >= nwg) return;
  const int n0w = cg * 512 + wave * 64;
  const bool cols_valid = n0w < n_pad;
  XV_FUZZ_INIT();

  // this lane's fragment chunk of K step s: columns k' = 32 s + 8 g .. + 7  ->  frame offset off[j], element d0 (bytes from
  // the unit's first staged frame, this lane's frame fr_i included).  Beyond noff * dp the weights are zero: any finite
  // data of the same frame will do (the last chunk of the last offset).
  int xo[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int k = 32 * s + 8 * fr_g;
    int j = k / dp, d0 = k - j * dp;
    if (j >= fa.noff) {
      j = fa.noff - 1;
      d0 = dp - 8;
    }
    xo[s] = ((fa.off[j] - min_off + fr_i) * dp + d0) * 2;
  }

  // ---- staging: global -> registers (prefetch) -> LDS (commit) --------------------------------------------------
  // Where the frames of a device row come from: the plan's per-group table (FirstArgs::grp_src), a window of which this
  // workgroup keeps in LDS (looked up per unit from global memory the lookup is a dependent load in front of the feature
  // loads; as the chain grp_utt -> dev_off / src_off it cost more than the unit's MFMAs).
  constexpr int kTabUnits = 96;                     // row blocks per table window
  constexpr int kTabGroups = kTabUnits * 4 + 8;     // + halo groups on both sides
  int4* gtab = (int4*)(xbuf + 4 * plane_bytes);
  int tab_rb0 = -(1 << 30), tab_g0 = 0;             // first row block of the window, its first group
  auto build_table = [&](int rb_first) __attribute__((always_inline)) {
    tab_rb0 = rb_first;
    tab_g0 = ((fa.row0 + rb_first * kFirstRows) >> 4) - 4;
    for (int i = tid; i < kTabGroups; i += 512) {
      const int g = tab_g0 + i;
      gtab[i] = (g >= 0 && g * 16 < fa.rows) ? fa.grp_src[g] : int4{0, 0, 0, 0};
    }
  };
  const int n_slots = (kFirstRows + span) * dp;
  int slot_fd[kFirstMaxSlots];   // frame (inside the staged rows) << 8 | element of this thread's slots; -1: none
#pragma unroll
  for (int i = 0; i < kFirstMaxSlots; ++i) {
    const int slot = tid + 512 * i;
    const int fr = slot / dp;
    slot_fd[i] = slot < n_slots ? (fr << 8) | (slot - fr * dp) : -1;
  }
  // the loads are unconditional (a slot without a source frame reads the batch's first element and is zeroed at commit): a conditional load
  // is merged with its default right where it is issued, i.e. waited for in front of the MFMAs instead of after them
  float pv[kFirstMaxSlots] = {0.f, 0.f, 0.f, 0.f};
  unsigned pv_ok = 0u;
  auto prefetch = [&](int rb) __attribute__((always_inline)) {
    const int r_first = fa.row0 + rb * kFirstRows + min_off;
    pv_ok = 0u;
#pragma unroll
    for (int i = 0; i < kFirstMaxSlots; ++i) {
      long idx = fa.feats_valid_idx;
      if (slot_fd[i] >= 0) {
        const int d = slot_fd[i] & 255;
        const int r = r_first + (slot_fd[i] >> 8);
        const int4 e = gtab[(r >> 4) - tab_g0];
        const int t = r & 15;
        if (d < fa.dim && t < e.y) {
          idx = (long)min(max(e.x + t, e.z), e.w) * fa.dim + d;
          pv_ok |= 1u << i;
        }
      }
      pv[i] = fa.feats[idx];
    }
    __builtin_amdgcn_sched_barrier(0);   // nothing that uses the values moves up here (their wait would come with it)
  };
  auto commit = [&](int buf) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    // the conversions below depend on nothing but pv: without this hipcc hoists them out of the loop over the two halves of
    // the interval, to right behind the loads - and waits for the next unit's features in front of this unit's MFMAs
#pragma unroll
    for (int i = 0; i < kFirstMaxSlots; ++i) asm volatile("" : "+v"(pv[i]));
    uint16_t* hi = (uint16_t*)(xbuf + buf * 2 * plane_bytes);
    uint16_t* lo = (uint16_t*)(xbuf + buf * 2 * plane_bytes + plane_bytes);
#pragma unroll
    for (int i = 0; i < kFirstMaxSlots; ++i) {
      const int slot = tid + 512 * i;
      if (slot_fd[i] >= 0) {
        const float v = (pv_ok >> i & 1u) ? pv[i] : 0.f;
        const uint16_t h = to16<F16>(v);
        hi[slot] = h;
        lo[slot] = to16<F16>(v - from16<F16>(h));
      }
    }
  };

  s16x8 wh[4][4];             // [fragment p of the wave's 64 columns][K step]
  int wl_rd[4];               // byte address of this lane's lo-plane row of fragment p, and its swizzle key
  int wl_key[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = wave * 64 + swap_fields(p * 16 + fr_i);
    wl_rd[p] = r * 256;
    wl_key[p] = wkey(r);
  }
  // Waves 0-3 / 4-7 (one of each per SIMD) run half a unit apart: inside one barrier interval group 0 multiplies unit u and
  // then stores it, group 1 first stores unit u - 1 (its accumulators survive the barrier) and then multiplies unit u - so
  // on every SIMD one wave feeds the matrix pipe while the other converts and stores, and the planes leave the chip all the
  // time instead of in bursts (in lockstep the kernel was the sum of its phases: 17 us of MFMAs + 33 us of stores).  The
  // interval is written as two passes over ONE body {store what is pending; multiply} so that the epilogue is instantiated
  // once (inlined at every place it is needed the kernel spilled ~100 registers).
  const int group = wave >> 2;
  f32x4 acc[4][4];
  int pend_mbase = 0;
  bool pending = false;

  // parameters and lo plane -> LDS, hi plane -> registers: every global load of a batch is issued before its first LDS store
  // (as a loop of load-store pairs this took one memory latency per pass: 16 us of an 80 us launch)
  {
    // thread t: column cg * 512 + t of the group (without BatchNorm: scale 1, offset 0 - the epilogue reads them unconditionally)
    float pr[3] = {0.f, 1.f, 0.f};
    {
      const int c = cg * 512 + tid;
      if (c < n_pad) {
        pr[0] = a.bias[c];
        if (a.bn) {
          pr[1] = a.scale[c];
          pr[2] = a.offset[c];
        }
      }
    }
    if (cols_valid) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const long wrow = (long)(n0w + swap_fields(p * 16 + fr_i)) * kFirstK + fr_g * 8;
#pragma unroll
        for (int st = 0; st < 4; ++st) wh[p][st] = *(const s16x8*)(fa.wc_hi + wrow + st * 32);
      }
    }
    u32x4 wv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = tid + 512 * k;
      const int r = i >> 4, c = i & 15;
      wv[k] = u32x4{0u, 0u, 0u, 0u};
      if (cg * 512 + r < n_pad) wv[k] = *(const u32x4*)(fa.wc_lo + (long)(cg * 512 + r) * kFirstK + c * 8);
    }
    build_table(rb0);   // its loads travel with the weights'
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = tid + 512 * k;
      const int r = i >> 4, c = i & 15;
      *(u32x4*)(wlo_s + r * 256 + ((c ^ wkey(r)) << 4)) = wv[k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) par_all[(tid >> 7) * 384 + k * 128 + (tid & 127)] = pr[k];
  }
  __syncthreads();
  prefetch(rb0);
  commit(0);
#pragma unroll 1
  for (int rb = rb0; rb <= rb1; ++rb) {   // one pass more than there are units: group 1's last store
    const bool has_unit = rb < rb1;
    const int buf = (rb - rb0) & 1;
    // the unit after this one must lie inside the table window (its rows are fetched during this unit)
    if (rb + 1 < rb1 && rb + 1 >= tab_rb0 + kTabUnits) {
      __syncthreads();
      build_table(rb + 1);
    }
    __syncthreads();   // this unit's frames are in LDS; the other buffer is free
    XV_FUZZ();
    if (rb + 1 < rb1) prefetch(rb + 1);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      XV_FUZZ();
      if (pending && (half == 0 || group == 0)) {
        pending = false;
        const float* par = par_all + (wave >> 1) * 384;
        EpiRegs er;
        // the parameters of the block's 16 columns per lane: held in registers (48) where the block's fragments are dead and
        // nothing else competes for them, re-read from LDS per 16-row group where the 4-bit residual is emitted as well
        constexpr bool LZ = PrecEmitsLo4(EPREC);
        epilogue_prefetch_lds<kEpiAct, LZ>(a, par, pend_mbase, (wave & 1) * 64, lane, er);
        float gm[4] = {0.f, 0.f, 0.f, 0.f};
        gemm_epilogue<EPREC, kEpiAct, LZ>(a, acc, pend_mbase, n0w, lane, er, gm, 0, par, (wave & 1) * 64);
      }
      if (half == 0 && has_unit) {
        if (cols_valid) {
          {
            const char* xh_b = xbuf + buf * 2 * plane_bytes;
            const char* xl_b = xh_b + plane_bytes;
            // Fragment reads rotate through the three products of a step (w_lo x_hi, w_hi x_hi, w_hi x_lo): a fragment set is
            // re-read for step st + 1 right behind the last 16 MFMAs that use it in step st, at least 16 MFMAs before its next use.  In this kernel a
            // wave multiplies ALONE on its SIMD (its partner converts and stores meanwhile), so a read waited for in front of a
            // step's MFMAs is matrix-pipe idle time: with the reads of a step issued together at its head the two multiply
            // phases of an interval took 2 x 3.5 us against 1.5 us of MFMA issue each.
            s16x8 xh[4], xl[4], wl[4];
            auto rd_wl = [&](int st) __attribute__((always_inline)) {
#pragma unroll
              for (int p = 0; p < 4; ++p) wl[p] = *(const s16x8*)(wlo_s + wl_rd[p] + (((st * 4 + fr_g) ^ wl_key[p]) << 4));
            };
            auto rd_xh = [&](int st) __attribute__((always_inline)) {
#pragma unroll
              for (int q = 0; q < 4; ++q) xh[q] = *(const s16x8*)(xh_b + q * 16 * dp * 2 + xo[st]);
            };
            auto rd_xl = [&](int st) __attribute__((always_inline)) {
#pragma unroll
              for (int q = 0; q < 4; ++q) xl[q] = *(const s16x8*)(xl_b + q * 16 * dp * 2 + xo[st]);
            };
            rd_wl(0);
            rd_xh(0);
            rd_xl(0);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
              // three products per accumulator, 16 independent MFMAs between two on the same one.  fp16: accumulators
              // tied in place from inline asm (left to itself hipcc rotates them through copies: spills); the first MFMA
              // of a unit is the builtin with a literal-zero addend - a v_mov clearing the accumulator right in front of
              // an asm MFMA is a VALU-write -> MFMA-read hazard hipcc cannot see (it produced garbage)
#pragma unroll
              for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  if (st == 0 || !F16) acc[p][q] = mfma16<F16>(wl[p], xh[q], st == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[p][q]);
                  else mfma16_f16_inplace(wl[p], xh[q], acc[p][q]);
                }
              __builtin_amdgcn_sched_barrier(0);
              if (st < 3) rd_wl(st + 1);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  if constexpr (F16) mfma16_f16_inplace(wh[p][st], xh[q], acc[p][q]);
                  else acc[p][q] = mfma16<F16>(wh[p][st], xh[q], acc[p][q]);
                }
              __builtin_amdgcn_sched_barrier(0);
              if (st < 3) rd_xh(st + 1);   // needed by the next step's first product: 16 MFMAs (>= 256 cycles) ahead of it
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  if constexpr (F16) mfma16_f16_inplace(wh[p][st], xl[q], acc[p][q]);
                  else acc[p][q] = mfma16<F16>(wh[p][st], xl[q], acc[p][q]);
                }
              __builtin_amdgcn_sched_barrier(0);
              if (st < 3) rd_xl(st + 1);
              __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (F16) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // last MFMA results -> the epilogue's VALU reads
          }
          pending = true;
          pend_mbase = fa.row0 + rb * kFirstRows;
        }
        XV_FUZZ();
        if (rb + 1 < rb1) commit(buf ^ 1);
      }
    }
  }
}

hipError_t launch_tdnn_first(const FirstArgs& a, int epi_prec, hipStream_t s) {
  if (!FirstLayerApplicable(a.dim, a.noff, a.off) || a.nrows <= 0 || a.nrows % kFirstRows || a.row0 % kFirstRows || !a.wc_hi || !a.wc_lo)
    return hipErrorInvalidValue;
  const int dp = (a.dim + 7) & ~7;
  const int lds = 4 * 384 * 4 + 512 * 256 + 4 * (kFirstRows + 32) * dp * 2 + (96 * 4 + 8) * 16;
  static std::atomic<unsigned long long> attr_done{0};
  int attr_dev = 0;
  if (lds_attr_needed(&attr_done, &attr_dev)) {
    for (const void* f : {(const void*)tdnn_first_kernel<kPrecFp16x3>, (const void*)tdnn_first_kernel<kPrecBf16x3>,
                          (const void*)tdnn_first_kernel<kPrecFp16x3E>, (const void*)tdnn_first_kernel<kPrecFp16x2>}) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
    }
    attr_done.fetch_or(1ull << (attr_dev & 63), std::memory_order_release);
  }
  const int ncg = (a.g.n_tiles * kBN + 511) / 512;
  const int grid = std::max(1, std::min(device_cu_count() / ncg, a.nrows / kFirstRows)) * ncg;
  const char* what;
  switch (epi_prec) {
    case kPrecFp16x3:
      XV_LAUNCH(tdnn_first_kernel<kPrecFp16x3>, dim3(grid), dim3(512), lds, s, a);
      what = "tdnn_first_kernel<fp16x3,act>";
      break;
    case kPrecBf16x3:
      XV_LAUNCH(tdnn_first_kernel<kPrecBf16x3>, dim3(grid), dim3(512), lds, s, a);
      what = "tdnn_first_kernel<bf16x3,act>";
      break;
    case kPrecFp16x3E:
      if (!a.g.out_lo4 || !a.g.out_lo4s) return hipErrorInvalidValue;
      XV_LAUNCH(tdnn_first_kernel<kPrecFp16x3E>, dim3(grid), dim3(512), lds, s, a);
      what = "tdnn_first_kernel<fp16x3,act+lo4>";
      break;
    case kPrecFp16x2:
      XV_LAUNCH(tdnn_first_kernel<kPrecFp16x2>, dim3(grid), dim3(512), lds, s, a);
      what = "tdnn_first_kernel<fp16x3,act1>";
      break;
    default: return hipErrorInvalidValue;
  }
  snprintf(g_last_kernel, sizeof g_last_kernel, "%s", what);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void compact_first_kernel(const uint16_t* src, int ldw, int seg_pad, int n_pad, int noff, int dim,
                                                            uint16_t* dst) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)n_pad * kFirstK) return;
  const int n = (int)(idx / kFirstK), k = (int)(idx - (long)n * kFirstK);
  const int dp = (dim + 7) & ~7;
  const int j = k / dp, d = k - j * dp;
  dst[idx] = (j < noff && d < dim) ? src[(long)n * ldw + j * seg_pad + d] : (uint16_t)0;
}

hipError_t launch_compact_first(const uint16_t* src, int ldw, int seg_pad, int n_pad, int noff, int dim, uint16_t* dst, hipStream_t s) {
  const long total = (long)n_pad * kFirstK;
  hipLaunchKernelGGL(compact_first_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, ldw, seg_pad, n_pad, noff, dim, dst);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// prep_input: packed fp32 feature rows -> aligned 16-bit planes.  One thread = 8 columns of a row.
template <int PREC>
__global__ __launch_bounds__(256) void prep_input_kernel(const PrepArgs a) {
  constexpr bool SPLIT = PrecXPlanes(PREC) == 2;   // activations carry a residual plane
  constexpr bool F16 = PrecF16(PREC);
  const int per_row = a.ld >> 3;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)a.rows * per_row;
  // the group-max tables of this pass (kPrecFp16Mx) are cleared by the first kernel of the pass
  for (long z = idx; z < a.n_zero_words; z += (long)gridDim.x * blockDim.x) a.zero_words[z] = 0u;
  if (idx >= total) return;
  const int row = (int)(idx / per_row);
  const int c8 = (int)(idx - (long)row * per_row) * 8;
  const int u = a.grp_utt[row >> 4];
  const float* src = nullptr;
  if (u >= 0) {
    const int t = row - a.dev_off[u];
    const int len = a.src_off[u + 1] - a.src_off[u];
    if (t < len + a.pad_left + a.pad_right) {
      // edge replication (frame-level outputs): device frame t is source frame clamp(t - pad_left, 0, len-1)
      const int ts = min(max(t - a.pad_left, 0), len - 1);
      src = a.feats + (long)(a.src_off[u] + ts) * a.dim;
    }
  }
  unsigned int hw[4], lw[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    float x0 = 0.f, x1 = 0.f;
    const int c = c8 + 2 * v;
    if (src) {
      if (c < a.dim) x0 = src[c];
      if (c + 1 < a.dim) x1 = src[c + 1];
    }
    const uint16_t h0 = to16<F16>(x0), h1 = to16<F16>(x1);
    hw[v] = (unsigned int)h0 | ((unsigned int)h1 << 16);
    if constexpr (SPLIT) {
      const uint16_t l0 = to16<F16>(x0 - from16<F16>(h0));
      const uint16_t l1 = to16<F16>(x1 - from16<F16>(h1));
      lw[v] = (unsigned int)l0 | ((unsigned int)l1 << 16);
    }
  }
  *(u32x4*)(a.out_hi + (long)row * a.ld + c8) = u32x4{hw[0], hw[1], hw[2], hw[3]};
  if constexpr (SPLIT) *(u32x4*)(a.out_lo + (long)row * a.ld + c8) = u32x4{lw[0], lw[1], lw[2], lw[3]};
}

hipError_t launch_prep_input(const PrepArgs& a, int precision, hipStream_t s) {
  const long total = (long)a.rows * (a.ld >> 3);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  switch (precision) {
    case kPrecBf16x3: XV_LAUNCH(prep_input_kernel<kPrecBf16x3>, grid, block, 0, s, a); break;
    case kPrecBf16: XV_LAUNCH(prep_input_kernel<kPrecBf16>, grid, block, 0, s, a); break;
    case kPrecFp16: XV_LAUNCH(prep_input_kernel<kPrecFp16>, grid, block, 0, s, a); break;
    case kPrecFp16x3: XV_LAUNCH(prep_input_kernel<kPrecFp16x3>, grid, block, 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Front-end.  cmn_prefix: one workgroup per utterance, per-column prefix sums in double (the tool this replaces also
// accumulates in double).  cmn_select: one thread per (kept frame, column): window bounds as in
// Kaldi's SlidingWindowCmn, mean from two prefix rows, out = (float)(x - mean).
// Round 6: the time axis is cut into S = 1024 / dp segments (dp = 32 or 64 >= dim); thread (s, d) sums its segment, the segment
// totals are combined in a fixed order (an exclusive sum over at most 32 of them), and a second pass writes the prefix rows.  One
// thread per column running down the WHOLE utterance (rounds 2-5) was 9000 dependent iterations for a 90-second recording - with
// the recipes' own pipeline on the device (fuse_pipe.h) that loop, not the network, was what a batch of long utterances waited
// for (54 M raw frames/s against 98 M without the front-end).  Every prefix value is a sum in one fixed order that depends on
// the utterance's length only: deterministic, and the same in whatever batch the utterance lands.
__global__ __launch_bounds__(1024) void cmn_prefix_kernel(const FrontEndArgs a) {
  const int u = blockIdx.x;
  const int dp = a.dim <= 32 ? 32 : 64, S = 1024 / dp;
  const int d = threadIdx.x % dp, sgm = threadIdx.x / dp;
  const int r0 = a.raw_off[u], len = a.raw_off[u + 1] - r0;
  const int L = (len + S - 1) / S;
  const int t0 = min(sgm * L, len), t1 = min(t0 + L, len);
  __shared__ double tot[32][64];
  const bool live = d < a.dim;
  const float* x = a.raw + (long)r0 * a.dim + d;
  double acc = 0.0;
  if (live)
    for (int t = t0; t < t1; ++t) acc += (double)x[(long)t * a.dim];
  tot[sgm][d] = acc;
  __syncthreads();
  if (!live) return;
  double base = 0.0;
  for (int k = 0; k < sgm; ++k) base += tot[k][d];
  double* p = a.prefix + ((long)r0 + u) * a.dim + d;
  if (sgm == 0) p[0] = 0.0;
  acc = base;
  for (int t = t0; t < t1; ++t) {
    acc += (double)x[(long)t * a.dim];
    p[(long)(t + 1) * a.dim] = acc;
  }
}

__global__ __launch_bounds__(256) void cmn_select_kernel(const FrontEndArgs a) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)a.n_out * a.dim) return;
  const int i = (int)(idx / a.dim), d = (int)(idx - (long)i * a.dim);
  const int u = a.sel_utt[i];
  const int r0 = a.raw_off[u], T = a.raw_off[u + 1] - r0;
  const int t = a.sel_row[i] - r0;
  const float x = a.raw[(long)a.sel_row[i] * a.dim + d];
  if (a.cmn_window <= 0) {
    a.out[idx] = x;
    return;
  }
  int ws, we;
  if (a.center) {
    ws = t - a.cmn_window / 2;
    we = ws + a.cmn_window;
  } else {
    ws = t - a.cmn_window;
    we = t + 1;
  }
  if (ws < 0) {
    we -= ws;
    ws = 0;
  }
  if (!a.center && we > t) we = max(t + 1, a.min_window);
  if (we > T) {
    ws -= (we - T);
    we = T;
    if (ws < 0) ws = 0;
  }
  const double* p = a.prefix + ((long)r0 + u) * a.dim + d;
  const double mean = (p[(long)we * a.dim] - p[(long)ws * a.dim]) / (double)(we - ws);
  a.out[idx] = (float)((double)x - mean);
}

// cm_expand: one workgroup per (64-row block, utterance).  The column's bytes are contiguous in the object (coalesced reads along
// the rows), the floats go through LDS so that the block's 64 x dim rows leave as one contiguous run.
__global__ __launch_bounds__(256) void cm_expand_kernel(const CmExpandArgs a) {
  // every multiplication and addition below is rounded on its own, like the host reader's: hipcc contracts a * b + c into one
  // fused operation by default (one ulp here and there), and ROCm's __fmul_rn / __fadd_rn are plain operators defined where
  // contraction is allowed - hence plain operators HERE, under the pragma
#pragma clang fp contract(off)
  const int u = blockIdx.y;
  const int r0 = blockIdx.x * 64;
  const int rows = a.raw_off[u + 1] - a.raw_off[u];
  if (r0 >= rows) return;
  const int D = a.dim;
  const uint8_t* obj = a.cm + a.cm_off[u];
  __shared__ float pt[4][64];
  __shared__ float tile[64 * 64];
  const int tid = threadIdx.x;
  auto rd32 = [&](const uint8_t* p) __attribute__((always_inline)) {
    return (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24);
  };
  if (tid < D) {
    const float mn = __uint_as_float(rd32(obj)), range = __uint_as_float(rd32(obj + 4));
    const uint8_t* h = obj + 16 + (size_t)tid * 8;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned v = (unsigned)h[2 * q] | ((unsigned)h[2 * q + 1] << 8);
      // min_value + range * 1.52590218966964e-05F * v, left to right, every operation rounded on its own
      const float scaled = range * 1.52590218966964e-05F;
      const float prod = scaled * (float)v;
      pt[q][tid] = mn + prod;
    }
  }
  __syncthreads();
  const uint8_t* data = obj + 16 + (size_t)D * 8;
  const int nr = min(64, rows - r0);
  for (int i = tid; i < 64 * D; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r >= nr) continue;
    const int v = data[(size_t)c * rows + r0 + r];
    const float p0 = pt[0][c], p25 = pt[1][c], p75 = pt[2][c], p100 = pt[3][c];
    float x;
    if (v <= 64) {
      const float t1 = (p25 - p0) * (float)v, t2 = t1 * (1 / 64.0f);
      x = p0 + t2;
    } else if (v <= 192) {
      const float t1 = (p75 - p25) * (float)(v - 64), t2 = t1 * (1 / 128.0f);
      x = p25 + t2;
    } else {
      const float t1 = (p100 - p75) * (float)(v - 192), t2 = t1 * (1 / 63.0f);
      x = p75 + t2;
    }
    tile[r * D + c] = x;
  }
  __syncthreads();
  float* dst = a.out + ((size_t)a.raw_off[u] + r0) * D;
  for (int i = tid; i < nr * D; i += 256) dst[i] = tile[i];
}

hipError_t launch_cm_expand(const CmExpandArgs& a, hipStream_t s) {
  if (a.dim < 1 || a.dim > 64 || a.n_utts < 1 || a.max_rows < 1) return hipErrorInvalidValue;
  XV_LAUNCH(cm_expand_kernel, dim3((unsigned)((a.max_rows + 63) / 64), (unsigned)a.n_utts), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_frontend(const FrontEndArgs& a, hipStream_t s) {
  if (a.dim > 64) return hipErrorInvalidValue;
  if (a.cmn_window > 0 && a.n_utts > 0) XV_LAUNCH(cmn_prefix_kernel, dim3(a.n_utts), dim3(1024), 0, s, a);
  if (a.n_out > 0) {
    const long total = (long)a.n_out * a.dim;
    XV_LAUNCH(cmn_select_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// frame_output: one workgroup per output row.  LogSoftmaxComponent: y = x - max - log(sum exp(x - max)).
// frame_output_lsm_kernel<N, VEC> keeps the whole row in registers (N floats per thread: rows of up to 256 N columns): one
// read of the logits, block-wide max and sum of exp(x - max) with the accurate expf / logf, one write of x - lse - 3.2 GB
// of traffic for the 102 k x 3856 log-posteriors of BASELINE config 5 instead of the three read passes + one write of the
// first version (1.23 -> 0.6 ms), and no statistics work in the head GEMM's epilogue (tried: +0.15 ms there).  VEC: 16-byte
// loads / stores (row length, row pitches and pointers multiples of four floats), else coalesced 4-byte accesses.
// frame_output_kernel is the plain gather, and the three-pass fallback for rows that do not fit the registers.
template <int N, bool VEC, bool IN16 = false>
__global__ __launch_bounds__(256) void frame_output_lsm_kernel(const FrameOutArgs a) {
  const int o = blockIdx.x;
  const long srow = a.out_row ? a.out_row[o] : o;
  const float* src = IN16 ? nullptr : a.src + srow * a.ld;
  const _Float16* src16 = IN16 ? (const _Float16*)a.src16 + srow * a.ld : nullptr;
  float* dst = a.out + (long)o * a.out_ld;
  const int tid = threadIdx.x;
  __shared__ float red[8];
  float v[N];
  float m = -INFINITY;
  if constexpr (VEC) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
      const int c = (tid + 256 * i) * 4;
      f32x4 x = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      if (c < a.dim) {
        if constexpr (IN16) {
          typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
          x = __builtin_convertvector(*(const f16x4*)(src16 + c), f32x4);
        } else {
          x = *(const f32x4*)(src + c);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[4 * i + k] = x[k];
        m = fmaxf(m, x[k]);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int c = tid + 256 * i;
      v[i] = c < a.dim ? (IN16 ? (float)src16[c] : src[c]) : -INFINITY;
      m = fmaxf(m, v[i]);
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) sum += expf(v[i] - m);   // columns beyond the row: exp(-inf) = 0
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
  __syncthreads();
  const float lse = m + logf((red[4] + red[5]) + (red[6] + red[7]));
  if constexpr (VEC) {
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
      const int c = (tid + 256 * i) * 4;
      if (c < a.dim) *(f32x4*)(dst + c) = f32x4{v[4 * i] - lse, v[4 * i + 1] - lse, v[4 * i + 2] - lse, v[4 * i + 3] - lse};
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int c = tid + 256 * i;
      if (c < a.dim) dst[c] = v[i] - lse;
    }
  }
}

__global__ __launch_bounds__(256) void frame_output_kernel(const FrameOutArgs a) {
  const int o = blockIdx.x;
  const long srow = a.out_row ? a.out_row[o] : o;
  const float* src = a.src + srow * a.ld;
  float* dst = a.out + (long)o * a.out_ld;
  const int tid = threadIdx.x;
  if (!a.log_softmax) {
    for (int c = tid; c < a.dim; c += 256) dst[c] = src[c];
    return;
  }
  __shared__ float red[4];
  float m = -INFINITY;
  for (int c = tid; c < a.dim; c += 256) m = fmaxf(m, src[c]);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int c = tid; c < a.dim; c += 256) sum += expf(src[c] - m);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  const float lse = m + logf((red[0] + red[1]) + (red[2] + red[3]));
  for (int c = tid; c < a.dim; c += 256) dst[c] = src[c] - lse;
}

hipError_t launch_frame_output(const FrameOutArgs& a, hipStream_t s) {
  if (a.n_out <= 0) return hipSuccess;
  const dim3 grid(a.n_out), block(256);
  if (a.src16) {   // fp16 logits (engine: single-pass fp16 mode with a LogSoftmax behind the head)
    if (!a.log_softmax || a.dim > 256 * 64) return hipErrorInvalidValue;
    const bool vec = ((a.dim | a.out_ld | a.ld) & 3) == 0 && ((uintptr_t)a.out & 15) == 0 && ((uintptr_t)a.src16 & 7) == 0;
    const int n = (a.dim + 255) / 256;
    if (vec) {
      if (n <= 8) XV_LAUNCH((frame_output_lsm_kernel<8, true, true>), grid, block, 0, s, a);
      else if (n <= 16) XV_LAUNCH((frame_output_lsm_kernel<16, true, true>), grid, block, 0, s, a);
      else if (n <= 32) XV_LAUNCH((frame_output_lsm_kernel<32, true, true>), grid, block, 0, s, a);
      else XV_LAUNCH((frame_output_lsm_kernel<64, true, true>), grid, block, 0, s, a);
    } else {
      if (n <= 8) XV_LAUNCH((frame_output_lsm_kernel<8, false, true>), grid, block, 0, s, a);
      else if (n <= 16) XV_LAUNCH((frame_output_lsm_kernel<16, false, true>), grid, block, 0, s, a);
      else if (n <= 32) XV_LAUNCH((frame_output_lsm_kernel<32, false, true>), grid, block, 0, s, a);
      else XV_LAUNCH((frame_output_lsm_kernel<64, false, true>), grid, block, 0, s, a);
    }
    return hipGetLastError();
  }
  if (a.log_softmax && a.dim <= 256 * 64) {
    const bool vec = ((a.dim | a.out_ld | a.ld) & 3) == 0 && (((uintptr_t)a.out | (uintptr_t)a.src) & 15) == 0;
    const int n = (a.dim + 255) / 256;   // floats per thread
    if (vec) {
      if (n <= 8) XV_LAUNCH((frame_output_lsm_kernel<8, true>), grid, block, 0, s, a);
      else if (n <= 16) XV_LAUNCH((frame_output_lsm_kernel<16, true>), grid, block, 0, s, a);
      else if (n <= 32) XV_LAUNCH((frame_output_lsm_kernel<32, true>), grid, block, 0, s, a);
      else XV_LAUNCH((frame_output_lsm_kernel<64, true>), grid, block, 0, s, a);
    } else {
      if (n <= 8) XV_LAUNCH((frame_output_lsm_kernel<8, false>), grid, block, 0, s, a);
      else if (n <= 16) XV_LAUNCH((frame_output_lsm_kernel<16, false>), grid, block, 0, s, a);
      else if (n <= 32) XV_LAUNCH((frame_output_lsm_kernel<32, false>), grid, block, 0, s, a);
      else XV_LAUNCH((frame_output_lsm_kernel<64, false>), grid, block, 0, s, a);
    }
    return hipGetLastError();
  }
  XV_LAUNCH(frame_output_kernel, grid, block, 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// pool_finalise: fixed-order reduction of the 16-row partials of one utterance (the order depends
// only on the position inside the utterance, so results do not depend on batch composition).
template <int PREC>
__global__ __launch_bounds__(256) void pool_finalise_kernel(const PoolArgs a) {
  constexpr bool SPLIT = PrecXPlanes(PREC) == 2;   // activations carry a residual plane
  constexpr bool F16 = PrecF16(PREC);
  const int b = blockIdx.y;
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= a.dim) return;
  const int g0 = a.utt_grp0[b], g1 = a.utt_grp1[b];
  double s1 = 0.0, s2 = 0.0;
  for (int g = g0; g < g1; ++g) {
    const float* p = a.partial + (long)g * 2 * a.ldp + col;
    s1 += (double)p[0];
    s2 += (double)p[a.ldp];
  }
  // Totals are accumulated in double (fixed order), then the moments are formed with the SAME fp32
  // rounding sequence as Kaldi's StatisticsPoolingComponent (scale by 1/n, x2 - mean*mean with separately
  // rounded product, floor, pow 0.5): degenerate columns (one pooled frame, constant activations) then
  // cancel exactly like the reference instead of leaving ~1e-7*x^2 of rounding noise above the 1e-10 floor.
  const float n = (float)a.utt_count[b];
  const float mu = __fdiv_rn((float)s1, n);
  const float ex2 = __fdiv_rn((float)s2, n);
  float m2 = mu * mu;
  asm volatile("" : "+v"(m2));  // keep the product separately rounded: hipcc would contract it into an FMA
  float var = ex2 - m2;
  var = fmaxf(var, a.var_floor);
  const float sd = __fsqrt_rn(var);
  const long base = (long)b * a.ld;
  const uint16_t mh = to16<F16>(mu), sh = to16<F16>(sd);
  a.out_hi[base + col] = mh;
  a.out_hi[base + a.dim + col] = sh;
  if constexpr (SPLIT) {
    a.out_lo[base + col] = to16<F16>(mu - from16<F16>(mh));
    a.out_lo[base + a.dim + col] = to16<F16>(sd - from16<F16>(sh));
  }
}

hipError_t launch_pool_finalise(const PoolArgs& a, int precision, hipStream_t s) {
  dim3 grid((a.dim + 255) / 256, a.B), block(256);
  switch (precision) {
    case kPrecBf16x3: XV_LAUNCH(pool_finalise_kernel<kPrecBf16x3>, grid, block, 0, s, a); break;
    case kPrecBf16: XV_LAUNCH(pool_finalise_kernel<kPrecBf16>, grid, block, 0, s, a); break;
    case kPrecFp16: XV_LAUNCH(pool_finalise_kernel<kPrecFp16>, grid, block, 0, s, a); break;
    case kPrecFp16x3: XV_LAUNCH(pool_finalise_kernel<kPrecFp16x3>, grid, block, 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Speaker-level back-end.  y = T (x - mean) is a small fp32 GEMM ([n x dim] . [dim x rows], 58 FLOP per byte of its
// own traffic: bound by the fp32 matrix rate, not by HBM), done with v_mfma_f32_16x16x4_f32 - exact fp32 products,
// fp32 accumulation in a fixed order.  One workgroup = 64 vectors (one wave = 16 of them) x up to 256 output rows;
// the transform tile [rows x 32] and the mean-subtracted vector tile [64 x 32] go through LDS (row pitch 36 floats:
// the 64 lanes of a fragment read hit 64 different banks).  The length normalisation is fused: a lane owns 4 rows of
// every 16-row fragment of one vector, so |y|^2 is an in-register sum plus two cross-lane adds.
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
constexpr int kBkVB = 64, kBkKT = 32, kBkLD = 36, kBkMaxRF = 16;

template <int RF>   // 16-row fragments of the transform handled per workgroup (rows beyond t_rows are zero)
__global__ __launch_bounds__(256) void backend_gemm_kernel(const BackendArgs a) {
  __shared__ float ts[RF * 16 * kBkLD];
  __shared__ float xs[kBkVB * kBkLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int v0 = blockIdx.x * kBkVB;
  const int fi = lane & 15, fk = lane >> 4;
  f32x4_t acc[RF];
#pragma unroll
  for (int r = 0; r < RF; ++r) acc[r] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // register staging of the next K tile (global loads in flight while the current tile is multiplied)
  constexpr int NT = RF * 16 / 8, NX = kBkVB / 8;
  float tr[NT], xr[NX];
  const int kk = tid & 31, r8 = tid >> 5;
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int r = r8 + 8 * i;
      tr[i] = (r < a.t_rows && k0 + kk < a.dim) ? a.t[(long)r * a.t_cols + k0 + kk] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int v = r8 + 8 * i;
      float x = 0.f;
      if (v0 + v < a.n && k0 + kk < a.dim) {
        x = a.x[(long)(v0 + v) * a.ldx + k0 + kk];
        if (a.mean) x -= a.mean[k0 + kk];
      }
      xr[i] = x;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < a.dim; k0 += kBkKT) {
    __syncthreads();
    // thread -> (row = tid / 32 + 8 * i, k = tid % 32): 128 contiguous bytes per row
#pragma unroll
    for (int i = 0; i < NT; ++i) ts[(r8 + 8 * i) * kBkLD + kk] = tr[i];
#pragma unroll
    for (int i = 0; i < NX; ++i) xs[(r8 + 8 * i) * kBkLD + kk] = xr[i];
    __syncthreads();
    if (k0 + kBkKT < a.dim) fetch(k0 + kBkKT);
#pragma unroll
    for (int k4 = 0; k4 < kBkKT / 4; ++k4) {
      const float b = xs[(wave * 16 + fi) * kBkLD + k4 * 4 + fk];
#pragma unroll
      for (int r = 0; r < RF; ++r) {
        const float w = ts[(r * 16 + fi) * kBkLD + k4 * 4 + fk];
        acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, acc[r], 0, 0, 0);
      }
    }
  }
  // lane: vector v0 + wave*16 + fi, rows r*16 + fk*4 + (0..3)
  const int v = v0 + wave * 16 + fi;
  const int out_dim = a.t_rows;
  float ss = 0.f;
#pragma unroll
  for (int r = 0; r < RF; ++r) {
    {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = r * 16 + fk * 4 + q;
        float y = acc[r][q];
        if (row < out_dim) {
          if (a.t_cols == a.dim + 1) y += a.t[(long)row * a.t_cols + a.dim];
        } else {
          y = 0.f;
        }
        acc[r][q] = y;
        ss = fmaf(y, y, ss);
      }
    }
  }
  ss += __shfl_xor(ss, 16);
  ss += __shfl_xor(ss, 32);
  float scale = 1.f;
  if (a.normalize || a.ratio) {
    const float norm = sqrtf(ss);
    const float ratio = a.scaleup ? norm / sqrtf((float)out_dim) : norm;
    if (a.ratio && fk == 0 && v < a.n) a.ratio[v] = ratio;
    if (a.normalize && ratio != 0.f) scale = 1.f / ratio;
  }
  if (v < a.n) {
    float* dst = a.out + (long)v * a.ldo;
#pragma unroll
    for (int r = 0; r < RF; ++r) {
      {
        const int row = r * 16 + fk * 4;
        if (row + 3 < out_dim && (a.ldo & 3) == 0 && (out_dim & 3) == 0) {
          *(f32x4_t*)(dst + row) = acc[r] * scale;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (row + q < out_dim) dst[row + q] = acc[r][q] * scale;
        }
      }
    }
  }
}

// No transform (or normalisation of an already transformed table): one wave per vector, streaming.
__global__ __launch_bounds__(256) void backend_rowwise_kernel(const BackendArgs a) {
  const int lane = threadIdx.x & 63;
  const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= a.n) return;
  const float* src = a.x + (long)v * a.ldx;
  float ss = 0.f;
  for (int k = lane; k < a.dim; k += 64) {
    float x = src[k];
    if (a.mean) x -= a.mean[k];
    ss = fmaf(x, x, ss);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) ss += __shfl_xor(ss, d);
  float scale = 1.f;
  if (a.normalize || a.ratio) {
    const float norm = sqrtf(ss);
    const float ratio = a.scaleup ? norm / sqrtf((float)a.dim) : norm;
    if (a.ratio && lane == 0) a.ratio[v] = ratio;
    if (a.normalize && ratio != 0.f) scale = 1.f / ratio;
  }
  float* dst = a.out + (long)v * a.ldo;
  for (int k = lane; k < a.dim; k += 64) {
    float x = src[k];
    if (a.mean) x -= a.mean[k];
    dst[k] = x * scale;
  }
}

static hipError_t launch_backend_gemm(const BackendArgs& a, hipStream_t s) {
  const dim3 grid((a.n + kBkVB - 1) / kBkVB), block(256);
  const int rf = (a.t_rows + 15) / 16;
  if (rf <= 2) XV_LAUNCH(backend_gemm_kernel<2>, grid, block, 0, s, a);
  else if (rf <= 4) XV_LAUNCH(backend_gemm_kernel<4>, grid, block, 0, s, a);
  else if (rf <= 7) XV_LAUNCH(backend_gemm_kernel<7>, grid, block, 0, s, a);
  else if (rf <= 10) XV_LAUNCH(backend_gemm_kernel<10>, grid, block, 0, s, a);
  else if (rf <= 13) XV_LAUNCH(backend_gemm_kernel<13>, grid, block, 0, s, a);
  else XV_LAUNCH(backend_gemm_kernel<16>, grid, block, 0, s, a);
  return hipGetLastError();
}

hipError_t launch_backend(const BackendArgs& a, hipStream_t s) {
  if (a.n <= 0) return hipSuccess;
  if (a.dim < 1 || (a.t && a.t_cols != a.dim && a.t_cols != a.dim + 1)) return hipErrorInvalidValue;
  if (!a.t) {
    XV_LAUNCH(backend_rowwise_kernel, dim3((a.n + 3) / 4), dim3(256), 0, s, a);
    return hipGetLastError();
  }
  if (a.t_rows <= kBkMaxRF * 16) return launch_backend_gemm(a, s);
  // more than 256 output rows: blocks of 256 rows without normalisation, then the row-wise pass on the result
  for (int r0 = 0; r0 < a.t_rows; r0 += kBkMaxRF * 16) {
    BackendArgs b = a;
    b.t = a.t + (long)r0 * a.t_cols;
    b.t_rows = min(kBkMaxRF * 16, a.t_rows - r0);
    b.out = a.out + r0;
    b.normalize = 0;
    b.ratio = nullptr;
    hipError_t e = launch_backend_gemm(b, s);
    if (e != hipSuccess) return e;
  }
  if (a.normalize || a.ratio) {
    BackendArgs c = a;
    c.x = a.out;
    c.ldx = a.ldo;
    c.dim = a.t_rows;
    c.mean = nullptr;
    c.t = nullptr;
    XV_LAUNCH(backend_rowwise_kernel, dim3((a.n + 3) / 4), dim3(256), 0, s, c);
    return hipGetLastError();
  }
  return hipSuccess;
}

template <typename ACC>
__global__ __launch_bounds__(256) void segment_mean_kernel(const SegMeanArgs a) {
  const int s = blockIdx.x;
  const int b = a.seg_off[s], e = a.seg_off[s + 1];
  for (int k = threadIdx.x; k < a.dim; k += 256) {
    ACC acc = 0;
    for (int i = b; i < e; ++i) acc += (ACC)a.x[(long)a.idx[i] * a.ldx + k];   // list order, like the AddVec loop
    float m = 0.f;
    if (e > b) {
      if constexpr (sizeof(ACC) == 8) m = (float)(acc * (1.0 / (double)(e - b)));
      else m = acc * (float)(1.0 / (double)(e - b));
    }
    a.out[(long)s * a.dim + k] = m;
  }
}

hipError_t launch_segment_mean(const SegMeanArgs& a, hipStream_t s) {
  if (a.n_seg <= 0) return hipSuccess;
  if (a.acc64) XV_LAUNCH(segment_mean_kernel<double>, dim3(a.n_seg), dim3(256), 0, s, a);
  else XV_LAUNCH(segment_mean_kernel<float>, dim3(a.n_seg), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
uint16_t host_f32_to_bf16(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
float host_bf16_to_f32(uint16_t h) {
  uint32_t u = ((uint32_t)h) << 16;
  float x;
  memcpy(&x, &u, 4);
  return x;
}
uint16_t host_f32_to_f16(float x) {
  _Float16 h = (_Float16)x;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}
float host_f16_to_f32(uint16_t u) {
  _Float16 h;
  memcpy(&h, &u, 2);
  return (float)h;
}

}  // namespace xv
