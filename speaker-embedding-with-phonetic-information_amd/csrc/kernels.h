// Device-kernel interface of the MI355X (gfx950) x-vector / c-vector extractor.
//
// Everything here is plain C++ (no torch, no HIP types beyond hipStream_t) so that the
// engine (engine.cc) and the kernel unit-test entry points in the C ABI can share it.
//
// Data layout in HBM ("flat packed frames"):
//   * every frame-level activation is a pair of 16-bit planes  hi[rows][ld], lo[rows][ld]
//     (bf16 hi + bf16 residual in the split-precision mode, a single bf16/fp16 plane in
//     the single-pass modes); rows are the frames of ALL utterances of the batch packed
//     back to back, each utterance starting on a 16-row boundary;
//   * a spliced affine layer  z[t] = W . concat_j y_src_j[t + off_j] + b  (SURVEY.md B.7;
//     reference graphs: egs/sre/v2/local/nnet3/xvector/run_xvector_new.sh:95-99) is a
//     GEMM whose K axis is the concatenation of "segments": segment j reads rows shifted
//     by off_j of plane pair src_j.  No spliced matrix is ever materialised.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xv {

constexpr int kBM = 128;      // rows (frames) per workgroup tile
constexpr int kBN = 128;      // output columns per workgroup tile
constexpr int kBK = 32;       // K elements per pipeline step (one 16x16x32 MFMA deep)
constexpr int kMaxSeg = 8;    // K segments (Append() terms) per layer
constexpr int kRowAlign = 16; // every utterance starts on a multiple of this many rows

enum Precision : int {
  kPrecBf16x3 = 0,  // split bf16: hi*hi + hi*lo + lo*hi, fp32 accumulate (parity mode)
  kPrecBf16 = 1,    // single-pass bf16 MFMA
  kPrecFp16 = 2,    // single-pass fp16 MFMA
  kPrecFp16x3 = 3,  // split fp16: hi*hi + hi*lo + lo*hi (22-bit operands), fp32 accumulate
  kPrecFp16x2 = 4,  // fp16 activations (one plane) x split fp16 weights (hi + lo): two MFMAs per product.  The weight
                    // rounding error - coherent over the frames of a chunk, so the pooling does not average it - is
                    // removed; the activation rounding error is independent per frame and averages out in the
                    // statistics pooling (measured: DESIGN.md section 3.1).  Frame-level layers only.
  kPrecAuto = 5,    // engine policy, not a kernel mode: the fast mode (kPrecFp16Mx where a layer's K walk allows it,
                    // else kPrecFp16x2) for chunks that pool >= a threshold of frames, kPrecFp16x3 for the others (same
                    // packed weights)
  kPrecFp16Mx = 6,  // kPrecFp16x2 with the second pass (x . w_lo) done at 4x the fp16 rate on block-scaled fp4 operands:
                    //   x . w_hi  (v_mfma_f32_16x16x32_f16, every 32-column K step)
                    // + q4(x) . q4(w - w_hi)  (v_mfma_scale_f32_16x16x128_f8f6f4, once per four K steps),
                    // q4(x) = the fp16 fragments converted in registers (v_cvt_scalef32_pk_fp4_f16) with one power-of-two
                    // scale per 16-row group (from the max |x| the producing epilogue recorded), q4(w - w_hi) = a packed
                    // e2m1 plane with one scale per output row.  1.25 MFMA passes per product; the residual term only has
                    // to be good to ~4 bits (it is 2^-11 of the product), DESIGN.md section 3.0.  Frame-level layers only.
  kPrecFp16Mx2 = 7, // kPrecFp16Mx plus a third term for what the fp16 rounding of the ACTIVATIONS dropped:
                    // + q4(y - fp16(y)) . q4(w)  (the same block-scaled MFMA, once per 128 columns of K): the producing
                    // epilogue writes, next to the fp16 plane, the e2m1 image of its rounding residual (one scale per row
                    // and 64 columns), the packer a 4-bit image of the weights; the consumer walks K a second time in
                    // 128-column steps whose 4-bit tiles have the shape of the fp16 tiles.  1.5 passes per product; the
                    // error no longer depends on how well the pooling averages the activation rounding.
  kPrecFp16x3E = 8, // kPrecFp16x3 whose planes epilogue emits (fp16 plane, 4-bit residual) instead of (hi, lo): the layers
                    // of a kPrecFp16Mx2 pass that cannot run it themselves (K walk not in whole 128-column blocks)
  kPrecFp16MxE = 9, // kPrecFp16Mx whose planes epilogue emits (fp16 plane, 4-bit residual): a layer of a kPrecFp16Mx2 pass that
                    // the calibration lets run the 1.25-pass arithmetic ("lite" layers, Engine::SetLiteMask) in front of a
                    // consumer that still walks the residual plane
};
constexpr bool PrecF16(int p) { return p == kPrecFp16 || p == kPrecFp16x3 || p == kPrecFp16x2 || p == kPrecAuto || p == kPrecFp16Mx || p == kPrecFp16Mx2 || p == kPrecFp16x3E || p == kPrecFp16MxE; }
constexpr bool PrecMx(int p) { return p == kPrecFp16Mx || p == kPrecFp16Mx2 || p == kPrecFp16MxE; }
constexpr bool PrecMx2(int p) { return p == kPrecFp16Mx2; }
constexpr bool PrecEmitsLo4(int p) { return p == kPrecFp16Mx2 || p == kPrecFp16x3E || p == kPrecFp16MxE; }   // planes epilogue: fp16 plane + 4-bit residual
constexpr int PrecXPlanes(int p) { return (p == kPrecBf16x3 || p == kPrecFp16x3 || p == kPrecFp16x3E) ? 2 : 1; }   // kernel modes only
// weight planes staged per K step (the 4-bit residual plane of kPrecFp16Mx uses the slot of the fp16 residual plane)
constexpr int PrecWPlanes(int p) { return (p == kPrecBf16x3 || p == kPrecFp16x3 || p == kPrecFp16x2 || p == kPrecAuto || p == kPrecFp16Mx || p == kPrecFp16Mx2 || p == kPrecFp16x3E || p == kPrecFp16MxE) ? 2 : 1; }
constexpr int PrecPasses(int p) { return PrecXPlanes(p) + PrecWPlanes(p) - 1; }   // MFMAs per algorithmic product (Mx: 1.25)

enum Epilogue : int {
  kEpiAct = 0,    // bias -> ReLU? -> BatchNorm? -> split to 16-bit planes
  kEpiF32 = 1,    // bias -> ReLU? -> BatchNorm? -> fp32 rows (embedding / raw affine output)
  kEpiStats = 2,  // bias -> ReLU? -> BatchNorm? -> per-16-row partial (sum, sum of squares)
  kEpiSplitK = 3, // internal: raw fp32 accumulators of one K slice into the split-K workspace
};

struct Seg {
  const uint16_t* hi;  // plane pointer at logical row 0 (halo rows live at negative indices)
  const uint16_t* lo;  // may be null in single-pass modes
  int ld;              // leading dimension in elements
  int row_shift;       // time offset of this Append() term
  int ksteps;          // K length of the segment / kBK
  int pad_;
  const unsigned* gmax;  // kPrecFp16Mx: [rows/16] max |x| (float bits) of the 16-row groups of this plane, or null
  // kPrecFp16Mx2: e2m1 image of (activation - its fp16 plane), [rows][ld / 2 bytes] (column c = nibble c & 1 of byte
  // c / 2), and its E8M0 scales [rows][Lo4ScalePitch(ld)] (one per row and 64 columns); both at logical row 0 like hi
  const uint8_t* lo4;
  const uint8_t* lo4s;
};
// bytes per row of a plane's residual-scale table: ld / 64 scales, padded to whole dwords (the consumers stage it with
// 4-byte LDS-DMA)
constexpr int Lo4ScalePitch(int ld) { return ((ld >> 6) + 3) & ~3; }

// A group of K segments that share one LDS activation tile (filled by the launcher, see kernels.hip).
struct Grp {
  const uint16_t* hi;
  const uint16_t* lo;
  int ld;
  int ksteps;   // 32-column chunks per segment
  int nshift;   // time offsets in the group
  int shift0;   // first (smallest) offset
  int dstep;    // spacing of the offsets
  int wcol0;    // weight column of (offset 0, chunk 0)
  int wstride;  // weight columns between consecutive offsets
  int pad_;
  const unsigned* gmax;   // see Seg
  const uint8_t* lo4s;    // kPrecFp16Mx2, groups of the second K walk (hi = the 4-bit plane, ld = its row pitch / 2): scales
  int ld4s;               // bytes per row of lo4s
  int pad2_;
};

// The order in which the GEMM kernels walk the K steps of a layer: consecutive Append() terms that read the same
// source at uniformly spaced, increasing time offsets (total span <= 16 rows) form a group whose activation tile is
// staged once per 32-column chunk; inside a group the walk is chunk -> offset.  Shared by the launcher (build_groups)
// and by the host code that packs the 4-bit residual plane of kPrecFp16Mx in walk order (engine.cc).
struct WalkGroup {
  int first_seg, nshift, ksteps, shift0, dstep, wcol0, wstride;
};
// src_key[j] equal <=> segments read the same plane (and leading dimension).  Returns the number of groups.
inline int PlanWalkGroups(int nseg, const long* src_key, const int* row_shift, const int* ksteps, WalkGroup* out) {
  int ng = 0, wcol = 0;
  for (int j = 0; j < nseg;) {
    WalkGroup& G = out[ng++];
    G.first_seg = j;
    G.ksteps = ksteps[j];
    G.nshift = 1;
    G.shift0 = row_shift[j];
    G.dstep = 0;
    G.wcol0 = wcol;
    G.wstride = ksteps[j] * kBK;
    int k = j + 1;
    while (k < nseg && src_key[k] == src_key[j] && ksteps[k] == ksteps[j]) {
      const int d = row_shift[k] - row_shift[k - 1];
      if (d <= 0 || (G.nshift > 1 && d != G.dstep) || row_shift[k] - row_shift[j] > 16) break;
      G.dstep = d;
      ++G.nshift;
      ++k;
    }
    wcol += G.nshift * G.wstride;
    j = k;
  }
  return ng;
}
// Weight column (element index into the K-contiguous fp16 planes) of every K step in walk order; returns the number
// of steps.  mx_ok (optional) = every group has a multiple of four steps, i.e. no 128-deep block of the walk
// straddles two sources.
inline int PlanWalkSteps(int ng, const WalkGroup* g, int* step_wcol, int cap, bool* mx_ok) {
  int n = 0;
  bool ok = true;
  for (int i = 0; i < ng; ++i) {
    if ((g[i].ksteps * g[i].nshift) % 4) ok = false;
    for (int kk = 0; kk < g[i].ksteps; ++kk)
      for (int ij = 0; ij < g[i].nshift; ++ij) {
        if (n < cap) step_wcol[n] = g[i].wcol0 + ij * g[i].wstride + kk * kBK;
        ++n;
      }
  }
  if (mx_ok) *mx_ok = ok;
  return n;
}

// The same for tdnn_gemm_kernel_p8, which walks K in tiles of 64 columns: group -> 64-column chunk -> offset -> the chunk's two
// 32-column halves.  A block of the 4-bit residual plane (four consecutive steps) is then a pair of consecutive K tiles.
inline int PlanWalkSteps64(int ng, const WalkGroup* g, int* step_wcol, int cap, bool* ok64) {
  int n = 0;
  bool ok = true;
  for (int i = 0; i < ng; ++i) {
    if (g[i].ksteps % 4) ok = false;
    for (int c = 0; c < g[i].ksteps / 2; ++c)
      for (int ij = 0; ij < g[i].nshift; ++ij)
        for (int k = 0; k < 2; ++k) {
          if (n < cap) step_wcol[n] = g[i].wcol0 + ij * g[i].wstride + (2 * c + k) * kBK;
          ++n;
        }
  }
  if (ok64) *ok64 = ok;
  return n;
}

// kPrecFp16Mx2 on tdnn_gemm_kernel_p8: first weight column of every 128-column step of ITS second walk - tiles of 256 columns
// (two steps), group -> 256-column chunk -> offset.  Needs every group to have a multiple of eight 32-column steps.
inline int PlanWalkLoSteps64(int ng, const WalkGroup* g, int* lo_wcol, int cap) {
  int n = 0;
  for (int i = 0; i < ng; ++i)
    for (int c = 0; c < g[i].ksteps / 8; ++c)
      for (int ij = 0; ij < g[i].nshift; ++ij)
        for (int k = 0; k < 2; ++k) {
          if (n < cap) lo_wcol[n] = g[i].wcol0 + ij * g[i].wstride + (2 * c + k) * 4 * kBK;
          ++n;
        }
  return n;
}

// kPrecFp16Mx2: first weight column of every 128-column step of the second walk (same groups and order: chunk -> offset);
// returns the number of steps.  Needs every group to have a multiple of four 32-column steps.
inline int PlanWalkLoSteps(int ng, const WalkGroup* g, int* lo_wcol, int cap) {
  int n = 0;
  for (int i = 0; i < ng; ++i)
    for (int kq = 0; kq < g[i].ksteps / 4; ++kq)
      for (int ij = 0; ij < g[i].nshift; ++ij) {
        if (n < cap) lo_wcol[n] = g[i].wcol0 + ij * g[i].wstride + kq * 4 * kBK;
        ++n;
      }
  return n;
}

struct GemmArgs {
  Seg seg[kMaxSeg];
  int nseg;
  Grp grp[2 * kMaxSeg];   // kPrecFp16Mx2: entries ngrp .. ngrp + ngrp_lo - 1 are the groups of the second K walk
  int ngrp;
  int total_ksteps;     // K steps of the first walk (32 columns each)
  int ngrp_lo;          // kPrecFp16Mx2 (set by the launcher)
  int lo_ksteps;        // K steps of the second walk (128 columns each): total_ksteps / 4
  const uint16_t* w_hi;  // [n_pad][ldw] row-major (Kaldi <LinearParams> orientation)
  const uint16_t* w_lo;
  int ldw;
  // kPrecFp16Mx: e2m1 residual plane [n_pad][ldw4 bytes]: 64 bytes per (row, block of four K steps in walk order) =
  // four 16-byte lane-group chunks g; nibble e of chunk g = weight column step_wcol[4 b + e / 8] + 8 g + e % 8;
  // value = (w - w_hi) / 2^(w4_scale[row][4 b + g] - 127): one scale per lane-group chunk, the granularity of the
  // scale operand of v_mfma_scale_f32_16x16x128_f8f6f4 (with one scale per row the 4-bit residual left 1/6 of the
  // weight rounding error, 7.0e-5 on the embedding; per chunk: 5.5e-5)
  const uint8_t* w4;
  int ldw4;
  const uint8_t* w4_scale;   // n_pad * total_ksteps E8M0 bytes in staging order (engine.h TileMxScales): 512 per (tile, block)
  // kPrecFp16Mx2: e2m1 image of the weights themselves for the second K walk, [n_pad][ldw4b bytes]: 64 bytes per (row,
  // step of that walk) = four 16-byte lane-group chunks g of 32 consecutive columns; scales as for w4, one per chunk,
  // 512 bytes per (tile, step) in staging order
  const uint8_t* w4b;
  int ldw4b;
  const uint8_t* w4b_scale;
  int m_tiles;           // rows / kBM
  int n_tiles;           // n_pad / kBN
  int relu;
  int bn;
  const float* bias;     // [n_pad]
  const float* scale;    // [n_pad] test-mode BatchNorm scale
  const float* offset;   // [n_pad] test-mode BatchNorm offset
  // kEpiAct
  uint16_t* out_hi;
  uint16_t* out_lo;
  int ldo;
  uint8_t* out_lo4;      // precisions with PrecEmitsLo4: 4-bit residual plane [rows][ldo / 2] + scales [rows][ldo / 64] (Seg)
  uint8_t* out_lo4s;
  unsigned* gmax_out;    // [rows/16] atomic max of |y| (float bits) per 16-row group over all columns (zeroed by the
                         // caller; prep_input does it), or null - what a kPrecFp16Mx consumer scales its fp4 copy by
  const int8_t* out_range;  // with gmax_out: [rows/16][2] first / last (exclusive) row of each group that is a computable
                            // frame of this layer; the others (chunk edges, computed from the neighbouring chunk's
                            // frames; alignment padding) must not enter the maximum, or a chunk's scales - and with
                            // them the last bits of its embedding - would depend on its neighbours in the batch.
                            // null: every row counts
  // kEpiF32
  float* out_f32;
  int ldf;
  int m_valid;           // rows >= m_valid are not stored (kEpiF32 only)
  // kEpiStats
  float* partial;        // [rows/16][2][ldp]
  int ldp;
  const int8_t* grp_range;  // [rows/16][2] first/last(exclusive) valid row inside the 16-row group
  // split-K (small-M layers after the pooling): ksplit > 1 -> K steps are divided over gridDim.y slices whose raw
  // accumulators go to splitk_ws[slice][rows][n_pad]; a second kernel adds them in slice order and applies the epilogue
  int ksplit;
  int ksteps_per_slice;
  float* splitk_ws;
  // start stagger of the 256x128 variant (set by the launcher): the first stagger_wgs workgroups sleep up to
  // stagger_units x 2048 cycles, see the kernel
  int stagger_wgs, stagger_units;
  // stream-K variant (set by the launcher): row tiles of the launch, workspace slots of raw accumulators
  // [workgroup][tile rows x 128], one flag per workgroup, value a flag must hold to count for this launch
  int sk_mtiles;
  int sk_lanes;          // column lanes per workgroup group (divides n_tiles)
  float* sk_ws;
  unsigned* sk_flags;
  unsigned sk_epoch;
  unsigned* sk_error;    // host-mapped word: of the launch stream: receives sk_epoch when a flag wait timed out (see sk_take_error)
  // tdnn_gemm_kernel_p8 (256 x 256 tiles, K tiles of 64 columns, walk order group -> 64-column chunk -> offset: PlanWalkSteps64).
  // p8 != 0: launch that kernel - the caller's statement that this layer runs it for every launch of the mode (its sums are
  // formed in another order than the 32-column kernels') and, for kPrecFp16Mx, that w4 / w4_scale are packed in its walk order.
  int p8;
  int p8_ktiles;         // K tiles of an output tile (set by the launcher)
  int p8_whole;          // partition policy (caller): 0 = K tiles dealt out evenly (stream-K exchange), 1 = whole output tiles only, 2 = whole
                         // tiles for layers of <= 8 K tiles; the launcher resolves it to 0 / 1 for the kernel
  int p8_ktiles_lo;      // kPrecFp16Mx2: tiles of the second walk (256 4-bit columns each); w4b / w4b_scale are then in ITS order
                         // (PlanWalkLoSteps64)
};

// Arms a (start, stop) event pair for the kernels of the NEXT launch_* call of this thread (profiling; see kernels.hip).
// Pass (nullptr, nullptr) to disarm.
void set_launch_events(hipEvent_t start, hipEvent_t stop);

// Launches the spliced-affine GEMM. Returns hipSuccess or the launch error.
hipError_t launch_tdnn_gemm(const GemmArgs& a, int precision, int epilogue, hipStream_t s);
// Name of the kernel instantiation the calling thread's last launch_tdnn_gemm call launched, e.g.
// "tdnn_gemm_kernel_sk<fp16mx,act,8>" (profiling labels; valid until the next launch of this thread).
const char* last_gemm_kernel();
// True when kPrecFp16Mx can run this launch (residual plane + scales present, every K group a multiple of four steps
// with a group-max table, even tile count); otherwise the caller launches kPrecFp16x2 on the same operands.
bool gemm_mx_applicable(const GemmArgs& a);
bool gemm_mx2_applicable(const GemmArgs& a);   // kPrecFp16Mx2: also the 4-bit planes, sources of whole 128-column steps
// The 1.5-pass arithmetic on tdnn_gemm_kernel_p8 (round 5: the race of its second walk's scale staging is fixed, every launch
// test bit-exact and bit-stable under load).  Measured against tdnn_gemm_kernel_sk<fp16mx2> on the bench workload it wins on the
// layers WITHOUT time offsets (tdnn4 0.130 against 0.141 ms, tdnn5 0.282 against 0.288) and loses on those with (tdnn2 / tdnn3
// 0.276 / 0.271 against 0.254 / 0.249: it recomputes its staging offsets per DMA for want of registers), so the packer gives
// only the former its weight image (engine.cc PackModel, blob version 7) and the rule is a property of the layer, never of a launch.
constexpr bool kP8Mx2Built = true;
// tdnn_gemm_kernel_p8 can run this launch in kPrecFp16 / kPrecFp16Mx (see GemmArgs::p8)
bool gemm_p8_applicable(const GemmArgs& a, int precision);
// Stream-K workspace (partial-tile exchange) of a stream: allocated on first use, released by the owner of the
// stream before it destroys it (Engine::~Engine).  sk_take_error: non-zero when a stream-K launch on stream s (of the
// current device) timed out waiting for another workgroup's partial tile since the last call; the word is per stream and
// is cleared by the call, so one engine's fault neither poisons the others nor repeats (checked by the engine after it
// synchronises the stream).
void release_stream_workspace(hipStream_t s);
unsigned sk_take_error(hipStream_t s);

// First layer(s) of the network - every Append() term reads the network INPUT (run_xvector_new.sh:95: tdnn1 over
// Append(-2,-1,0,1,2) of the 23 MFCCs; train_am.sh:32 the same for the phonetic branch) - as one kernel that replaces
// prep_input + the generic GEMM: the fp32 feature rows are read where the caller left them (packed utterances ->
// device rows through the plan's tables, with the edge replication of frame-level outputs), split into fp16 hi / lo in LDS,
// and multiplied in the three-pass arithmetic against weights that stay in REGISTERS for the whole launch (K is short:
// noff * dim columns, compacted to noff * roundup(dim, 8) <= 128 - 115 -> 120 instead of the 5 x 32 of the generic walk),
// so the kernel's only traffic is the planes it writes.  Epilogue = the generic planes epilogue (kernels.hip).
constexpr int kFirstK = 128;          // compact K of the weight image: k' = j * roundup(dim, 8) + d, zero elsewhere
constexpr int kFirstRows = 64;        // frames per work unit (x all columns of a 512-column group)
constexpr int kFirstMaxSlots = 4;     // staged elements per thread and unit
inline bool FirstLayerApplicable(int dim, int noff, const int* off) {
  if (noff < 1 || noff > 8 || dim < 1) return false;
  const int dp = (dim + 7) / 8 * 8;
  int lo = off[0], hi = off[0];
  for (int j = 1; j < noff; ++j) {
    lo = off[j] < lo ? off[j] : lo;
    hi = off[j] > hi ? off[j] : hi;
  }
  // dp <= 24: the kernel's LDS (tables + the 128 KiB weight plane + four feature buffers of 96 rows x dp halves) must fit 160 KiB
  return dp <= 24 && noff * dp <= kFirstK && hi - lo <= 30 && (kFirstRows + hi - lo) * dp <= 512 * kFirstMaxSlots;
}
struct FirstArgs {
  GemmArgs g;               // the epilogue's side: n_tiles, relu, bn, bias / scale / offset, out_hi / out_lo / out_lo4 / out_lo4s,
                            // ldo, gmax_out, out_range - all indexed by ABSOLUTE device row (no region shift)
  const float* feats;       // as PrepArgs
  long feats_valid_idx;     // element index of a float that exists (first row of the batch: row_offsets[0] * dim): what the
                            // unconditional loads of slots without a source frame read.  (feats itself is the caller's base
                            // shifted DOWN by row_offsets[0] rows, so element 0 may lie outside the allocation.)
  // where the frames of every 16-row group come from (built with the plan): {source row of the group's frame 0 before
  // the edge clamp, frames the chunk holds from there on (replicated edge frames included; 0: the group lies outside every
  // chunk), source row of the chunk's first frame, of its last frame}
  const int4* grp_src;
  int rows;                 // device rows of the batch (extent of the table: rows / 16 groups)
  int dim;
  int row0, nrows;          // this launch computes device rows [row0, row0 + nrows), both multiples of 64
  int noff;
  int off[8];               // time offsets in Append() order
  const uint16_t* wc_hi;    // compact weight planes [n_pad][kFirstK] (launch_compact_first)
  const uint16_t* wc_lo;
};
// epi_prec selects what the planes epilogue writes: kPrecFp16x3 / kPrecBf16x3 (hi + lo planes), kPrecFp16x3E (fp16 plane +
// 4-bit residual), kPrecFp16x2 (fp16 plane only); the products are three-pass in every case
hipError_t launch_tdnn_first(const FirstArgs& a, int epi_prec, hipStream_t s);
// [n_pad][ldw] planes of the generic K walk (segment j at column j * seg_pad) -> compact planes [n_pad][kFirstK]
hipError_t launch_compact_first(const uint16_t* src, int ldw, int seg_pad, int n_pad, int noff, int dim, uint16_t* dst, hipStream_t s);

// fp32 packed features [src rows][dim] -> 16-bit planes [dev rows][ld] (zero padded columns,
// zero rows for alignment padding).  grp_utt[g] = utterance of 16-row group g or -1.
struct PrepArgs {
  const float* feats;       // packed rows as handed over by the caller
  const int32_t* src_off;   // [B+1] row offsets into feats
  const int32_t* dev_off;   // [B]   first device row of each utterance (multiple of 16)
  const int32_t* grp_utt;   // [rows/16]
  int rows;                 // device rows (multiple of kBM)
  int dim;                  // feature dimension (e.g. 23)
  int ld;                   // plane leading dimension (multiple of kBK)
  uint16_t* out_hi;
  uint16_t* out_lo;
  // frame-level outputs (nnet3-compute semantics): the chunk is extended by pad_left copies of its first frame and
  // pad_right copies of its last one, so that every input frame gets an output row (0/0 for x-vector extraction)
  int pad_left, pad_right;
  unsigned* zero_words;     // cleared by the same launch (the group-max tables of the pass), or null
  int n_zero_words;
};
hipError_t launch_prep_input(const PrepArgs& a, int precision, hipStream_t s);

// Statistics pooling finalise (mean, stddev with variance floor), SURVEY.md B.7:
//   mu = S/n ; sigma = sqrt(max(Q/n - mu^2, floor)) ; out row b = [mu | sigma] as planes.
struct PoolArgs {
  const float* partial;     // [rows/16][2][ldp]
  int ldp;
  const int32_t* utt_grp0;  // [B] first 16-row group of the utterance
  const int32_t* utt_grp1;  // [B] one past its last group
  const int32_t* utt_count; // [B] number of pooled frames
  int B;
  int dim;                  // pooled layer dimension (e.g. 1500)
  float var_floor;
  uint16_t* out_hi;         // [B_pad][ld]
  uint16_t* out_lo;
  int ld;
};
hipError_t launch_pool_finalise(const PoolArgs& a, int precision, hipStream_t s);

// Frame-level output: gathers the rows that correspond to input frames out of the [rows][ld] fp32 result of the last
// GEMM into the caller's packed matrix, optionally applying LogSoftmaxComponent row-wise (rows of up to 16384 columns are
// kept in registers: one read, one write).
struct FrameOutArgs {
  const float* src;         // [device rows][ld]
  const uint16_t* src16;    // or the same as fp16 (log_softmax only; the single-pass fp16 mode keeps its logits in 16 bits)
  int ld;
  const int32_t* out_row;   // [n_out] device row of every output row
  int n_out;
  int dim;                  // output dimension (unpadded)
  int log_softmax;
  float* out;               // [n_out][out_ld]
  int out_ld;
};
hipError_t launch_frame_output(const FrameOutArgs& a, hipStream_t s);

// Feature front-end on the device: sliding-window cepstral mean subtraction
// (apply-cmvn-sliding --norm-vars=false, centred or not) followed by the selection of voiced frames
// (select-voiced-frames), the pipeline of egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:79.
struct FrontEndArgs {
  const float* raw;         // packed raw feature rows [raw rows][dim]
  const int32_t* raw_off;   // [n_utts + 1] row offsets of the utterances
  double* prefix;           // [raw rows + n_utts][dim] per-utterance exclusive prefix sums (workspace)
  int n_utts;
  int dim;
  const int32_t* sel_row;   // [n_out] absolute raw row of every kept frame
  const int32_t* sel_utt;   // [n_out] its utterance
  int n_out;
  int cmn_window;           // <= 0: no mean subtraction
  int center;
  int min_window;
  float* out;               // [n_out][dim]
};
hipError_t launch_frontend(const FrontEndArgs& a, hipStream_t s);

// Kaldi's compressed matrices ("CM": the storage format of the recipes' raw features, make_mfcc.sh --compress true) expanded on
// the device, in front of the front-end above: the host uploads one byte per element instead of expanding four on a core first
// (4.5 ns per element in round 5, 1 ns with a table per column - still more than everything else a reader thread does).
// Object of utterance u, at cm + cm_off[u] (any alignment): {float min_value, range; int32 rows, cols; uint16 percentiles[cols][4];
// uint8 data[cols][rows]} - what follows the "CM " token in the archive.  Element (r, c) = the piecewise-linear value of byte
// data[c][r] between the column's 0 / 25 / 75 / 100 % points (compressed-matrix.h: CharToFloat), written to
// out[(raw_off[u] + r) * dim + c] with exactly the host reader's float operations (no fused multiply-add): the same bits.
struct CmExpandArgs {
  const uint8_t* cm;
  const int64_t* cm_off;    // [n_utts]
  const int32_t* raw_off;   // [n_utts + 1] rows of every utterance in out
  int n_utts, dim, max_rows;
  float* out;               // [raw_off[n_utts]][dim]
};
hipError_t launch_cm_expand(const CmExpandArgs& a, hipStream_t s);

// Speaker-level back-end on the device (the chain of egs/sre/v2/run_sre10.sh:238-241:
// ivector-subtract-global-mean | transform-vec | ivector-normalize-length):
//   y = T (x - mean)   [T linear (cols == dim) or affine (cols == dim + 1), each stage optional]
//   y *= sqrt(out_dim) / |y|   (normalize; scaleup = 0: y /= |y|; a zero vector is left alone)
struct BackendArgs {
  const float* x;        // [n][ldx]
  int n, dim, ldx;
  const float* mean;     // [dim] or null
  const float* t;        // [t_rows][t_cols] row-major or null
  int t_rows, t_cols;
  int normalize, scaleup;
  float* out;            // [n][ldo], out_dim = t ? t_rows : dim
  int ldo;
  float* ratio;          // [n] length ratio |y| / sqrt(out_dim) (or |y|) before normalisation, or null
};
hipError_t launch_backend(const BackendArgs& a, hipStream_t s);

// Segment means (ivector-mean, extract_xvectors_new.sh:106-107): out[s] = mean of the rows idx[seg_off[s] ..
// seg_off[s+1]) of x, accumulated in list order in fp32 (acc64 = 0) or fp64 (acc64 = 1).  Empty segments give zeros.
struct SegMeanArgs {
  const float* x;        // [n][ldx]
  int dim, ldx;
  const int32_t* seg_off;  // [n_seg + 1]
  const int32_t* idx;      // [seg_off[n_seg]] row numbers
  int n_seg;
  int acc64;
  float* out;            // [n_seg][dim]
};
hipError_t launch_segment_mean(const SegMeanArgs& a, hipStream_t s);

// 16-bit helpers shared by host packing code (round-to-nearest-even, like v_cvt_pk_bf16_f32).
uint16_t host_f32_to_bf16(float x);
float host_bf16_to_f32(uint16_t h);
uint16_t host_f32_to_f16(float x);
float host_f16_to_f32(uint16_t h);

}  // namespace xv
