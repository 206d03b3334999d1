// Device engine: packed weights resident in HBM + per-batch plans + the kernel sequence.
//
// Replaces NnetComputer::Run() / RunNnetComputation() of nnet3-xvector-compute (SURVEY.md §3.1 HOT
// LOOP 3) for a whole batch of chunks at once.  A "plan" is the analogue of the cached compiled
// computation Kaldi keeps per distinct chunk length (CachingOptimizingCompiler, §8(a) row a6): it
// holds the row geometry of one batch shape.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "kernels.h"
#include "program.h"

namespace xv {

struct EngineError : public std::runtime_error {
  explicit EngineError(const std::string& m) : std::runtime_error(m) {}
};

// Flat, position-independent image of a lowered + padded + precision-split model.  This is what one
// rank broadcasts to the others over RCCL and what an Engine is built from.
std::vector<uint8_t> PackModel(const TdnnProgram& prog, int precision);
// The same with precision kPrecDefault (-1, XV_PREC_DEFAULT) resolved by the one policy every entry point shares (C ABI,
// command-line tools, Python host, multi-GPU launcher): kPrecFp16Mx2 for a pooled output whose layers can all run it,
// kPrecFp16x3 where PackModel rejects that mode for the model and for frame-level outputs.  *resolved = the mode packed.
constexpr int kPrecDefault = -1;
std::vector<uint8_t> PackModelPolicy(const TdnnProgram& prog, int precision, int* resolved = nullptr);
const char* PrecisionName(int precision);   // "fp16mx2", ...; "?" for an unknown value

// kPrecFp16Mx: e2m1 image of one row of weight residuals res[k_pad] (K-contiguous, padded like the fp16 planes) in the
// order the kernels walk the K steps (step_wcol from PlanWalkSteps, kernels.h), with one E8M0 scale per block of four
// steps and lane group g (the 32 values columns 8 g .. 8 g + 7 of each of the four steps - what one lane group of a
// 16x16x128 operand holds, the granularity of the instruction's scale operand): scales[k_pad / 32], index 4 * block + g.
void PackMxRow(const float* res, int k_pad, const int* step_wcol, uint8_t* row, uint8_t* scales);
// kPrecFp16Mx2: e2m1 image of one row of the weights themselves, w[k_pad], for the second K walk: 64 bytes per 128-column
// step (lo_wcol from PlanWalkLoSteps), four lane-group chunks of 32 consecutive columns, one E8M0 scale per chunk
// (scales[4 * n_lo], index 4 * step + g).
void PackMxWeightsRow(const float* w, const int* lo_wcol, int n_lo, uint8_t* row, uint8_t* scales);
// The scales of all rows, natural[n_pad][nsteps], in the order the kernels stage them: 512 bytes per (128-row tile, block
// of four steps) = [64-row half][fragment row i][lane group g][fragment w of the half]: lane (i, g) of a wave reads the
// four scale bytes of its four weight fragments of a half as one dword.  weights_are_operand_a: the planes / f32
// epilogues (LDS rows of a weight tile permuted, kernels.hip swap_fields); false: the statistics epilogue.
void TileMxScales(const uint8_t* natural, int n_pad, int nsteps, bool weights_are_operand_a, uint8_t* tiled);

struct BlobLayerInfo {
  std::string name;
  int in_dim, out_dim, k_pad, n_pad, relu, bn, log_softmax, segment_level, left, right;
  bool has_w4 = false;   // carries the 4-bit residual plane of kPrecFp16Mx
  bool has_w4b = false;  // and the 4-bit weight image of kPrecFp16Mx2
  bool has_w4p = false;  // and the residual plane in the walk order of tdnn_gemm_kernel_p8
  std::vector<LayerSource> src;
};

struct BlobInfo {
  int precision, input_dim, pooled_layer, pool_dim, pool_left, pool_right, output_layer, output_dim,
      output_is_segment, left_context, right_context, min_frames;
  float variance_floor;
  uint64_t fingerprint = 0;   // of the packed image (PackModel); what a shared calibration file names
  std::vector<BlobLayerInfo> layers;
  double Macs(int T) const;
};
BlobInfo ParseBlobInfo(const uint8_t* blob, size_t n);
// Header + layer table (the first data_offset bytes) of a packed image that lives in device memory.
std::vector<uint8_t> ReadBlobHead(const void* device_blob, size_t n);

class Engine {
 public:
  struct Plan;
  // Throws EngineError when no usable HIP device exists - there is no CPU fallback in the product.
  // device_image (optional): the same n bytes already resident on `device` (e.g. received by RCCL broadcast); the
  // weights are then copied device to device and `blob` only has to hold the header and the layer table.
  Engine(const uint8_t* blob, size_t n, int device, const void* device_image = nullptr);
  ~Engine();
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;

  const BlobInfo& info() const { return info_; }
  int device() const { return device_; }

  // row_offsets: B+1 offsets (in rows) of the chunks inside the packed feature matrix.
  std::shared_ptr<Plan> MakePlan(const int32_t* row_offsets, int B);
  // feats_dev: packed fp32 rows [row_offsets[B]][input_dim] in device memory;
  // out_dev: [B][out_ld] fp32 (device).  Asynchronous on `stream` (nullptr = the engine's stream).
  void Forward(const Plan& plan, const float* feats_dev, float* out_dev, int out_ld, hipStream_t stream);
  // Host convenience: H2D, forward, D2H, synchronise.  Segment-level output: out is [B][output_dim]; frame-level
  // output: out is [sum of chunk lengths][output_dim] (one row per input frame), chunk b starting at OutRowOffset.
  void ForwardHost(const float* feats, const int32_t* row_offsets, int B, float* out);
  // Asynchronous host path (table jobs): kNumHostSlots batches queued or in flight, each with its own pinned staging
  // buffers, device staging and plan tables; batch number `seq` runs on the stream of lane (seq % lanes), so
  // consecutive batches alternate lanes while a third one is already queued behind them.  Usage per slot:
  //   float* f = HostFeats(slot, rows);          // pinned buffer to pack the chunks of the batch into
  //   SubmitHost(slot, seq, row_offsets, B);     // H2D, forward, D2H - all asynchronous, returns at once
  //   const float* out = WaitHost(slot);         // blocks until the batch is done; [B][output_dim] (pinned)
  // A slot must be waited for before it is reused.  row_offsets[0] must be 0.
  static constexpr int kNumHostSlots = 3;
  float* HostFeats(int slot, size_t rows);
  // With a FrontEndJob the pinned buffer holds RAW feature rows; sliding CMN + frame selection run on the device in
  // front of the network (same stream) and their output never visits the host.  row_offsets then describe the
  // selected rows (row_offsets[B] == fe->n_out).
  struct FrontEndJob {
    const int32_t* raw_off;   // [n_utts + 1] rows of every utterance inside the raw buffer
    int n_utts;
    const int32_t* sel_row;   // [n_out] absolute raw row of every kept frame, ascending
    const int32_t* sel_utt;   // [n_out] its utterance
    int n_out;
    int cmn_window;
    bool center;
    int min_window;
    // The pinned buffer holds COMPRESSED objects instead of float rows (Kaldi "CM" matrices, kernels.h CmExpandArgs): cm_off[u]
    // = byte offset of utterance u's object in it, cm_bytes = all of them, max_rows = the longest utterance.  Null: float rows.
    const int64_t* cm_off = nullptr;
    size_t cm_bytes = 0;
    int max_rows = 0;
  };
  void SubmitHost(int slot, long seq, const int32_t* row_offsets, int B, const FrontEndJob* fe = nullptr);
  const float* WaitHost(int slot);
  bool frame_mode() const { return frame_mode_; }
  // Feature front-end on the device: sliding-window CMN (cmn_window <= 0: none) + selection of the rows listed in
  // sel_row (absolute raw rows, ascending; sel_utt = their utterance).  Host buffers in / out, blocking.
  void FrontEndHost(const float* raw, const int32_t* raw_off, int n_utts, const int32_t* sel_row, const int32_t* sel_utt,
                    int n_out, int cmn_window, bool center, int min_window, float* out);

  // ---- calibrated arithmetic (contexts packed as kPrecFp16Mx2, pooled output) ----------------------------------------
  // The packed image of kPrecFp16Mx2 holds everything the lighter kPrecFp16Mx arithmetic needs (and the three-pass
  // one), so one context can run any of the three on its fast chunks:
  //   kPrecFp16Mx2  what the image was packed for: 1.5 MFMA passes, error independent of the model (the state after creation)
  //   kPrecFp16Mx   1.25 passes, no activation residual planes; meets the bar on some models only - chunks that pool >= 300 frames
  //   kPrecFp16x3   every chunk in the three-pass arithmetic
  // Calibrate() measures, on the caller's own chunks, the embeddings of the two fast modes against the three-pass ones and
  // switches the context to kPrecFp16Mx when its worst relative error stays within `tol`, else keeps kPrecFp16Mx2 (or drops
  // to kPrecFp16x3 when even that exceeds 1e-4).  Each mode is measured on the chunks it runs fast (kPrecFp16Mx: >= 300 pooled
  // frames, kPrecFp16Mx2: >= 160); kPrecFp16Mx needs at least kCalibMinChunks such chunks to be chosen.  Synchronous; results of later calls are bit-reproducible for a given choice.
  struct Calibration {
    int chosen = 0;           // kPrecFp16Mx / kPrecFp16Mx2 / kPrecFp16x3 (or the context's precision when it cannot switch)
    int checked = 0;          // chunks that entered the comparison
    float err_mx = 0.f;       // worst max|d| / max|ref| of kPrecFp16Mx over them
    float err_mx2 = 0.f;      // the same for kPrecFp16Mx2
    int checked_mx = 0;       // chunks among `checked` that kPrecFp16Mx runs fast: err_mx is over these, and fewer than
                              // kCalibMinChunks of them never select kPrecFp16Mx
    uint64_t lite_mask = 0;   // chosen == kPrecFp16Mx2 only: the layers (bit = layer index) that run the 1.25-pass arithmetic
                              // inside the 1.5-pass context (SetLiteMask) - the most expensive ones the tolerance allows
    float err_lite = 0.f;     // worst error of that mixture over ALL `checked_mx` chunks, both halves (0 when lite_mask == 0)
    // The mixture is SELECTED on one half of the sample (even positions of the picked chunks) and CONFIRMED on the other
    // (odd positions), which took no part in the selection: layers are dropped, last added first, until the held-out half
    // is within the tolerance too.
    int checked_holdout = 0;  // chunks of the held-out half that kPrecFp16Mx would run fast
    float err_holdout = 0.f;  // worst error of the adopted mixture over them (0 when lite_mask == 0)
    int lite_dropped = 0;     // layers the selection half had admitted and the held-out half threw out again
    // Projected tail of the per-chunk error: mean + kTailSigmas standard deviations over the chunks that confirmed a configuration
    // (never below the worst of them).  A configuration is accepted only while this is within tol x kTailOverTol (Calibrate).
    float tail = 0.f;         // of what was adopted: fp16mx / plain fp16mx2 on the whole sample, a mixture on its held-out half
    float tail_mx = 0.f;      // of fp16mx on the chunks it runs fast, adopted or not
  };
  static constexpr int kCalibMinChunks = 16;
  static constexpr double kTailSigmas = 6.0;
  static constexpr float kTailOverTol = 1.10f;        // fp16mx outright, mixtures
  static constexpr float kTailOverTolPacked = 1.20f;  // plain fp16mx2 (else the three-pass arithmetic)
  static constexpr int kMaxLanes = 4;       // XVEC_LANES is clamped to this
  static constexpr int kMaxLiteLayers = 64; // lite_mask is a uint64: only layers with index < 64 can be "lite" (SetLiteMask drops the rest)
  bool can_switch_fast_mode() const { return info_.precision == kPrecFp16Mx2 && !frame_mode_; }
  int fast_mode() const { return can_switch_fast_mode() ? fast_mode_ : info_.precision; }
  void SetFastMode(int mode);   // throws unless can_switch_fast_mode() and mode is one of the three; clears the lite mask
  // Mixed arithmetic inside the kPrecFp16Mx2 mode: the frame-level layers named by `mask` (bit = layer index; bits of layers
  // that cannot run it are ignored) compute their products in 1.25 passes (kPrecFp16Mx: no walk over the residual plane of
  // their inputs); a lite layer still writes the residual plane of its own output when a consumer walks it (kPrecFp16MxE).
  // Fast chunks are then those of the kPrecFp16Mx threshold.  Throws unless the current fast mode is kPrecFp16Mx2.
  void SetLiteMask(uint64_t mask);
  uint64_t lite_mask() const { return lite_mask_; }
  Calibration Calibrate(const float* feats, const int32_t* row_offsets, int B, float tol);

  hipStream_t stream() const { return stream_; }
  int num_lanes() const { return (int)lanes_.size(); }
  // last forward's per-kernel launch list (name, m_tiles*n_tiles) for logging / tests
  size_t weight_bytes() const { return blob_data_bytes_; }
  // Per-launch timing with HIP events stamped by the kernel dispatches themselves (hipExtLaunchKernelGGL), on the
  // stream the kernels are launched on.
  void SetProfiling(bool on) { prof_on_ = on; }
  // "label<TAB>launches<TAB>total_ms" lines for everything recorded since the last report; resets.
  std::string ProfileReport();

 private:
  struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
  };
  struct DevLayer {
    const uint16_t* w_hi;
    const uint16_t* w_lo;
    const float* bias;
    const float* scale;
    const float* offset;
    const uint8_t* w4;
    const uint8_t* w4_scale;
    int ldw4;
    const uint8_t* w4b;         // kPrecFp16Mx2
    const uint8_t* w4b_scale;
    int ldw4b;
    const uint8_t* w4p = nullptr;        // residual plane + scales in the K-walk order of tdnn_gemm_kernel_p8 (row pitch ldw4)
    const uint8_t* w4p_scale = nullptr;
    const uint8_t* w4bp = nullptr;       // kPrecFp16Mx2 on that kernel: 4-bit weight image of its second walk (rows of k_pad * 2 bytes) + scales
    const uint8_t* w4bp_scale = nullptr;
    // layers that read only the network input and run tdnn_first_kernel: compact weight planes [n_pad][kFirstK], built on
    // the device from the packed image at construction (owned: first_buf)
    bool first = false;
    const uint16_t* wc_hi = nullptr;
    const uint16_t* wc_lo = nullptr;
  };
  struct ActBuf {
    Buf act_hi, act_lo;   // frame-level: [halo + rows + halo][n_pad]; segment-level: [b_pad][n_pad]
    Buf act_lo4, act_lo4s;   // kPrecFp16Mx2: 4-bit residual of act_hi [..][n_pad / 2] and its scales [..][Lo4ScalePitch(n_pad)]
  };
  // A lane = one in-flight batch: its own stream and its own activation / workspace buffers.  Consecutive
  // forward calls on the engine's own streams alternate between lanes, so the tail of one batch (partially filled
  // last round of tiles, the 8-workgroup embedding GEMM, small kernels) overlaps with the next batch's kernels.
  struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;   // recorded at the end of every forward on this lane
    bool busy = false;
    std::vector<ActBuf> act;     // per layer
    Buf in_hi, in_lo, partial, stats_hi, stats_lo, out_f32, splitk_ws, frame_f32;
    Buf gmax;              // [layer][gmax_stride] group maxima of the activation planes (kPrecFp16Mx)
    int gmax_stride = 0;
    int cap_rows = 0, cap_b = 0;
  };
  struct HostSlot {
    void* h_feats = nullptr;   // pinned
    size_t h_feats_bytes = 0;
    void* h_out = nullptr;     // pinned
    size_t h_out_bytes = 0;
    void* h_tables = nullptr;  // pinned copy of the plan tables
    size_t h_tables_bytes = 0;
    Buf d_feats, d_out, d_tables;
    Buf d_raw, d_prefix, d_fetab;   // front-end: raw rows, per-utterance prefix sums (double), tables
    Buf d_cm;                       // front-end on compressed input: the uploaded objects
    void* h_fetab = nullptr;        // pinned
    size_t h_fetab_bytes = 0;
    hipEvent_t done = nullptr;
    hipEvent_t h2d_done = nullptr;   // uploads run on copy_stream_, ahead of the lane that will consume them
    bool pending = false;
    int lane = 0;                 // the lane (stream) the batch was submitted on: whose stream-K fault word concerns this slot
    std::unique_ptr<Plan> plan;   // its tables live in d_tables (not owned by the plan)
  };
  HostSlot host_slots_[kNumHostSlots];
  void EnsurePinned(void** p, size_t* have, size_t bytes);
  // host part of a plan: fills everything but the device pointers and returns the table image
  void FillPlan(const int32_t* row_offsets, int B, Plan* plan, std::vector<uint8_t>* tables) const;
  static void BindPlan(Plan* plan, const void* device_tables);
  void ForwardOnLane(size_t lane, const Plan& plan, const float* feats_dev, float* out_dev, int out_ld, hipStream_t stream);
  void Check(hipError_t e, const char* what) const;
 public:
  // throws when a kernel of this process reported a fault the HIP API cannot see (called after synchronising)
  // lane >= 0: only that lane's launches are the caller's (a fault another lane's batch raised stays recorded for the slot it
  // belongs to); -1: everything this engine launched
  void CheckKernelFaults(int lane = -1) const;
  mutable unsigned fault_mask_ = 0;   // bit i: lane i's stream reported a timed-out stream-K wait; bit 31: the engine's own stream;
                                      // bit 30: a caller's stream (xv_forward_batch_device)
  std::vector<hipStream_t> ext_streams_;   // callers' streams this engine launched on (their fault words are read with the others)
 private:
  void Ensure(Buf* b, size_t bytes, bool zero, hipStream_t consumer = nullptr);   // zero: filled on `consumer`, the stream that will use it
  void EnsureCapacity(Lane& L, int rows, int b_pad, hipStream_t s);
  uint16_t* ActBase(const Buf& b, int ld) const;

  BlobInfo info_;
  int device_ = 0;
  bool frame_mode_ = false;   // output is one row per input frame (nnet3-compute semantics), no pooling
  int pad_left_ = 0, pad_right_ = 0;
  int nplanes_ = 1;
  // precision policy (engine.cc, constructor): kernel mode of everything but the frame-level GEMMs of "fast" chunks,
  // whether fast chunks exist at all, and the number of pooled frames that makes a chunk fast
  static constexpr int kDefaultFastMinPooled = 300;
  int slow_prec_ = 0;
  bool has_fast_ = false;
  bool fast_mx_ = false;
  int p8_whole_ = 0;   // XVEC_DEBUG=p8_whole: partition policy of tdnn_gemm_kernel_p8 (engine.cc)
  bool use_p8_ = true;      // tdnn_gemm_kernel_p8 for the layers and modes it can run (XVEC_DEBUG=p8=0: never)
  bool fast_mx2_ = false;   // kPrecFp16Mx2: every frame-level layer of a fast chunk runs it (or kPrecFp16x3E on the input)
  // Frame-level log-posteriors in the single-pass fp16 mode: the head's logits stay a 16-bit plane like every other layer's
  // output of that mode (2 instead of 4 bytes per logit written by the head GEMM and read by the LogSoftmax pass: the two
  // were store- and bandwidth-bound on 1.6 GB each way); the log-posteriors themselves are fp32.  Every other mode keeps
  // fp32 logits.
  bool logits16() const { return frame_mode_ && info_.precision == kPrecFp16 && info_.layers[info_.output_layer].log_softmax; }
  int fast_min_pooled_ = 0;
  int fast_mode_ = 0;       // see SetFastMode
  uint64_t lite_mask_ = 0;  // see SetLiteMask
  std::vector<char> lite_, lite_emits_;   // per layer: runs the 1.25-pass arithmetic in the kPrecFp16Mx2 mode / and still emits its residual plane
  int mx2_min_pooled_ = 0, mx_min_pooled_ = 0;   // thresholds of the two fast modes (XVEC_FAST_MIN_POOLED overrides both)
  hipStream_t stream_ = nullptr;
  hipStream_t copy_stream_ = nullptr;   // host-slot uploads (SubmitHost)
  void* d_blob_ = nullptr;
  size_t blob_data_bytes_ = 0;
  std::vector<DevLayer> layers_;
  std::vector<Buf> first_buf_;     // compact first-layer weight planes (two per such layer)
  bool need_prep_ = true;          // some layer still reads the input planes prep_input writes
  int in_ld_ = 0;
  int stats_ld_ = 0;
  std::vector<Lane> lanes_;
  size_t next_lane_ = 0;
  Buf feats_stage_, out_stage_;
  Buf fe_raw_, fe_tab_, fe_prefix_, fe_out_;
  std::map<std::vector<int32_t>, std::shared_ptr<Plan>> plan_cache_;
  bool prof_on_ = false;
  std::vector<std::string> prof_labels_;
  std::vector<std::vector<hipEvent_t>> prof_runs_;
};

struct Engine::Plan {
  int B = 0, b_pad = 0, rows = 0, src_rows = 0;
  int rows_fast = 0;             // device rows [0, rows_fast) hold the chunks that run the two-pass kernels
  std::vector<int32_t> src_off;  // [B+1]
  void* d_tables = nullptr;      // one device allocation holding all tables below (owned unless borrowed)
  bool borrowed_tables = false;  // tables live in a host slot's buffer
  size_t o_src = 0, o_dev = 0, o_gu = 0, o_gr = 0, o_g0 = 0, o_g1 = 0, o_cn = 0, o_or = 0, o_ar = 0, o_gs = 0;  // table offsets
  int ngrp = 0;                  // 16-row groups of the batch
  const int8_t* d_act_range = nullptr;   // [layer][ngrp][2] computable rows of each group, per layer (kPrecFp16Mx)
  const int32_t* d_src_off = nullptr;
  const int32_t* d_dev_off = nullptr;
  const int32_t* d_grp_utt = nullptr;
  const void* d_grp_src = nullptr;       // [ngrp] int4: where the frames of each 16-row group come from (kernels.h, FirstArgs)
  const int8_t* d_grp_range = nullptr;
  const int32_t* d_utt_grp0 = nullptr;
  const int32_t* d_utt_grp1 = nullptr;
  const int32_t* d_utt_count = nullptr;
  // frame-level output mode: output row -> device row, and where each chunk's rows start in the packed output
  const int32_t* d_out_row = nullptr;
  std::vector<int32_t> out_off;  // [B+1]
  int n_out = 0;
  ~Plan();
};

}  // namespace xv
