// The per-utterance loop of nnet3-xvector-compute: chunking, edge padding of short chunks, batched
// forward passes and the length-weighted average (the parameters come
// from egs/sre/v2/local/nnet3/xvector/run_xvector_new.sh:83,88 and extract_xvectors_new.sh:62-68).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "engine.h"

namespace xv {

struct Chunk {
  int utt;       // index of the utterance in the call
  int start;     // first source row inside the utterance
  int len;       // frames taken from the utterance (= the averaging weight)
  int left_pad;  // copies of the first row put in front (pad_input)
  int right_pad; // copies of the last row appended
};

// Chunk list of one utterance of `num_rows` frames.  Returns false (with a reason) when Kaldi would
// count the utterance as failed.  `min_net_frames` is the network's own minimum (left+right context+1).
bool PlanChunks(int utt, int num_rows, int chunk_size, int min_chunk_size, bool pad_input, int min_net_frames,
                std::vector<Chunk>* out, std::string* why);

struct ExtractOptions {
  int chunk_size = -1;
  int min_chunk_size = 100;
  bool pad_input = true;
  int max_batch_rows = 1 << 17;   // rows of (padded) chunks per forward pass
  int max_batch_chunks = 4096;
  // optional device front-end in front of the extractor (replaces the apply-cmvn-sliding | select-voiced-frames pipes
  // of extract_xvectors_new.sh:79): sliding CMN when cmn_window > 0, voiced-frame selection when a VAD table is given
  int cmn_window = 0;
  bool cmn_center = true;
  int cmn_min_window = 100;
  std::string vad_rspecifier;
  // optional device back-end behind the extractor (replaces the ivector-subtract-global-mean | transform-vec |
  // ivector-normalize-length pipes of egs/sre/v2/run_sre10.sh:240 for the test-side vectors): applied to every
  // embedding before it is written.  Table jobs only (table_extract.cc).
  std::vector<float> backend_mean;        // empty: no mean subtraction
  std::vector<float> backend_transform;   // [rows][cols] row-major, empty: none
  int backend_t_rows = 0, backend_t_cols = 0;
  bool backend_normalize = false;
  bool backend_scaleup = true;
  // table jobs: before the first batch, Engine::Calibrate on the first chunk of the first calibrate_utts utterances (contexts
  // that can switch their arithmetic: packed as fp16mx2; see engine.h)
  bool calibrate = false;
  float calibrate_tol = 7.5e-5f;   // three quarters of the 1e-4 bar, on the WORST calibration chunk
  int calibrate_utts = 64;
  // the SHARED choice of a recipe (calib_file.h): when the file exists its choice is applied and nothing is measured; when it
  // does not, the job measures (as with calibrate), publishes atomically and adopts what the file then holds.  Empty: none.
  std::string calibration_file;
};

// feats: packed host rows; utterance u = rows row_offsets[u] .. row_offsets[u+1]-1.
// out[u*output_dim ..] receives the embedding when ok[u] == 1; why[u] (optional) gets the failure reason.
void ExtractUtterances(Engine* eng, const ExtractOptions& opt, const float* feats, const int32_t* row_offsets, int n_utts,
                       float* out, int32_t* ok, std::vector<std::string>* why);

// The same computation split in two so that a table job can keep two batches in flight (the packing / writing of one
// overlaps the device work of the other).  Start() plans the chunks and, when they fit one forward batch, packs them
// into the engine's pinned buffer of `slot` and submits them asynchronously; Finish() waits, averages and reports.
// Utterance sets that need several forward batches (long recordings cut into chunks) are processed synchronously
// inside Finish().  feats / row_offsets must stay valid until Finish() returns.
class ExtractJob {
 public:
  // seq = running number of the batch (selects the engine lane)
  void Start(Engine* eng, const ExtractOptions& opt, int slot, long seq, const float* feats, const int32_t* row_offsets,
             int n_utts);
  // The same without a packed copy of the features: utterance u = utt[u][0 .. rows[u] * input_dim), copied straight
  // into the engine's pinned staging buffer (table jobs: one host copy per byte instead of two).  The pointers must
  // stay valid until Finish() returns.
  void StartPtrs(Engine* eng, const ExtractOptions& opt, int slot, long seq, const float* const* utt, const int32_t* rows,
                 int n_utts);
  // Table jobs with the device front-end: raw[u] = the utterance's raw rows (raw_rows[u] of them), vad[u] = its VAD
  // decisions (or null: keep every row; every utterance keeps at least one).  The raw rows are staged, CMN + selection + network
  // run on the device without a host round trip, and true is returned - for single-chunk utterances, for utterances cut into
  // several chunks and for short chunks padded by edge replication alike (all of them selections of the kept rows); false
  // (nothing submitted, the caller uses FrontEndHost + Start) only when the batch does not fit one forward batch.
  // cm / cm_bytes (optional): utterances that arrive as COMPRESSED views of a mapped archive (kio.h Matrix::cm) - when every
  // utterance of the batch has one, the objects are staged as they are (one byte per element) and expanded on the device;
  // raw[u] may then be null.  A batch in which only some utterances are compressed returns false like any other batch the
  // device path cannot take (the caller expands on the host).
  bool StartFrontEnd(Engine* eng, const ExtractOptions& opt, int slot, long seq, int n_utts, const float* const* raw,
                     const int32_t* raw_rows, const float* const* vad, const uint8_t* const* cm = nullptr,
                     const size_t* cm_bytes = nullptr);
  void Finish(float* out, int32_t* ok, std::vector<std::string>* why);
  bool active() const { return eng_ != nullptr; }

 private:
  Engine* eng_ = nullptr;
  ExtractOptions opt_;
  int slot_ = 0, n_utts_ = 0;
  const float* feats_ = nullptr;
  const int32_t* row_offsets_ = nullptr;
  bool async_ = false;
  std::vector<const float*> utt_ptr_;   // StartPtrs: per-utterance rows (packed on demand for the synchronous fallback)
  std::vector<int32_t> utt_rows_;
  std::vector<float> fallback_pack_;
  std::vector<int32_t> fallback_offs_;
  std::vector<Chunk> chunks_;
  std::vector<int32_t> ok_;
  std::vector<std::string> why_;
};

}  // namespace xv
