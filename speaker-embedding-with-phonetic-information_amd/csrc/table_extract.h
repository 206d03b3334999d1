// The utterance loop of nnet3-xvector-compute over Kaldi tables: feature rspecifier in, vector wspecifier out
// (SURVEY.md §3.1 HOT LOOP 1; call sites egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:86-93).
// A reader thread parses the feature table into batches while the GPU works on the previous batch; failures
// are per utterance (warn + count), never fatal for the shard.
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "engine.h"
#include "extractor.h"

namespace xv {

struct TableExtractResult {
  long num_success = 0;
  long num_fail = 0;
  double frames = 0;        // frames of the utterances written
  double seconds = 0;       // wall time of the loop
  int reader_status = 0;    // exit status of the feature input pipe (0 if not a pipe)
};

// level: "LOG" or "WARNING"
typedef std::function<void(const char* level, const std::string& msg)> LogFn;

TableExtractResult RunTableExtraction(Engine* engine, const ExtractOptions& opt, const std::string& feature_rspecifier,
                                      const std::string& vector_wspecifier, const LogFn& log);
// One process, several GPUs (nnet3-xvector-compute --devices=...; the `--nj 1` form of extract_xvectors_new.sh:83-93): whole
// batches dealt round-robin to engines that hold the same model, finalised in submission order through one writer - the
// output is byte-identical to a one-GPU run.  The arithmetic is calibrated once on engines[0] and applied to the others.
TableExtractResult RunTableExtraction(const std::vector<Engine*>& engines, const ExtractOptions& opt,
                                      const std::string& feature_rspecifier, const std::string& vector_wspecifier, const LogFn& log);

// Engine::Calibrate on the first chunk of up to opt.calibrate_utts utterances: (key, rows, row-major data) triples as the
// readers deliver them.  The device front-end of opt (sliding CMN, VAD selection) is applied first, like the job will.
// Utterances the job would skip (wrong dimension, no VAD entry) are skipped here.  Logs one line with the outcome.
struct CalibUtt {
  const std::string* key;
  const float* data;
  int rows, cols;
};
Engine::Calibration CalibrateOnUtterances(Engine* engine, const ExtractOptions& opt, const std::vector<CalibUtt>& utts,
                                          const LogFn& log);
// The same on opt.calibrate_utts utterances of a feature table: spread evenly over the WHOLE list where the table's objects
// can be addressed (archive file, script file: the reference's lists are speaker-sorted, utils/data/split_data.sh:18-21, so a
// head of the list is one or two speakers), the head of the stream otherwise.  Multi-GPU jobs: one rank calibrates on the
// whole list and the choice is applied on every rank, so that an N-way sharded job computes what the 1-way job does.
// `index` (optional): receives the index of an addressable table that the sample was drawn from - every entry, those with an
// error included, and the message of whatever stopped the indexing early - so that the extraction's batching pass can reuse it
// instead of visiting every header a second time.
struct TableIndex {
  bool valid = false;
  std::vector<MatrixTableIndexer::Entry> entries;
  std::string error;
};
Engine::Calibration CalibrateOnTable(Engine* engine, const ExtractOptions& opt, const std::string& feature_rspecifier,
                                     const LogFn& log, TableIndex* index = nullptr);

// nnet3-compute style job: one output MATRIX per utterance (a row per input frame) from a frame-level model
// (reference call sites: sid/nnet3_cvector/cvector/extract_log_post.sh:77-84, sid/nnet3_cvector/am/extract_bn.sh:68,
// steps/nnet3/make_bottleneck_features_new.sh:109).  apply_exp turns log-posteriors into posteriors (--apply-exp).
TableExtractResult RunTableCompute(Engine* engine, int max_batch_rows, bool apply_exp, const std::string& feature_rspecifier,
                                   const std::string& matrix_wspecifier, const LogFn& log);

}  // namespace xv
