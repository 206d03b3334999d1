// The shared choice of arithmetic of a recipe: a small text file beside the model.
//
// Why it exists.  The reference runs `nj` independent processes on split lists and concatenates their outputs
// (egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:72,79,91-99); Kaldi's fp32 gives an utterance the same vector whatever
// shard it lands in.  The lighter arithmetics of this library (fp16mx, a per-layer mixture inside fp16mx2) are admitted by a
// MEASUREMENT on sample chunks (Engine::Calibrate) - a choice that depends on the sample.  Were every process to measure its own
// shard, two shards could run different arithmetics and an utterance's embedding would depend on `nj` ("missing" 1).
// So the command-line tools never choose from their own data by default: `--precision=default` is plain fp16mx2, a function of the
// model alone.  A measured choice is used only when it is SHARED: `--calibration=<file>` (or $XVEC_CALIBRATION, which the
// unchanged wrapper scripts pass through their environment):
//   * the file exists: its choice is applied (the model fingerprint in it must match the packed image, else the job ends with
//     an error - a stale file is never silently ignored or silently used);
//   * it does not: the job measures on its own sample, PUBLISHES the outcome atomically (temporary file + link(2): the first
//     publisher wins, also over NFS), then reads the file back and applies what it holds - its own choice or the winner's.
// Every job of a recipe therefore computes in the arithmetic the file names, whichever of them wrote it and however the lists
// were split; `run.pl JOB=1:nj` jobs that start together all measure, one publishes, all adopt that one.
#pragma once
#include <stdint.h>

#include <string>

namespace xv {

struct SharedChoice {
  uint64_t model = 0;      // BlobInfo::fingerprint of the image the choice was measured on
  int precision = -1;      // kPrecFp16Mx / kPrecFp16Mx2 / kPrecFp16x3
  uint64_t lite_mask = 0;  // kPrecFp16Mx2: the layers in 1.25 passes
  float tolerance = 0.f;   // what it was measured against (information)
  std::string note;        // one line of provenance: sample size, worst error, projected tail (information)
};

// false: the file does not exist.  Throws KioError when it exists but cannot be read or parsed.
bool ReadCalibrationFile(const std::string& path, SharedChoice* out);
// Publishes `mine` unless the file already exists, then reads back what the file holds into *adopted (the winner's choice).
// Returns true when `mine` is what was published.  Throws KioError when the directory cannot be written - a job that cannot
// share its choice must not run on it.
bool PublishCalibrationFile(const std::string& path, const SharedChoice& mine, SharedChoice* adopted);

// Applies a choice read from `path` to an engine: the fingerprint must be that of the engine's image and the lite-mask must name
// layers the model can run in 1.25 passes, else KioError (a stale or foreign file is an error, never silently ignored or used).
class Engine;
void AdoptSharedChoice(Engine* engine, const SharedChoice& sc, const std::string& path);

}  // namespace xv
