// ivector-mean / ivector-subtract-global-mean / transform-vec / ivector-normalize-length - drop-in command lines for
// the vector tools the reference runs on the extracted embeddings (one executable, dispatching on its name):
//   ivector-mean <spk2utt-rspecifier> <ivector-rspecifier> <ivector-wspecifier> [<num-utt-wspecifier>]
//   ivector-mean <ivector-rspecifier> <mean-wxfilename>
//       egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:106-107, egs/sre/v2/run_sre10.sh:219-221,238
//   ivector-subtract-global-mean [<mean-rxfilename>] <ivector-rspecifier> <ivector-wspecifier>     run_sre10.sh:229,239,240
//   transform-vec <transform-rxfilename> <vec-rspecifier> <vec-wspecifier>                          run_sre10.sh:233,239,240
//   ivector-normalize-length [--normalize=true --scaleup=true] <ivector-rspecifier> <ivector-wspecifier>
// The arithmetic runs on the HIP device through libxvec_hip.so (backend.h); without a GPU the tools fail (exit 255).
// Semantics are upstream Kaldi's (not vendored in the reference, hence restated): speaker means accumulate in fp32 in
// spk2utt order, the global mean in fp64; log lines and exit codes follow the Kaldi idiom (0 iff something was written).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "backend.h"
#include "kio.h"

namespace {

std::string g_prog = "ivector-mean";

void LogLine(const char* level, int line, const std::string& msg) {
  fprintf(stderr, "%s (%s[xvec-hip-0.1]:main():ivector_tools_main.cc:%d) %s\n", level, g_prog.c_str(), line, msg.c_str());
}
#define XLOG(msg)                       \
  do {                                  \
    std::ostringstream _o;              \
    _o << msg;                          \
    LogLine("LOG", __LINE__, _o.str()); \
  } while (0)
#define XWARN(msg)                          \
  do {                                      \
    std::ostringstream _o;                  \
    _o << msg;                              \
    LogLine("WARNING", __LINE__, _o.str()); \
  } while (0)

struct Args {
  std::vector<std::string> pos;
  bool binary = true;          // --binary (ivector-mean's mean file)
  bool normalize = true;       // ivector-normalize-length
  bool scaleup = true;
  bool subtract_mean = true;   // ivector-subtract-global-mean
  int device = -1;
};

bool ParseBool(const std::string& v, bool* out) {
  if (v == "true" || v == "t" || v == "1" || v.empty()) *out = true;
  else if (v == "false" || v == "f" || v == "0") *out = false;
  else return false;
  return true;
}

int PickDevice(int requested) {
  if (requested >= 0) return requested;
  const char* e = getenv("XVEC_DEVICE");
  return (e && *e) ? atoi(e) : 0;
}

// All vectors of a table, packed; every vector must have the dimension of the first one.
struct Packed {
  std::vector<std::string> keys;
  std::vector<float> data;
  int dim = 0;
  int n() const { return (int)keys.size(); }
};

// Reads up to `cap` vectors (cap < 0: all).  Returns false when the table is exhausted and nothing was read.
bool ReadBatch(xv::SequentialVectorReader& r, int cap, Packed* p, long* n_err) {
  p->keys.clear();
  p->data.clear();
  std::string key, err;
  std::vector<float> v;
  while ((cap < 0 || p->n() < cap) && r.Next(&key, &v, &err)) {
    if (!err.empty()) {
      XWARN("Failed to read vector for key " << key << ": " << err);
      ++*n_err;
      continue;
    }
    if (p->dim == 0) p->dim = (int)v.size();
    if ((int)v.size() != p->dim || p->dim == 0)
      throw xv::KioError("vector " + key + " has dimension " + std::to_string(v.size()) + ", expected " + std::to_string(p->dim));
    p->keys.push_back(key);
    p->data.insert(p->data.end(), v.begin(), v.end());
  }
  return p->n() > 0;
}

double Norm(const float* v, int n) {
  double s = 0;
  for (int i = 0; i < n; ++i) s += (double)v[i] * v[i];
  return sqrt(s);
}

constexpr int kBatch = 65536;   // vectors per device call of the streaming tools

int IvectorMean(const Args& a) {
  const int dev = PickDevice(a.device);
  if (a.pos.size() == 2) {
    // global mean -> Kaldi vector object
    xv::SequentialVectorReader r(a.pos[0]);
    Packed p;
    long n_err = 0;
    ReadBatch(r, -1, &p, &n_err);
    if (p.n() == 0) {
      fprintf(stderr, "ERROR (%s) No iVectors read\n", g_prog.c_str());
      return 255;
    }
    std::vector<int32_t> off = {0, p.n()}, idx(p.n());
    for (int i = 0; i < p.n(); ++i) idx[i] = i;
    std::vector<float> mean(p.dim);
    xv::SegmentMean(dev, p.data.data(), p.n(), p.dim, off.data(), idx.data(), 1, /*acc64=*/true, mean.data());
    XLOG("Read " << p.n() << " iVectors.");
    xv::WriteVectorObject(a.pos[1], a.binary, mean.data(), p.dim);
    return 0;
  }
  if (a.pos.size() != 3 && a.pos.size() != 4) return -2;
  std::vector<xv::TokenList> spk2utt = xv::ReadTokenVectorTable(a.pos[0]);
  // the vector table is accessed by utterance key, as Kaldi's RandomAccessBaseFloatVectorReader
  xv::SequentialVectorReader r(a.pos[1]);
  Packed p;
  long n_read_err = 0;
  ReadBatch(r, -1, &p, &n_read_err);
  std::unordered_map<std::string, int> row;
  for (int i = 0; i < p.n(); ++i) row.emplace(p.keys[i], i);
  xv::TableWriter w(a.pos[2]);
  std::unique_ptr<xv::TableWriter> wn;
  if (a.pos.size() == 4) wn.reset(new xv::TableWriter(a.pos[3]));
  std::vector<int32_t> off = {0}, idx;
  std::vector<int> spk_of_seg;
  long num_spk_err = 0, num_utt_done = 0, num_utt_err = 0;
  for (size_t s = 0; s < spk2utt.size(); ++s) {
    const xv::TokenList& e = spk2utt[s];
    if (e.tokens.empty()) {
      fprintf(stderr, "ERROR (%s) Speaker with no utterances.\n", g_prog.c_str());
      return 255;
    }
    int count = 0;
    for (const std::string& utt : e.tokens) {
      auto it = row.find(utt);
      if (it == row.end()) {
        XWARN("No iVector present in input for utterance " << utt);
        ++num_utt_err;
      } else {
        idx.push_back(it->second);
        ++count;
        ++num_utt_done;
      }
    }
    if (count == 0) {
      XWARN("Not producing output for speaker " << e.key << " since no utterances had iVectors");
      ++num_spk_err;
    } else {
      off.push_back((int32_t)idx.size());
      spk_of_seg.push_back((int)s);
    }
  }
  const int n_seg = (int)spk_of_seg.size();
  std::vector<float> means((size_t)n_seg * (p.dim ? p.dim : 1));
  if (n_seg) xv::SegmentMean(dev, p.data.data(), p.n(), p.dim, off.data(), idx.data(), n_seg, /*acc64=*/false, means.data());
  std::vector<double> spk_sum(p.dim, 0.0);
  double spk_sumsq = 0;
  for (int g = 0; g < n_seg; ++g) {
    const float* m = means.data() + (size_t)g * p.dim;
    const std::string& spk = spk2utt[spk_of_seg[g]].key;
    w.WriteVec(spk, m, p.dim);
    if (wn) wn->WriteInt32(spk, off[g + 1] - off[g]);
    for (int k = 0; k < p.dim; ++k) {
      spk_sum[k] += m[k];
      spk_sumsq += (double)m[k] * m[k];
    }
  }
  w.Close();
  if (wn) wn->Close();
  XLOG("Computed mean of " << n_seg << " speakers (" << num_spk_err << " with no utterances), consisting of "
                           << num_utt_done << " utterances (" << num_utt_err << " absent from input).");
  if (n_seg != 0) {
    double mean_sq = 0;
    for (int k = 0; k < p.dim; ++k) mean_sq += (spk_sum[k] / n_seg) * (spk_sum[k] / n_seg);
    XLOG("Norm of mean of speakers is " << sqrt(mean_sq) << ", root-mean-square speaker-iVector length divided by sqrt(dim) is "
                                       << sqrt(spk_sumsq / ((double)n_seg * p.dim)));
  }
  return n_seg != 0 ? 0 : 1;
}

int SubtractGlobalMean(const Args& a) {
  const int dev = PickDevice(a.device);
  if (a.pos.size() == 2) {
    // the mean of the input itself
    xv::SequentialVectorReader r(a.pos[0]);
    Packed p;
    long n_err = 0;
    ReadBatch(r, -1, &p, &n_err);
    XLOG("Read " << p.n() << " iVectors.");
    xv::TableWriter w(a.pos[1]);
    if (p.n() != 0) {
      std::vector<int32_t> off = {0, p.n()}, idx(p.n());
      for (int i = 0; i < p.n(); ++i) idx[i] = i;
      std::vector<float> mean(p.dim), out(p.data.size());
      xv::SegmentMean(dev, p.data.data(), p.n(), p.dim, off.data(), idx.data(), 1, true, mean.data());
      XLOG("Norm of iVector mean was " << Norm(mean.data(), p.dim));
      xv::BackendOptions o;
      o.mean = a.subtract_mean ? mean.data() : nullptr;
      xv::BackendApply(dev, p.data.data(), p.n(), p.dim, o, out.data(), nullptr);
      for (int i = 0; i < p.n(); ++i) w.WriteVec(p.keys[i], out.data() + (size_t)i * p.dim, p.dim);
    }
    w.Close();
    XLOG("Wrote " << p.n() << " mean-subtracted iVectors");
    return p.n() != 0 ? 0 : 1;
  }
  if (a.pos.size() != 3) return -2;
  std::vector<float> mean;
  xv::ReadVectorObject(a.pos[0], &mean);
  xv::SequentialVectorReader r(a.pos[1]);
  xv::TableWriter w(a.pos[2]);
  Packed p;
  long n_err = 0, n_done = 0;
  std::vector<float> out;
  while (ReadBatch(r, kBatch, &p, &n_err)) {
    if (p.dim != (int)mean.size())
      throw xv::KioError("iVector dimension " + std::to_string(p.dim) + " does not match the mean's " + std::to_string(mean.size()));
    out.resize(p.data.size());
    xv::BackendOptions o;
    o.mean = mean.data();
    xv::BackendApply(dev, p.data.data(), p.n(), p.dim, o, out.data(), nullptr);
    for (int i = 0; i < p.n(); ++i) w.WriteVec(p.keys[i], out.data() + (size_t)i * p.dim, p.dim);
    n_done += p.n();
  }
  w.Close();
  XLOG("Wrote " << n_done << " mean-subtracted iVectors");
  return n_done != 0 ? 0 : 1;
}

int TransformVec(const Args& a) {
  if (a.pos.size() != 3) return -2;
  const int dev = PickDevice(a.device);
  xv::Matrix t;
  xv::ReadMatrixObject(a.pos[0], &t);
  xv::SequentialVectorReader r(a.pos[1]);
  xv::TableWriter w(a.pos[2]);
  Packed p;
  long n_err = 0, n_done = 0;
  std::vector<float> out;
  while (ReadBatch(r, kBatch, &p, &n_err)) {
    if (t.cols != p.dim && t.cols != p.dim + 1)
      throw xv::KioError("Dimension mismatch: input vector has dimension " + std::to_string(p.dim) + " and transform has " +
                         std::to_string(t.cols) + " columns.");
    out.resize((size_t)p.n() * t.rows);
    xv::BackendOptions o;
    o.transform = t.data.data();
    o.t_rows = t.rows;
    o.t_cols = t.cols;
    xv::BackendApply(dev, p.data.data(), p.n(), p.dim, o, out.data(), nullptr);
    for (int i = 0; i < p.n(); ++i) w.WriteVec(p.keys[i], out.data() + (size_t)i * t.rows, t.rows);
    n_done += p.n();
  }
  w.Close();
  XLOG("Applied transform to " << n_done << " vectors.");
  return n_done != 0 ? 0 : 1;
}

int NormalizeLength(const Args& a) {
  if (a.pos.size() != 2) return -2;
  const int dev = PickDevice(a.device);
  xv::SequentialVectorReader r(a.pos[0]);
  xv::TableWriter w(a.pos[1]);
  Packed p;
  long n_err = 0, n_done = 0;
  double tot_ratio = 0, tot_ratio2 = 0;
  std::vector<float> out, ratio;
  while (ReadBatch(r, kBatch, &p, &n_err)) {
    out.resize(p.data.size());
    ratio.resize(p.n());
    xv::BackendOptions o;
    o.normalize = a.normalize;
    o.scaleup = a.scaleup;
    xv::BackendApply(dev, p.data.data(), p.n(), p.dim, o, out.data(), ratio.data());
    for (int i = 0; i < p.n(); ++i) {
      if (ratio[i] == 0.f) XWARN("Zero iVector");
      w.WriteVec(p.keys[i], out.data() + (size_t)i * p.dim, p.dim);
      tot_ratio += ratio[i];
      tot_ratio2 += (double)ratio[i] * ratio[i];
    }
    n_done += p.n();
  }
  w.Close();
  XLOG("Processed " << n_done << " iVectors.");
  if (n_done != 0) {
    const double avg = tot_ratio / n_done, var = tot_ratio2 / n_done - avg * avg;
    XLOG("Average ratio of iVector to expected length was " << avg << ", standard deviation was " << sqrt(var > 0 ? var : 0));
  }
  return n_done != 0 ? 0 : 1;
}

const char* Usage(const std::string& prog) {
  if (prog == "ivector-subtract-global-mean")
    return "Copies a table of iVectors but subtracts the global mean as it does so.  The mean may be specified as the first\n"
           "argument; if not, the sum of the input iVectors is used.\n"
           "Usage: ivector-subtract-global-mean [--subtract-mean=true] <ivector-rspecifier> <ivector-wspecifier>\n"
           " or:   ivector-subtract-global-mean <mean-rxfilename> <ivector-rspecifier> <ivector-wspecifier>\n";
  if (prog == "transform-vec")
    return "This program applies a linear or affine transform to individual vectors, e.g. iVectors.\n"
           "Usage: transform-vec [options] <transform-rxfilename> <feats-rspecifier> <feats-wspecifier>\n";
  if (prog == "ivector-normalize-length")
    return "Normalize length of iVectors to equal sqrt(feature-dimension)\n"
           "Usage: ivector-normalize-length [--normalize=true] [--scaleup=true] <ivector-rspecifier> <ivector-wspecifier>\n";
  return "With 3 or 4 arguments, averages iVectors over all the utterances of each speaker using the spk2utt file.\n"
         "With 2 arguments, averages all the iVectors of the input and writes the mean as a single vector.\n"
         "Usage: ivector-mean <spk2utt-rspecifier> <ivector-rspecifier> <ivector-wspecifier> [<num-utt-wspecifier>]\n"
         " or:   ivector-mean [--binary=true] <ivector-rspecifier> <mean-wxfilename>\n";
}

}  // namespace

int main(int argc, char** argv) {
  xv::InstallMappedFileFaultHandler(strrchr(argv[0], '/') ? strrchr(argv[0], '/') + 1 : argv[0]);
  const char* slash = strrchr(argv[0], '/');
  g_prog = slash ? slash + 1 : argv[0];
  Args a;
  std::string cmdline = g_prog;
  for (int i = 1; i < argc; ++i) {
    std::string s = argv[i];
    cmdline += " " + s;
    if (s.compare(0, 2, "--") == 0 && a.pos.empty()) {
      size_t eq = s.find('=');
      const std::string name = s.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
      const std::string val = eq == std::string::npos ? "" : s.substr(eq + 1);
      bool ok = true;
      if (name == "help") {
        fputs(Usage(g_prog), stderr);
        return 0;
      } else if (name == "binary") ok = ParseBool(val, &a.binary);
      else if (name == "normalize") ok = ParseBool(val, &a.normalize);
      else if (name == "scaleup") ok = ParseBool(val, &a.scaleup);
      else if (name == "subtract-mean") ok = ParseBool(val, &a.subtract_mean);
      else if (name == "device") a.device = atoi(val.c_str());
      else if (name == "verbose" || name == "print-args" || name == "config") ok = true;   // accepted, no effect
      else {
        fprintf(stderr, "ERROR (%s) Invalid option %s\n\n%s", g_prog.c_str(), s.c_str(), Usage(g_prog));
        return 255;
      }
      if (!ok) {
        fprintf(stderr, "ERROR (%s) Invalid value for option %s\n", g_prog.c_str(), s.c_str());
        return 255;
      }
      continue;
    }
    a.pos.push_back(s);
  }
  fprintf(stderr, "%s \n", cmdline.c_str());
  try {
    int rc;
    if (g_prog == "ivector-subtract-global-mean") rc = SubtractGlobalMean(a);
    else if (g_prog == "transform-vec") rc = TransformVec(a);
    else if (g_prog == "ivector-normalize-length") rc = NormalizeLength(a);
    else rc = IvectorMean(a);
    if (rc == -2) {
      fputs(Usage(g_prog), stderr);
      return 1;
    }
    return rc;
  } catch (const std::exception& e) {
    fprintf(stderr, "ERROR (%s) %s\n", g_prog.c_str(), e.what());
    return 255;
  }
}
