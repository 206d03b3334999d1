// nnet3-xvector-compute - drop-in command line for the binary the reference's extraction scripts call:
//   egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:86-87,92-93
//   egs/sre/v2/sid/nnet3/xvector/extract_xvectors.sh:80-81,86-87, extract_output_new.sh:82-83,88-89
//
//   nnet3-xvector-compute [options] <raw-nnet-rxfilename> <features-rspecifier> <vector-wspecifier>
//
// Contract honoured: Kaldi-style --name=value options, the three positional forms
// (model through a "nnet3-copy ... |" pipe, features through an "ark:cmd | cmd |" pipe, ark,scp output),
// per-utterance warn-and-skip, "Done N utterances, failed for M", exit 0 iff N > 0, 255 on exception.
// The compute path is the HIP library only: there is no CPU implementation behind --use-gpu=no (the flag
// is accepted so that unchanged recipes run; it is reported and the job still runs on the MI355X).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "engine.h"
#include "knobs.h"
#include "extractor.h"
#include "fuse_pipe.h"
#include "kio.h"
#include "nnet3_raw.h"
#include "program.h"
#include "multi_gpu.h"
#include "table_extract.h"

namespace {

const char* kProg = "nnet3-xvector-compute";  // or "nnet3-compute" (same source, frame-level job), see main()
bool g_frame_job = false;
bool g_apply_exp = false;
int g_verbose = 0;

void LogLine(const char* level, int line, const std::string& msg) {
  fprintf(stderr, "%s (%s[xvec-hip-0.4]:main():nnet3_xvector_compute_main.cc:%d) %s\n", level, kProg, line, msg.c_str());
}
#define XLOG(msg)                        \
  do {                                   \
    std::ostringstream _o;               \
    _o << msg;                           \
    LogLine("LOG", __LINE__, _o.str());  \
  } while (0)
#define XWARN(msg)                          \
  do {                                      \
    std::ostringstream _o;                  \
    _o << msg;                              \
    LogLine("WARNING", __LINE__, _o.str()); \
  } while (0)

const char* kUsage =
    "Propagate features through an xvector neural network model and write the output vectors.\n"
    "\"Xvector\" is our term for a vector or embedding which is the output of a particular type of\n"
    "neural network architecture found in speaker recognition.  MI355X (gfx950) native build.\n"
    "\n"
    "Usage: nnet3-xvector-compute [options] <raw-nnet-in> <features-rspecifier> <vector-wspecifier>\n"
    "e.g.: nnet3-xvector-compute final.raw scp:feats.scp ark:nnet_prediction.ark\n"
    "\n"
    "Options:\n"
    "  --use-gpu=yes|no|optional|wait   accepted for script compatibility; compute always runs on the HIP device\n"
    "  --chunk-size=<int>               chunk length in frames, -1 = whole utterance (default -1)\n"
    "  --min-chunk-size=<int>           minimum chunk length (default 100)\n"
    "  --pad-input=true|false           pad short chunks by edge replication instead of skipping (default true)\n"
    "  --output-node=<name>             compute this node as the output (native form of nnet3-copy --nnet-config)\n"
    "  --nnet-config=<file>             node config lines applied to the model before lowering\n"
    "  --precision=default|fp16mx2|fp16x3|bf16x3|auto|fp16mx|fp16x2|bf16|fp16\n"
    "                                   arithmetic of the MFMA GEMMs.  default: fp16mx2 where every layer can run it, else\n"
    "                                   fp16x3 (nnet3-compute: always fp16x3) - a function of the MODEL alone, so that an\n"
    "                                   utterance gets the same vector whatever job, shard or batch it lands in.\n"
    "                                   fp16x3: split fp16, three MFMAs per\n"
    "                                   product, fp32-grade (3e-7..4e-6 from the fp32 oracle).  fp16mx2: fp16 product +\n"
    "                                   two block-scaled 4-bit products that correct the fp16 rounding of the weights\n"
    "                                   and of the activations, 1.5 passes, ~1.4x faster: 3-5.5e-5 on every model\n"
    "                                   tried; chunks that pool < 160 frames run fp16x3.  auto: chunks that pool >= 300\n"
    "                                   frames run fp16mx (weights corrected only, 1.25 passes, ~1.9x faster than\n"
    "                                   fp16x3), the others fp16x3; its error is the activation rounding averaged by the\n"
    "                                   pooling - 5-8e-5 on models with Kaldi-initialisation-like weights, 1-2e-4 on\n"
    "                                   heavy-tailed ones (DESIGN.md section 3.0): check it on your model first.\n"
    "  --calibration=<file>             with --precision=default (default: $XVEC_CALIBRATION): the SHARED choice of a lighter\n"
    "                                   arithmetic for this model.  The file exists: its choice (fp16mx, or fp16mx2 with some\n"
    "                                   layers in 1.25 passes) is applied; the model fingerprint in it must match.  It does\n"
    "                                   not: the first chunk of 64 utterances of this job - spread evenly over the list of a\n"
    "                                   table that can be addressed, the head of a stream - is computed in fp16x3, fp16mx and\n"
    "                                   fp16mx2, the lightest arithmetic within --calibrate-tol is PUBLISHED to the file\n"
    "                                   (atomically; the first of several concurrent jobs wins) and every job adopts what\n"
    "                                   the file holds.  All jobs of a recipe thus compute in one arithmetic, however the\n"
    "                                   lists were split (extract_xvectors_new.sh:72,91-99)\n"
    "  --calibrate=true|false --calibrate-tol=<float> --calibrate-utts=<int>\n"
    "                                   (default false, 7.5e-5, 64) --calibrate=true without --calibration: measure and choose\n"
    "                                   for THIS job only - the choice then depends on the job's own sample, and two shards\n"
    "                                   of one list may compute in different arithmetics (one-process jobs: --devices)\n"
    "  --fast-min-pooled=<int>          auto / fp16mx2: chunks that pool at least this many frames take the fast\n"
    "                                   kernels (default 300 / 160, or $XVEC_FAST_MIN_POOLED)\n"
    "  --batch-frames=<int>             frames per device batch (default 131072)\n"
    "  --device=<int>                   HIP device index (default: $XVEC_DEVICE, else job index mod #devices)\n"
    "  --devices=all|<i,j,...>          ONE process driving several GPUs (default: $XVEC_DEVICES): the model is read and packed\n"
    "                                   once, sent to the listed devices with one RCCL broadcast, whole batches are dealt\n"
    "                                   to them in turn and written in table order - the output is byte-identical to a\n"
    "                                   one-GPU run.  This is what `extract_xvectors_new.sh --nj 1 --use-gpu true` needs to\n"
    "                                   use the whole node without a change of the script (nnet3-xvector-compute only)\n"
    "  --cmn-window=<int> --cmn-center=true|false --vad-rspecifier=<rspecifier>\n"
    "                                   run apply-cmvn-sliding (--norm-vars=false) and select-voiced-frames on the device\n"
    "                                   in front of the network; the features rspecifier is then the raw feats.scp.\n"
    "                                   (A features rspecifier that IS the recipes' pipeline - 'ark:apply-cmvn-sliding\n"
    "                                   --norm-vars=false --center=.. --cmn-window=.. <table> ark:- | select-voiced-frames ark:-\n"
    "                                   <vad table> ark:- |', extract_xvectors_new.sh:79 - is recognised and run this way by\n"
    "                                   itself; the log says so.  Anything else in a pipe is run as a command.)\n"
    "  --backend-mean=<vec> --backend-transform=<mat> --backend-normalize-length=true|false [--backend-scaleup=true]\n"
    "                                   apply ivector-subtract-global-mean | transform-vec | ivector-normalize-length\n"
    "                                   on the device to every embedding before it is written\n"
    "  --profile-json=<file>            per-kernel device time of the whole job (HIP events stamped by the dispatches)\n"
    "  --config=<file>  --verbose=<int>  --print-args=true|false  --help\n";

struct Options {
  std::string use_gpu = "no";
  int chunk_size = -1;
  int min_chunk_size = 100;
  bool pad_input = true;
  std::string output_node;
  std::string nnet_config;
  std::string precision = "default";
  int batch_frames = 1 << 17;
  int fast_min_pooled = -1;
  bool calibrate = false;          // a job measures for itself only when asked to (the choice then depends on its sample)
  std::string calibration;         // --calibration / $XVEC_CALIBRATION: the shared choice of the recipe (calib_file.h)
  float calibrate_tol = 7.5e-5f;   // three quarters of the 1e-4 bar, on the WORST calibration chunk
  int calibrate_utts = 64;         // utterances sampled (spread over the list of an addressable table, the head of a stream)
  int device = -1;
  std::string devices;             // --devices: "all" or a list; empty: one device (--device rule)
  bool print_args = true;
  int cmn_window = 0;
  bool cmn_center = true;
  std::string vad_rspecifier;
  std::string backend_mean, backend_transform;
  std::string profile_json;
  bool backend_normalize = false, backend_scaleup = true;
};

bool ParseBool(const std::string& v, bool* out) {
  if (v == "true" || v == "t" || v == "1" || v.empty()) *out = true;
  else if (v == "false" || v == "f" || v == "0") *out = false;
  else return false;
  return true;
}

// returns false on a fatal option error
bool ApplyOption(const std::string& name, const std::string& value, bool has_value, Options* o, std::string* err);

bool ReadConfigFile(const std::string& path, Options* o, std::string* err) {
  std::ifstream f(path);
  if (!f) {
    *err = "cannot open config file " + path;
    return false;
  }
  std::string line;
  while (std::getline(f, line)) {
    size_t h = line.find('#');
    if (h != std::string::npos) line = line.substr(0, h);
    size_t a = line.find_first_not_of(" \t\r");
    if (a == std::string::npos) continue;
    size_t b = line.find_last_not_of(" \t\r");
    line = line.substr(a, b - a + 1);
    if (line.compare(0, 2, "--") != 0) {
      *err = "bad line in config file: " + line;
      return false;
    }
    size_t eq = line.find('=');
    std::string name = line.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
    std::string val = eq == std::string::npos ? "" : line.substr(eq + 1);
    if (!ApplyOption(name, val, eq != std::string::npos, o, err)) return false;
  }
  return true;
}

bool ApplyOption(const std::string& name_in, const std::string& value, bool has_value, Options* o, std::string* err) {
  std::string name = name_in;
  for (char& c : name)
    if (c == '_') c = '-';
  auto need_int = [&](int* dst) {
    char* end = nullptr;
    long v = strtol(value.c_str(), &end, 10);
    if (!has_value || end == value.c_str() || *end) {
      *err = "invalid integer for --" + name + ": '" + value + "'";
      return false;
    }
    *dst = (int)v;
    return true;
  };
  if (name == "use-gpu") {
    if (value != "yes" && value != "no" && value != "optional" && value != "wait" && value != "true" && value != "false") {
      *err = "invalid value for --use-gpu: " + value;
      return false;
    }
    o->use_gpu = value;
  } else if (name == "chunk-size") return need_int(&o->chunk_size);
  else if (name == "min-chunk-size") return need_int(&o->min_chunk_size);
  else if (name == "batch-frames") return need_int(&o->batch_frames);
  else if (name == "fast-min-pooled") return need_int(&o->fast_min_pooled);
  else if (name == "calibration") {
    if (!has_value || value.empty()) {
      *err = "--calibration needs a file name";
      return false;
    }
    o->calibration = value;
  } else if (name == "calibrate") {
    if (!ParseBool(value, &o->calibrate)) {
      *err = "invalid boolean for --calibrate: " + value;
      return false;
    }
  } else if (name == "calibrate-tol") {
    char* end = nullptr;
    o->calibrate_tol = strtof(value.c_str(), &end);
    if (!has_value || end == value.c_str() || *end || !(o->calibrate_tol > 0.f)) {
      *err = "invalid value for --calibrate-tol: " + value;
      return false;
    }
  }
  else if (name == "calibrate-utts") return need_int(&o->calibrate_utts);
  else if (name == "device") return need_int(&o->device);
  else if (name == "devices") {
    if (!has_value || value.empty()) {
      *err = "--devices needs a value (all, or a comma-separated list of device indices)";
      return false;
    }
    o->devices = value;
  }
  else if (name == "cmn-window") return need_int(&o->cmn_window);
  else if (name == "vad-rspecifier") o->vad_rspecifier = value;
  else if (name == "profile-json") o->profile_json = value;
  else if (name == "backend-mean") o->backend_mean = value;
  else if (name == "backend-transform") o->backend_transform = value;
  else if (name == "backend-normalize-length") {
    if (!ParseBool(value, &o->backend_normalize)) {
      *err = "invalid boolean for --backend-normalize-length: " + value;
      return false;
    }
  } else if (name == "backend-scaleup") {
    if (!ParseBool(value, &o->backend_scaleup)) {
      *err = "invalid boolean for --backend-scaleup: " + value;
      return false;
    }
  }
  else if (name == "cmn-center") {
    if (!ParseBool(value, &o->cmn_center)) {
      *err = "invalid boolean for --cmn-center: " + value;
      return false;
    }
  }
  else if (name == "verbose") return need_int(&g_verbose);
  else if (name == "pad-input") {
    if (!ParseBool(value, &o->pad_input)) {
      *err = "invalid boolean for --pad-input: " + value;
      return false;
    }
  } else if (name == "print-args") {
    if (!ParseBool(value, &o->print_args)) {
      *err = "invalid boolean for --print-args: " + value;
      return false;
    }
  } else if (name == "apply-exp") {
    if (!ParseBool(value, &g_apply_exp)) {
      *err = "invalid boolean for --apply-exp: " + value;
      return false;
    }
  } else if (name == "output-node") o->output_node = value;
  else if (name == "nnet-config") o->nnet_config = value;
  else if (name == "precision") o->precision = value;
  else if (name == "config") return ReadConfigFile(value, o, err);
  else if (name == "ivectors" || name == "online-ivectors" ||
           (name == "frame-subsampling-factor" && value != "1") || (name == "use-priors" && value != "false")) {
    // options of upstream's nnet3-compute that CHANGE the result (steps/nnet3/compute_output.sh:117,
    // make_bottleneck_features_new.sh:109 with an ivector directory): ignoring them would write different numbers under
    // the same command line
    *err = "option --" + name + (has_value ? "=" + value : "") + " is not supported (it changes the output; this tool has no i-vector input, "
           "output subsampling or prior subtraction)";
    return false;
  } else {
    // the rest of upstream's surface (compiler / optimisation / cached-compiler options, chunking of the recurrent-network
    // tools: --frames-per-chunk, --extra-left-context ...) has no meaning for these feed-forward networks: accept and
    // ignore, like the contract in SURVEY.md §8(b) asks
    XWARN("ignoring option --" << name << (has_value ? "=" + value : ""));
  }
  return true;
}

int JobIndexFromWspecifier(const std::string& w) {
  // ".../xvector_name.<JOB>.ark" (extract_xvectors_new.sh:87,93)
  size_t ark = w.find(".ark");
  if (ark == std::string::npos || ark == 0) return -1;
  size_t e = ark, b = e;
  while (b > 0 && isdigit((unsigned char)w[b - 1])) --b;
  if (b == e || b == 0 || w[b - 1] != '.') return -1;
  return atoi(w.substr(b, e - b).c_str());
}

}  // namespace

int main(int argc, char** argv) {
  xv::InstallMappedFileFaultHandler(strrchr(argv[0], '/') ? strrchr(argv[0], '/') + 1 : argv[0]);
  {
    // one source, two drop-ins: invoked as `nnet3-compute` it does the frame-level job (a matrix per utterance)
    const char* base = strrchr(argv[0], '/') ? strrchr(argv[0], '/') + 1 : argv[0];
    if (strcmp(base, "nnet3-compute") == 0) {
      g_frame_job = true;
      kProg = "nnet3-compute";
    }
  }
  try {
    Options opt;
    std::vector<std::string> pos;
    std::string err;
    for (int i = 1; i < argc; ++i) {
      std::string a = argv[i];
      if (a == "--help" || a == "-h") {
        fputs(kUsage, stderr);
        return 0;
      }
      if (a.compare(0, 2, "--") == 0 && pos.empty()) {
        size_t eq = a.find('=');
        std::string name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
        std::string val = eq == std::string::npos ? "" : a.substr(eq + 1);
        if (!ApplyOption(name, val, eq != std::string::npos, &opt, &err)) {
          fprintf(stderr, "%s: %s\n\n%s", kProg, err.c_str(), kUsage);
          return 1;
        }
      } else {
        pos.push_back(a);
      }
    }
    if (opt.print_args) {
      std::string cmd;
      for (int i = 0; i < argc; ++i) {
        std::string a = argv[i];
        bool quote = a.find_first_of(" |;&()<>") != std::string::npos;
        cmd += (i ? " " : "") + (quote ? "'" + a + "'" : a);
      }
      fprintf(stderr, "%s\n", cmd.c_str());
    }
    if (pos.size() != 3) {
      fputs(kUsage, stderr);
      return 1;
    }
    const std::string nnet_rx = pos[0], vec_wspec = pos[2];
    std::string feat_rspec = pos[1];
    // The feature pipeline the reference's extraction scripts build (extract_xvectors_new.sh:79: apply-cmvn-sliding |
    // select-voiced-frames, two CPU tools and two pipes) is recognised as text and run on the device instead (fuse_pipe.h) -
    // unless the caller asked for a front-end of their own, or XVEC_DEBUG=fuse_pipe=0 says to run the commands.
    int fused_min_window = 0;
    if (!g_frame_job && opt.cmn_window == 0 && opt.vad_rspecifier.empty() && xv::DebugKnobInt("fuse_pipe", 1) != 0) {
      xv::FusedPipeline fp;
      if (xv::RecognizeFeaturePipeline(feat_rspec, &fp)) {
        XLOG("feature pipeline recognised (apply-cmvn-sliding --norm-vars=false --center=" << (fp.center ? "true" : "false")
             << " --cmn-window=" << fp.cmn_window << (fp.vad_rspecifier.empty() ? "" : " | select-voiced-frames " + fp.vad_rspecifier)
             << "): it runs on the device, reading " << fp.feats_rspecifier << " directly (XVEC_DEBUG=fuse_pipe=0 runs the commands)");
        feat_rspec = fp.feats_rspecifier;
        opt.cmn_window = fp.cmn_window;
        opt.cmn_center = fp.center;
        opt.vad_rspecifier = fp.vad_rspecifier;
        fused_min_window = fp.min_cmn_window;
      }
    }

    // default: the library's one policy (XV_PREC_DEFAULT, engine.cc PackModelPolicy): fp16mx2 (1.5 MFMA passes, 3-5.5e-5 from
    // the fp32 oracle on every model tried, the heavy-tailed ones included) for pooled outputs where every layer can run
    // it, else fp16x3 (fp32-grade, three passes); frame-level outputs: fp16x3
    int precision;
    if (opt.precision == "default") precision = xv::kPrecDefault;
    else if (opt.precision == "bf16x3") precision = xv::kPrecBf16x3;
    else if (opt.precision == "bf16") precision = xv::kPrecBf16;
    else if (opt.precision == "fp16") precision = xv::kPrecFp16;
    else if (opt.precision == "fp16x3") precision = xv::kPrecFp16x3;
    else if (opt.precision == "fp16x2") precision = xv::kPrecFp16x2;
    else if (opt.precision == "auto") precision = xv::kPrecAuto;
    else if (opt.precision == "fp16mx") precision = xv::kPrecFp16Mx;
    else if (opt.precision == "fp16mx2") precision = xv::kPrecFp16Mx2;
    else {
      fprintf(stderr, "%s: invalid --precision=%s\n", kProg, opt.precision.c_str());
      return 1;
    }
    if (opt.use_gpu == "no" || opt.use_gpu == "false")
      XWARN("--use-gpu=no requested, but this build has no CPU compute path: running on the HIP device "
            "(the option is accepted so that unchanged recipes work)");

    // ---- model ---------------------------------------------------------------------------------
    const auto t_start = std::chrono::steady_clock::now();
    auto since_start = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    if (opt.fast_min_pooled >= 0) setenv("XVEC_FAST_MIN_POOLED", std::to_string(opt.fast_min_pooled).c_str(), 1);   // read by the engine
    // (Bringing the HIP runtime up on a second thread meanwhile was tried: its 0.1-0.2 s of mmap / ioctl traffic hold the
    // address-space lock the page faults of the 28 MB model read wait for - the read went from 0.09 to 0.30 s.)
    xv::RawNnet net;
    net.ReadFrom(nnet_rx);
    if (!opt.nnet_config.empty()) {
      std::ifstream f(opt.nnet_config);
      if (!f) throw xv::KioError("cannot open --nnet-config file " + opt.nnet_config);
      std::stringstream ss;
      ss << f.rdbuf();
      net.ApplyNnetConfig(ss.str());
    }
    if (!opt.output_node.empty()) net.ApplyNnetConfig("output-node name=output input=" + opt.output_node);
    xv::TdnnProgram prog = xv::LowerToProgram(net, "output");
    if (g_verbose >= 1) XLOG("lowered model:\n" << prog.Describe());
    if (!g_frame_job && !prog.output_is_segment)
      throw xv::KioError("the output node is frame-level; nnet3-xvector-compute expects a pooled (x-vector) output");
    if (g_frame_job && prog.output_is_segment)
      throw xv::KioError("the output node follows the statistics pooling; nnet3-compute expects a frame-level output");

    const double t_model = since_start();
    // ---- device --------------------------------------------------------------------------------
    const bool policy_default = opt.precision == "default";
    const std::vector<uint8_t> blob = xv::PackModelPolicy(prog, precision, &precision);
    const double t_pack = since_start();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      throw xv::EngineError("no HIP device available and this build has no CPU path");
    int device = opt.device;
    if (device < 0 && getenv("XVEC_DEVICE")) device = atoi(getenv("XVEC_DEVICE"));
    if (device < 0) {
      int job = JobIndexFromWspecifier(vec_wspec);
      device = job > 0 ? (job - 1) % ndev : 0;
    }
    if (device >= ndev) device %= ndev;
    const double t_hip = since_start();
    if (policy_default) opt.precision = std::string("default = ") + xv::PrecisionName(precision);
    // one process, several GPUs (--devices / XVEC_DEVICES): one RCCL broadcast of the packed image, an engine per device
    std::string dev_spec = opt.devices;
    if (dev_spec.empty() && opt.device < 0 && getenv("XVEC_DEVICES")) dev_spec = getenv("XVEC_DEVICES");
    std::vector<int> dev_list = xv::ParseDeviceList(dev_spec, ndev);
    if (!dev_list.empty() && g_frame_job) {
      XWARN("--devices is ignored by nnet3-compute (one device: " << dev_list[0] << ")");
      device = dev_list[0];
      dev_list.clear();
    }
    std::vector<std::unique_ptr<xv::Engine>> engines;
    // test knob: XVEC_DEBUG=engines_on_one_device=n builds n engines on the ONE device (each from the host image, no RCCL - the library
    // refuses two ranks on one device), so that the several-engine path of the table loop (a consumer thread per engine, ordered
    // writer) can be exercised on a one-GPU box.  No speed-up to be had from it.
    const int n_same = xv::DebugKnobInt("engines_on_one_device", 0);
    if (dev_list.empty() && n_same > 1 && n_same <= 8 && !g_frame_job) {
      for (int i = 0; i < n_same; ++i) engines.emplace_back(new xv::Engine(blob.data(), blob.size(), device));
      XLOG("XVEC_DEBUG=engines_on_one_device: " << n_same << " engines on device " << device << " (test knob)");
    } else if (dev_list.empty()) {
      engines.emplace_back(new xv::Engine(blob.data(), blob.size(), device));
    } else {
      engines = xv::CreateEnginesBroadcast(blob, dev_list);
      device = dev_list[0];
    }
    xv::Engine& engine = *engines[0];
    if (getenv("XVEC_TIMING"))
      XLOG("start-up stages: model read + lowered " << t_model << " s, weights packed " << (t_pack - t_model)
                                                       << " s, HIP runtime up " << (t_hip - t_pack)
                                                       << " s, engine (upload, buffers, streams) " << (since_start() - t_hip) << " s");
    {
      std::ostringstream dv;
      if (dev_list.empty()) {
        dv << "device " << device << " of " << ndev;
      } else {
        dv << dev_list.size() << " of " << ndev << " devices (";
        for (size_t i = 0; i < dev_list.size(); ++i) dv << (i ? "," : "") << dev_list[i];
        dv << "; weights sent with one RCCL broadcast from device " << dev_list[0] << ")";
      }
      XLOG("model: " << prog.layers.size() << " layers, context " << prog.left_context << "/" << prog.right_context
                     << ", embedding dim " << prog.output_dim << "; " << dv.str() << ", precision " << opt.precision << ", "
                     << (engine.weight_bytes() >> 20) << " MiB of packed weights");
    }

    // (every engine of a --devices job records its own launches: the report sums them per kernel, so that the launch counts and
    // times stand next to utterance and frame counts of the SAME whole job - ADVICE r05)
    if (!opt.profile_json.empty())
      for (auto& e : engines) e->SetProfiling(true);
    // --profile-json: {"kernels": [{"name", "launches", "total_ms"}], "utterances", "failed", "frames", "seconds"}
    auto write_profile = [&](const xv::TableExtractResult& r) {
      if (opt.profile_json.empty()) return;
      std::vector<std::string> names;
      std::map<std::string, std::pair<long, double>> sum;
      for (auto& e : engines) {
        std::istringstream rep(e->ProfileReport());
        std::string line;
        while (std::getline(rep, line)) {
          const size_t a = line.find('\t'), b = line.rfind('\t');
          if (a == std::string::npos || b == a) continue;
          const std::string name = line.substr(0, a);
          if (!sum.count(name)) names.push_back(name);
          sum[name].first += atol(line.substr(a + 1, b - a - 1).c_str());
          sum[name].second += atof(line.substr(b + 1).c_str());
        }
      }
      std::ostringstream js;
      js.precision(9);
      js << "{\"kernels\": [";
      bool first = true;
      for (const std::string& name : names) {
        js << (first ? "" : ", ") << "{\"name\": \"" << name << "\", \"launches\": " << sum[name].first << ", \"total_ms\": " << sum[name].second << "}";
        first = false;
      }
      js << "], \"utterances\": " << r.num_success << ", \"failed\": " << r.num_fail << ", \"frames\": " << (long)r.frames
         << ", \"seconds\": " << r.seconds << ", \"device\": " << device << ", \"precision\": \"" << opt.precision << "\"}\n";
      xv::Output out;
      out.Open(opt.profile_json);
      out.Puts(js.str());
      out.Close();
    };

    if (g_frame_job) {
      xv::TableExtractResult fr = xv::RunTableCompute(
          &engine, opt.batch_frames, g_apply_exp, feat_rspec, vec_wspec,
          [](const char* level, const std::string& m) { LogLine(level, 0, m); });
      XLOG("Time taken " << fr.seconds << "s: real-time factor assuming 100 frames/sec is "
                         << (fr.seconds * 100.0 / std::max(fr.frames, 1.0)));
      XLOG("Done " << fr.num_success << " utterances, failed for " << fr.num_fail);
      write_profile(fr);
      return fr.num_success != 0 ? 0 : 1;
    }
    // ---- the utterance loop (reader thread -> batches -> device -> ark,scp writer) ---------------------------
    xv::ExtractOptions eo;
    eo.chunk_size = opt.chunk_size;
    eo.min_chunk_size = opt.min_chunk_size;
    eo.pad_input = opt.pad_input;
    eo.max_batch_rows = opt.batch_frames;
    eo.cmn_window = opt.cmn_window;
    eo.cmn_center = opt.cmn_center;
    eo.vad_rspecifier = opt.vad_rspecifier;
    if (fused_min_window > 0) eo.cmn_min_window = fused_min_window;
    eo.calibrate = policy_default && opt.calibrate && !g_frame_job;   // no-op unless the context can switch (fp16mx2)
    eo.calibrate_tol = opt.calibrate_tol;
    if (policy_default && !g_frame_job) {
      const char* e = getenv("XVEC_CALIBRATION");
      eo.calibration_file = !opt.calibration.empty() ? opt.calibration : std::string(e ? e : "");
    } else if (!opt.calibration.empty()) {
      XWARN("--calibration is used with --precision=default only; ignored");
    }
    if (opt.calibrate_utts > 0) eo.calibrate_utts = opt.calibrate_utts;
    if (!opt.backend_mean.empty()) xv::ReadVectorObject(opt.backend_mean, &eo.backend_mean);
    if (!opt.backend_transform.empty()) {
      xv::Matrix t;
      xv::ReadMatrixObject(opt.backend_transform, &t);
      eo.backend_transform = t.data;
      eo.backend_t_rows = t.rows;
      eo.backend_t_cols = t.cols;
    }
    eo.backend_normalize = opt.backend_normalize;
    eo.backend_scaleup = opt.backend_scaleup;
    std::vector<xv::Engine*> eng_ptrs;
    for (auto& e : engines) eng_ptrs.push_back(e.get());
    xv::TableExtractResult res = xv::RunTableExtraction(
        eng_ptrs, eo, feat_rspec, vec_wspec, [](const char* level, const std::string& m) { LogLine(level, 0, m); });
    if (res.reader_status != 0) XWARN("feature input command exited with status " << res.reader_status);
    XLOG("Time taken " << res.seconds << "s: real-time factor assuming 100 frames/sec is "
                       << (res.seconds * 100.0 / std::max(res.frames, 1.0)));
    XLOG("Done " << res.num_success << " utterances, failed for " << res.num_fail);
    write_profile(res);
    return res.num_success != 0 ? 0 : 1;
  } catch (const std::exception& e) {
    fprintf(stderr, "ERROR (%s[xvec-hip-0.4]:main()) %s\n", kProg, e.what());
    return -1;
  }
}
