// extern "C" boundary of libxvec_hip.so - see include/xvec_hip.h.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <memory>
#include <string>
#include <vector>

#include "../../include/xvec_hip.h"
#include "backend.h"
#include "engine.h"
#include "extractor.h"
#include "kernels.h"
#include "calib_file.h"
#include "fuse_pipe.h"
#include "multi_gpu.h"
#include "nnet3_raw.h"
#include "program.h"
#include "table_extract.h"

struct xv_model {
  xv::TdnnProgram prog;
};

struct xv_ctx {
  std::unique_ptr<xv::Engine> eng;
  bool calibrate = false;       // xv_ctx_set_calibration: table jobs calibrate on the head of their table first
  float calibrate_tol = 7.5e-5f;   // three quarters of the 1e-4 bar, on the WORST calibration chunk
  std::string calibration_file;    // xv_ctx_set_calibration_file: the shared choice of the recipe (calib_file.h)
};

namespace {

thread_local std::string g_err;

xv_status Fail(xv_status s, const std::string& m) {
  g_err = m;
  return s;
}

template <class F>
xv_status Guard(F&& f) {
  try {
    g_err.clear();
    return f();
  } catch (const xv::KioError& e) {
    return Fail(XV_ERR_IO, e.what());
  } catch (const xv::EngineError& e) {
    return Fail(XV_ERR_DEVICE, e.what());
  } catch (const std::bad_alloc&) {
    return Fail(XV_ERR_INTERNAL, "out of host memory");
  } catch (const std::exception& e) {
    return Fail(XV_ERR_INTERNAL, e.what());
  } catch (...) {
    return Fail(XV_ERR_INTERNAL, "unknown exception");
  }
}

void FillInfo(const xv::BlobInfo& b, xv_model_info_t* info) {
  memset(info, 0, sizeof *info);
  info->input_dim = b.input_dim;
  info->output_dim = b.output_dim;
  info->left_context = b.left_context;
  info->right_context = b.right_context;
  info->min_frames = b.min_frames;
  info->num_layers = (int32_t)b.layers.size();
  info->output_is_segment = b.output_is_segment;
}

xv_status LoadCommon(xv::RawNnet& net, const char* nnet_config, const char* output_node, xv_model** out) {
  if (nnet_config && *nnet_config) net.ApplyNnetConfig(nnet_config);
  std::unique_ptr<xv_model> m(new xv_model);
  try {
    m->prog = xv::LowerToProgram(net, output_node && *output_node ? output_node : "output");
  } catch (const xv::KioError& e) {
    return Fail(XV_ERR_MODEL, e.what());
  }
  *out = m.release();
  return XV_OK;
}

}  // namespace

extern "C" {

const char* xv_last_error(void) { return g_err.c_str(); }
// XVEC_KERNELS_SHA: first 16 hex digits of the SHA-1 of kernels.hip + kernels.h this library was built from (Makefile; the stamp
// tools/parse_rocprof.py puts on profiles/pmc_traffic.json): tells a stale or a differently configured build of the library from
// the tree's (tests/test_gpu_fuzz.py compares the product and the schedule-fuzzing build with it).
#ifndef XVEC_KERNELS_SHA
#define XVEC_KERNELS_SHA "unknown"
#endif
#ifdef XVEC_SCHED_FUZZ
#define XVEC_BUILD_KIND "; schedule-fuzzing build"
#else
#define XVEC_BUILD_KIND ""
#endif
const char* xv_version(void) { return "xvec_hip 0.4 (gfx950; kernels " XVEC_KERNELS_SHA XVEC_BUILD_KIND ")"; }

xv_status xv_model_load(const void* raw, size_t n, const char* nnet_config, const char* output_node, xv_model** out) {
  if (!raw || !out) return Fail(XV_ERR_ARG, "xv_model_load: null argument");
  return Guard([&] {
    xv::RawNnet net;
    net.Read(std::string((const char*)raw, n));
    return LoadCommon(net, nnet_config, output_node, out);
  });
}

xv_status xv_model_load_rxfilename(const char* rxfilename, const char* nnet_config, const char* output_node,
                                   xv_model** out) {
  if (!rxfilename || !out) return Fail(XV_ERR_ARG, "xv_model_load_rxfilename: null argument");
  return Guard([&] {
    xv::RawNnet net;
    net.ReadFrom(rxfilename);
    return LoadCommon(net, nnet_config, output_node, out);
  });
}

void xv_model_free(xv_model* m) { delete m; }

xv_status xv_model_info(const xv_model* m, xv_model_info_t* info) {
  if (!m || !info) return Fail(XV_ERR_ARG, "xv_model_info: null argument");
  memset(info, 0, sizeof *info);
  info->input_dim = m->prog.input_dim;
  info->output_dim = m->prog.output_dim;
  info->left_context = m->prog.left_context;
  info->right_context = m->prog.right_context;
  info->min_frames = m->prog.min_frames;
  info->num_layers = (int32_t)m->prog.layers.size();
  info->output_is_segment = m->prog.output_is_segment;
  return XV_OK;
}

double xv_model_macs(const xv_model* m, int32_t frames) { return m ? m->prog.Macs(frames) : 0.0; }

size_t xv_model_describe(const xv_model* m, char* buf, size_t n) {
  if (!m) return 0;
  const std::string s = m->prog.Describe();
  if (buf && n) {
    const size_t k = s.size() < n - 1 ? s.size() : n - 1;
    memcpy(buf, s.data(), k);
    buf[k] = 0;
  }
  return s.size() + 1;
}

xv_status xv_model_pack(const xv_model* m, int precision, void* blob, size_t* nbytes) {
  if (!m || !nbytes) return Fail(XV_ERR_ARG, "xv_model_pack: null argument");
  return Guard([&] {
    std::vector<uint8_t> b = xv::PackModelPolicy(m->prog, precision);
    if (blob) {
      if (*nbytes < b.size()) return Fail(XV_ERR_ARG, "xv_model_pack: buffer too small");
      memcpy(blob, b.data(), b.size());
    }
    *nbytes = b.size();
    return XV_OK;
  });
}

xv_status xv_ctx_create(const xv_model* m, int device, int precision, xv_ctx** out) {
  if (!m || !out) return Fail(XV_ERR_ARG, "xv_ctx_create: null argument");
  return Guard([&] {
    std::vector<uint8_t> b = xv::PackModelPolicy(m->prog, precision);
    std::unique_ptr<xv_ctx> c(new xv_ctx);
    c->eng.reset(new xv::Engine(b.data(), b.size(), device));
    *out = c.release();
    return XV_OK;
  });
}

xv_status xv_ctx_create_from_blob(const void* blob, size_t nbytes, int device, xv_ctx** out) {
  if (!blob || !out) return Fail(XV_ERR_ARG, "xv_ctx_create_from_blob: null argument");
  return Guard([&] {
    std::unique_ptr<xv_ctx> c(new xv_ctx);
    c->eng.reset(new xv::Engine((const uint8_t*)blob, nbytes, device));
    *out = c.release();
    return XV_OK;
  });
}

xv_status xv_ctx_create_from_device_blob(const void* blob_dev, size_t nbytes, int device, xv_ctx** out) {
  if (!blob_dev || !out) return Fail(XV_ERR_ARG, "xv_ctx_create_from_device_blob: null argument");
  return Guard([&] {
    if (hipSetDevice(device) != hipSuccess) return Fail(XV_ERR_DEVICE, "xv_ctx_create_from_device_blob: bad device");
    // header + layer table come back to the host (a few KB); the weights stay on the device
    std::vector<uint8_t> head = xv::ReadBlobHead(blob_dev, nbytes);
    std::unique_ptr<xv_ctx> c(new xv_ctx);
    c->eng.reset(new xv::Engine(head.data(), nbytes, device, blob_dev));
    *out = c.release();
    return XV_OK;
  });
}

void xv_ctx_free(xv_ctx* c) { delete c; }

xv_status xv_ctx_info(const xv_ctx* c, xv_model_info_t* info, int32_t* precision, int32_t* device) {
  if (!c) return Fail(XV_ERR_ARG, "xv_ctx_info: null context");
  if (info) FillInfo(c->eng->info(), info);
  if (precision) *precision = c->eng->info().precision;
  if (device) *device = c->eng->device();
  return XV_OK;
}

xv_status xv_forward_batch(xv_ctx* c, const float* feats, const int32_t* row_offsets, int32_t B, float* out) {
  if (!c || !feats || !row_offsets || !out) return Fail(XV_ERR_ARG, "xv_forward_batch: null argument");
  return Guard([&] {
    for (int b = 0; b < B; ++b)
      if (row_offsets[b + 1] - row_offsets[b] < (c->eng->frame_mode() ? 1 : c->eng->info().min_frames))
        return Fail(XV_ERR_ARG, "xv_forward_batch: chunk " + std::to_string(b) + " has fewer than min_frames rows");
    c->eng->ForwardHost(feats, row_offsets, B, out);
    return XV_OK;
  });
}

xv_status xv_forward_batch_device(xv_ctx* c, const float* feats_dev, const int32_t* row_offsets, int32_t B,
                                  float* out_dev, int32_t out_ld, void* hip_stream) {
  if (!c || !feats_dev || !row_offsets || !out_dev) return Fail(XV_ERR_ARG, "xv_forward_batch_device: null argument");
  return Guard([&] {
    for (int b = 0; b < B; ++b)
      if (row_offsets[b + 1] - row_offsets[b] < (c->eng->frame_mode() ? 1 : c->eng->info().min_frames))
        return Fail(XV_ERR_ARG, "xv_forward_batch_device: chunk " + std::to_string(b) + " has fewer than min_frames rows");
    if (out_ld < c->eng->info().output_dim) return Fail(XV_ERR_ARG, "xv_forward_batch_device: out_ld < output_dim");
    std::shared_ptr<xv::Engine::Plan> plan = c->eng->MakePlan(row_offsets, B);
    c->eng->Forward(*plan, feats_dev, out_dev, out_ld, (hipStream_t)hip_stream);
    return XV_OK;
  });
}

xv_status xv_ctx_synchronize(xv_ctx* c) {
  if (!c) return Fail(XV_ERR_ARG, "xv_ctx_synchronize: null context");
  return Guard([&] {
    if (hipSetDevice(c->eng->device()) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
      return Fail(XV_ERR_DEVICE, "hipDeviceSynchronize failed");
    c->eng->CheckKernelFaults();
    return XV_OK;
  });
}

static void FillCalibration(const xv::Engine::Calibration& c, xv_calibration* out) {
  if (!out) return;
  out->chosen = c.chosen;
  out->checked = c.checked;
  out->err_mx = c.err_mx;
  out->err_mx2 = c.err_mx2;
  out->checked_mx = c.checked_mx;
  out->err_lite = c.err_lite;
  out->lite_mask = c.lite_mask;
  out->err_holdout = c.err_holdout;
  out->checked_holdout = c.checked_holdout;
  out->lite_dropped = c.lite_dropped;
  out->tail = c.tail;
}

xv_status xv_ctx_calibrate(xv_ctx* c, const float* feats, const int32_t* row_offsets, int32_t B, float tol, xv_calibration* out) {
  if (!c || !feats || !row_offsets || B < 1) return Fail(XV_ERR_ARG, "xv_ctx_calibrate: bad argument");
  return Guard([&] {
    FillCalibration(c->eng->Calibrate(feats, row_offsets, B, tol), out);
    return XV_OK;
  });
}

xv_status xv_ctx_set_fast_mode(xv_ctx* c, int32_t precision) {
  if (!c) return Fail(XV_ERR_ARG, "xv_ctx_set_fast_mode: null context");
  return Guard([&] {
    if (precision == c->eng->fast_mode() && c->eng->lite_mask() == 0) return XV_OK;
    if (!c->eng->can_switch_fast_mode() || (precision != XV_PREC_FP16MX2 && precision != XV_PREC_FP16MX && precision != XV_PREC_FP16X3))
      return Fail(XV_ERR_ARG, "xv_ctx_set_fast_mode: the context must be packed as XV_PREC_FP16MX2 (pooled output) and the mode one of "
                              "XV_PREC_FP16MX2, XV_PREC_FP16MX, XV_PREC_FP16X3");
    c->eng->SetFastMode(precision);
    return XV_OK;
  });
}

xv_status xv_ctx_fast_mode(const xv_ctx* c, int32_t* precision) {
  if (!c || !precision) return Fail(XV_ERR_ARG, "xv_ctx_fast_mode: null argument");
  *precision = c->eng->fast_mode();
  return XV_OK;
}

xv_status xv_ctx_set_lite_layers(xv_ctx* c, uint64_t mask) {
  if (!c) return Fail(XV_ERR_ARG, "xv_ctx_set_lite_layers: null context");
  return Guard([&] {
    if (mask == c->eng->lite_mask()) return XV_OK;
    if (!c->eng->can_switch_fast_mode() || c->eng->fast_mode() != XV_PREC_FP16MX2)
      return Fail(XV_ERR_ARG, "xv_ctx_set_lite_layers: the context must be running XV_PREC_FP16MX2");
    c->eng->SetLiteMask(mask);
    return XV_OK;
  });
}

xv_status xv_ctx_lite_layers(const xv_ctx* c, uint64_t* mask) {
  if (!c || !mask) return Fail(XV_ERR_ARG, "xv_ctx_lite_layers: null argument");
  *mask = c->eng->lite_mask();
  return XV_OK;
}

xv_status xv_calibrate_table(xv_ctx* c, const char* feature_rspecifier, int32_t chunk_size, int32_t min_chunk_size, int32_t pad_input,
                             int32_t max_utts, float tol, xv_calibration* out) {
  if (!c || !feature_rspecifier) return Fail(XV_ERR_ARG, "xv_calibrate_table: null argument");
  return Guard([&] {
    xv::ExtractOptions opt;
    opt.chunk_size = chunk_size;
    opt.min_chunk_size = min_chunk_size;
    opt.pad_input = pad_input != 0;
    opt.calibrate_tol = tol;
    if (max_utts > 0) opt.calibrate_utts = max_utts;
    if (getenv("XVEC_CMN_WINDOW")) opt.cmn_window = atoi(getenv("XVEC_CMN_WINDOW"));
    if (getenv("XVEC_VAD_RSPECIFIER")) opt.vad_rspecifier = getenv("XVEC_VAD_RSPECIFIER");
    FillCalibration(xv::CalibrateOnTable(c->eng.get(), opt, feature_rspecifier,
                                         [](const char* level, const std::string& m) {
                                           fprintf(stderr, "%s (xvec_hip:xv_calibrate_table) %s\n", level, m.c_str());
                                         }),
                    out);
    return XV_OK;
  });
}

xv_status xv_ctx_set_calibration(xv_ctx* c, int32_t enable, float tol) {
  if (!c) return Fail(XV_ERR_ARG, "xv_ctx_set_calibration: null context");
  c->calibrate = enable != 0;
  if (tol > 0.f) c->calibrate_tol = tol;
  return XV_OK;
}

xv_status xv_ctx_set_calibration_file(xv_ctx* c, const char* path) {
  if (!c) return Fail(XV_ERR_ARG, "xv_ctx_set_calibration_file: null context");
  c->calibration_file = path ? path : "";
  return XV_OK;
}

xv_status xv_recognize_feature_pipeline(const char* rspecifier, int32_t* found, char* feats, size_t feats_cap, char* vad, size_t vad_cap,
                                        int32_t* cmn_window, int32_t* min_cmn_window, int32_t* center) {
  if (!rspecifier || !found) return Fail(XV_ERR_ARG, "xv_recognize_feature_pipeline: null argument");
  return Guard([&] {
    xv::FusedPipeline p;
    *found = xv::RecognizeFeaturePipeline(rspecifier, &p) ? 1 : 0;
    if (*found) {
      if ((feats && p.feats_rspecifier.size() + 1 > feats_cap) || (vad && p.vad_rspecifier.size() + 1 > vad_cap))
        return Fail(XV_ERR_ARG, "xv_recognize_feature_pipeline: output buffer too small");
      if (feats) memcpy(feats, p.feats_rspecifier.c_str(), p.feats_rspecifier.size() + 1);
      if (vad) memcpy(vad, p.vad_rspecifier.c_str(), p.vad_rspecifier.size() + 1);
      if (cmn_window) *cmn_window = p.cmn_window;
      if (min_cmn_window) *min_cmn_window = p.min_cmn_window;
      if (center) *center = p.center ? 1 : 0;
    }
    return XV_OK;
  });
}

xv_status xv_calibration_file_read(const char* path, int32_t* found, uint64_t* model, int32_t* precision, uint64_t* lite_mask) {
  if (!path || !*path || !found) return Fail(XV_ERR_ARG, "xv_calibration_file_read: null argument");
  return Guard([&] {
    xv::SharedChoice sc;
    *found = xv::ReadCalibrationFile(path, &sc) ? 1 : 0;
    if (*found) {
      if (model) *model = sc.model;
      if (precision) *precision = sc.precision;
      if (lite_mask) *lite_mask = sc.lite_mask;
    }
    return XV_OK;
  });
}

xv_status xv_calibration_file_publish(const char* path, uint64_t model, int32_t precision, uint64_t lite_mask, float tol, const char* note,
                                      int32_t* published, uint64_t* adopted_model, int32_t* adopted_precision,
                                      uint64_t* adopted_lite_mask) {
  if (!path || !*path) return Fail(XV_ERR_ARG, "xv_calibration_file_publish: null argument");
  if (precision != XV_PREC_FP16MX && precision != XV_PREC_FP16MX2 && precision != XV_PREC_FP16X3)
    return Fail(XV_ERR_ARG, "xv_calibration_file_publish: the arithmetic must be one of XV_PREC_FP16MX, XV_PREC_FP16MX2, XV_PREC_FP16X3");
  if (lite_mask && precision != XV_PREC_FP16MX2)
    return Fail(XV_ERR_ARG, "xv_calibration_file_publish: lite layers go with XV_PREC_FP16MX2 only");
  return Guard([&] {
    xv::SharedChoice mine, got;
    mine.model = model;
    mine.precision = precision;
    mine.lite_mask = lite_mask;
    mine.tolerance = tol;
    mine.note = note ? note : "";
    for (char& ch : mine.note)
      if (ch == '\n' || ch == '\r') ch = ' ';
    const bool won = xv::PublishCalibrationFile(path, mine, &got);
    if (published) *published = won ? 1 : 0;
    if (adopted_model) *adopted_model = got.model;
    if (adopted_precision) *adopted_precision = got.precision;
    if (adopted_lite_mask) *adopted_lite_mask = got.lite_mask;
    return XV_OK;
  });
}

xv_status xv_ctx_model_fingerprint(const xv_ctx* c, uint64_t* fingerprint) {
  if (!c || !fingerprint) return Fail(XV_ERR_ARG, "xv_ctx_model_fingerprint: null argument");
  *fingerprint = c->eng->info().fingerprint;
  return XV_OK;
}

xv_status xv_ctx_share_calibration(xv_ctx* c, const char* path, float tol, const char* note, int32_t* outcome) {
  if (!c || !path || !*path) return Fail(XV_ERR_ARG, "xv_ctx_share_calibration: null argument");
  return Guard([&] {
    xv::SharedChoice sc;
    int how = 0;
    if (!xv::ReadCalibrationFile(path, &sc)) {
      xv::SharedChoice mine;
      mine.model = c->eng->info().fingerprint;
      mine.precision = c->eng->fast_mode();
      mine.lite_mask = c->eng->lite_mask();
      mine.tolerance = tol;
      mine.note = note ? note : "";
      for (char& ch : mine.note)
        if (ch == '\n' || ch == '\r') ch = ' ';
      how = xv::PublishCalibrationFile(path, mine, &sc) ? 1 : 2;
    }
    xv::AdoptSharedChoice(c->eng.get(), sc, path);
    if (outcome) *outcome = how;
    return XV_OK;
  });
}

xv_status xv_ctx_set_profiling(xv_ctx* c, int32_t enable) {
  if (!c) return Fail(XV_ERR_ARG, "xv_ctx_set_profiling: null context");
  c->eng->SetProfiling(enable != 0);
  return XV_OK;
}

size_t xv_ctx_profile_report(xv_ctx* c, char* buf, size_t n) {
  if (!c) return 0;
  std::string s;
  if (Guard([&] {
        s = c->eng->ProfileReport();
        return XV_OK;
      }) != XV_OK)
    return 0;
  if (buf && n) {
    const size_t k = s.size() < n - 1 ? s.size() : n - 1;
    memcpy(buf, s.data(), k);
    buf[k] = 0;
  }
  return s.size() + 1;
}

xv_status xv_extract_utterances(xv_ctx* c, const float* feats, const int32_t* row_offsets, int32_t n_utts,
                                int32_t chunk_size, int32_t min_chunk_size, int32_t pad_input, float* out, int32_t* ok) {
  if (!c || !feats || !row_offsets || !out || !ok) return Fail(XV_ERR_ARG, "xv_extract_utterances: null argument");
  return Guard([&] {
    xv::ExtractOptions opt;
    opt.chunk_size = chunk_size;
    opt.min_chunk_size = min_chunk_size;
    opt.pad_input = pad_input != 0;
    xv::ExtractUtterances(c->eng.get(), opt, feats, row_offsets, n_utts, out, ok, nullptr);
    return XV_OK;
  });
}

xv_status xv_extract_table(xv_ctx* c, const char* feature_rspecifier, const char* vector_wspecifier, int32_t chunk_size,
                           int32_t min_chunk_size, int32_t pad_input, int32_t batch_frames, int64_t* num_done,
                           int64_t* num_failed) {
  if (!c || !feature_rspecifier || !vector_wspecifier) return Fail(XV_ERR_ARG, "xv_extract_table: null argument");
  return Guard([&] {
    xv::ExtractOptions opt;
    opt.chunk_size = chunk_size;
    opt.min_chunk_size = min_chunk_size;
    opt.pad_input = pad_input != 0;
    if (batch_frames > 0) opt.max_batch_rows = batch_frames;
    opt.calibrate = c->calibrate;
    opt.calibrate_tol = c->calibrate_tol;
    opt.calibration_file = c->calibration_file;
    if (getenv("XVEC_CMN_WINDOW")) opt.cmn_window = atoi(getenv("XVEC_CMN_WINDOW"));
    if (getenv("XVEC_VAD_RSPECIFIER")) opt.vad_rspecifier = getenv("XVEC_VAD_RSPECIFIER");
    xv::TableExtractResult r = xv::RunTableExtraction(
        c->eng.get(), opt, feature_rspecifier, vector_wspecifier, [](const char* level, const std::string& m) {
          fprintf(stderr, "%s (xvec_hip:xv_extract_table) %s\n", level, m.c_str());
        });
    if (num_done) *num_done = r.num_success;
    if (num_failed) *num_failed = r.num_fail;
    return XV_OK;
  });
}

xv_status xv_frontend_cmvn_select(xv_ctx* c, const float* raw, const int32_t* raw_off, int32_t n_utts, const float* vad,
                                  int32_t cmn_window, int32_t center, float* out, int32_t* out_off) {
  if (!c || !raw || !raw_off || !out || !out_off || n_utts < 0) return Fail(XV_ERR_ARG, "xv_frontend_cmvn_select: bad argument");
  return Guard([&] {
    std::vector<int32_t> sel_row, sel_utt;
    out_off[0] = 0;
    for (int u = 0; u < n_utts; ++u) {
      for (int32_t r = raw_off[u]; r < raw_off[u + 1]; ++r)
        if (!vad || vad[r] != 0.f) {
          sel_row.push_back(r);
          sel_utt.push_back(u);
        }
      out_off[u + 1] = (int32_t)sel_row.size();
    }
    c->eng->FrontEndHost(raw, raw_off, n_utts, sel_row.data(), sel_utt.data(), (int)sel_row.size(), cmn_window, center != 0,
                         100, out);
    return XV_OK;
  });
}

xv_status xv_plan_chunks(int32_t num_rows, int32_t chunk_size, int32_t min_chunk_size, int32_t pad_input,
                         int32_t min_net_frames, int32_t cap, int32_t* start, int32_t* len, int32_t* left_pad,
                         int32_t* right_pad, int32_t* n_chunks) {
  if (!n_chunks) return Fail(XV_ERR_ARG, "xv_plan_chunks: null argument");
  return Guard([&] {
    std::vector<xv::Chunk> ch;
    std::string why;
    *n_chunks = 0;
    if (!xv::PlanChunks(0, num_rows, chunk_size, min_chunk_size, pad_input != 0, min_net_frames, &ch, &why))
      return Fail(XV_ERR_ARG, why);
    *n_chunks = (int32_t)ch.size();
    for (int i = 0; i < (int)ch.size() && i < cap; ++i) {
      if (start) start[i] = ch[i].start;
      if (len) len[i] = ch[i].len;
      if (left_pad) left_pad[i] = ch[i].left_pad;
      if (right_pad) right_pad[i] = ch[i].right_pad;
    }
    return XV_OK;
  });
}

xv_status xv_backend_apply(int device, const float* x, int32_t n, int32_t dim, const float* mean, const float* transform,
                           int32_t t_rows, int32_t t_cols, int32_t normalize, int32_t scaleup, float* out, float* ratio) {
  if (n < 0 || dim < 1 || (n > 0 && (!x || !out)) || (transform && t_rows < 1))
    return Fail(XV_ERR_ARG, "xv_backend_apply: bad argument");
  if (transform && t_cols != dim && t_cols != dim + 1)
    return Fail(XV_ERR_ARG, "Dimension mismatch: input vector has dimension " + std::to_string(dim) + " and transform has " +
                                std::to_string(t_cols) + " columns");
  return Guard([&] {
    xv::BackendOptions o;
    o.mean = mean;
    o.transform = transform;
    o.t_rows = t_rows;
    o.t_cols = t_cols;
    o.normalize = normalize != 0;
    o.scaleup = scaleup != 0;
    xv::BackendApply(device, x, n, dim, o, out, ratio);
    return XV_OK;
  });
}

xv_status xv_segment_mean(int device, const float* x, int32_t n, int32_t dim, const int32_t* seg_off, const int32_t* idx,
                          int32_t n_seg, int32_t acc64, float* out) {
  if (n < 0 || dim < 1 || n_seg < 0 || (n_seg > 0 && (!x || !seg_off || !out)))
    return Fail(XV_ERR_ARG, "xv_segment_mean: bad argument");
  for (int s = 0; s < n_seg; ++s)
    if (seg_off[s + 1] < seg_off[s] || seg_off[s] < 0) return Fail(XV_ERR_ARG, "xv_segment_mean: segment offsets must not decrease");
  if (n_seg > 0 && seg_off[n_seg] > 0 && !idx) return Fail(XV_ERR_ARG, "xv_segment_mean: null index list");
  return Guard([&] {
    xv::SegmentMean(device, x, n, dim, seg_off, idx, n_seg, acc64 != 0, out);
    return XV_OK;
  });
}

// One process, several GPUs: multi_gpu.cc (one ncclBroadcast of the packed image, bounded wait, contexts from the device copies).
xv_status xv_ctx_create_broadcast(const xv_model* m, const int* devices, int n, int precision, xv_ctx** out) {
  if (!m || !devices || !out || n < 1) return Fail(XV_ERR_ARG, "xv_ctx_create_broadcast: bad argument");
  return Guard([&]() -> xv_status {
    const std::vector<uint8_t> blob = xv::PackModelPolicy(m->prog, precision);
    std::vector<std::unique_ptr<xv::Engine>> engines = xv::CreateEnginesBroadcast(blob, std::vector<int>(devices, devices + n));
    for (int i = 0; i < n; ++i) {
      xv_ctx* c = new xv_ctx;
      c->eng = std::move(engines[i]);
      out[i] = c;
    }
    return XV_OK;
  });
}

xv_status xv_kernel_tdnn_gemm(const xv_gemm_desc* d) {
  if (!d) return Fail(XV_ERR_ARG, "xv_kernel_tdnn_gemm: null descriptor");
  return Guard([&] {
    if (d->nseg < 1 || d->nseg > xv::kMaxSeg || d->rows % xv::kBM || d->n_pad % xv::kBN)
      return Fail(XV_ERR_ARG, "xv_kernel_tdnn_gemm: bad geometry");
    xv::GemmArgs a;
    memset(&a, 0, sizeof a);
    a.nseg = d->nseg;
    for (int j = 0; j < d->nseg; ++j) {
      if (d->seg[j].k_len % xv::kBK) return Fail(XV_ERR_ARG, "xv_kernel_tdnn_gemm: k_len must be a multiple of 32");
      a.seg[j].hi = (const uint16_t*)d->seg[j].hi;
      a.seg[j].lo = (const uint16_t*)d->seg[j].lo;
      a.seg[j].ld = d->seg[j].ld;
      a.seg[j].row_shift = d->seg[j].row_shift;
      a.seg[j].ksteps = d->seg[j].k_len / xv::kBK;
      a.seg[j].gmax = (const unsigned*)d->seg[j].gmax;
      a.seg[j].lo4 = (const uint8_t*)d->seg[j].lo4;
      a.seg[j].lo4s = (const uint8_t*)d->seg[j].lo4_scale;
      a.total_ksteps += a.seg[j].ksteps;
    }
    a.w_hi = (const uint16_t*)d->w_hi;
    a.w_lo = (const uint16_t*)d->w_lo;
    a.ldw = d->ldw;
    a.m_tiles = d->rows / xv::kBM;
    a.n_tiles = d->n_pad / xv::kBN;
    a.relu = d->relu;
    a.bn = d->bn;
    a.bias = d->bias;
    a.scale = d->scale;
    a.offset = d->offset;
    a.out_hi = (uint16_t*)d->out_hi;
    a.out_lo = (uint16_t*)d->out_lo;
    a.ldo = d->ldo;
    a.out_f32 = d->out_f32;
    a.ldf = d->ldf;
    a.m_valid = d->m_valid;
    a.partial = d->partial;
    a.ldp = d->ldp;
    a.grp_range = d->grp_range;
    a.w4 = (const uint8_t*)d->w4;
    a.ldw4 = d->ldw4;
    a.w4_scale = (const uint8_t*)d->w4_scale;
    a.gmax_out = (unsigned*)d->gmax_out;
    a.w4b = (const uint8_t*)d->w4b;
    a.ldw4b = d->ldw4b;
    a.w4b_scale = (const uint8_t*)d->w4b_scale;
    a.out_lo4 = (uint8_t*)d->out_lo4;
    a.out_lo4s = (uint8_t*)d->out_lo4_scale;
    a.p8 = d->p8;
    if (a.p8) {
      if (!xv::gemm_p8_applicable(a, d->precision) || (d->epilogue != xv::kEpiAct && d->epilogue != xv::kEpiStats))
        return Fail(XV_ERR_ARG, "xv_kernel_tdnn_gemm: p8 needs XV_PREC_FP16, XV_PREC_FP16MX or XV_PREC_FP16MX2, epilogue 0 or 2, rows and n_pad "
                                "multiples of 256, K groups of whole 64-column tiles (128-column blocks for XV_PREC_FP16MX, 256 for XV_PREC_FP16MX2)");
    } else
    if (d->precision == xv::kPrecFp16Mx2 && !xv::gemm_mx2_applicable(a))
      return Fail(XV_ERR_ARG, "xv_kernel_tdnn_gemm: XV_PREC_FP16MX2 needs what XV_PREC_FP16MX needs and the 4-bit planes of the "
                              "weights and of every source (whole 128-column steps)");
    if (!a.p8 && d->precision == xv::kPrecFp16Mx && !xv::gemm_mx_applicable(a))
      return Fail(XV_ERR_ARG, "xv_kernel_tdnn_gemm: XV_PREC_FP16MX needs the residual plane, a group-max table per source, "
                              "K groups of whole 128-column blocks and an even number of 128-row tiles");
    hipError_t e = xv::launch_tdnn_gemm(a, d->precision, d->epilogue, (hipStream_t)d->hip_stream);
    if (e != hipSuccess) return Fail(XV_ERR_DEVICE, std::string("tdnn_gemm launch: ") + hipGetErrorString(e));
    return XV_OK;
  });
}

static xv_status PackMxResidualImpl(bool walk64, const float* w, const uint16_t* w_hi_f16, int32_t n_pad, int32_t nseg,
                                    const int32_t* seg_src, const int32_t* seg_shift, const int32_t* seg_klen, uint8_t* w4,
                                    uint8_t* w4_scale) {
  if (!w || !w_hi_f16 || !seg_src || !seg_shift || !seg_klen || !w4 || !w4_scale || nseg < 1 || nseg > xv::kMaxSeg || n_pad < 1)
    return Fail(XV_ERR_ARG, "xv_pack_mx_residual: bad argument");
  return Guard([&] {
    long key[xv::kMaxSeg];
    int shift[xv::kMaxSeg], ksteps[xv::kMaxSeg], k_pad = 0;
    for (int j = 0; j < nseg; ++j) {
      if (seg_klen[j] % xv::kBK) return Fail(XV_ERR_ARG, "xv_pack_mx_residual: k_len must be a multiple of 32");
      key[j] = seg_src[j];
      shift[j] = seg_shift[j];
      ksteps[j] = seg_klen[j] / xv::kBK;
      k_pad += seg_klen[j];
    }
    xv::WalkGroup wg[xv::kMaxSeg];
    const int ng = xv::PlanWalkGroups(nseg, key, shift, ksteps, wg);
    std::vector<int> step_wcol(k_pad / xv::kBK);
    bool ok = false;
    if (walk64) xv::PlanWalkSteps64(ng, wg, step_wcol.data(), (int)step_wcol.size(), &ok);
    else xv::PlanWalkSteps(ng, wg, step_wcol.data(), (int)step_wcol.size(), &ok);
    if (!ok) return Fail(XV_ERR_ARG, "xv_pack_mx_residual: a K group is not a multiple of four steps");
    const int ldw4 = k_pad / xv::kBK / 4 * 64;
    std::vector<float> res(k_pad);
    for (int n = 0; n < n_pad; ++n) {
      for (int k = 0; k < k_pad; ++k) res[k] = w[(size_t)n * k_pad + k] - xv::host_f16_to_f32(w_hi_f16[(size_t)n * k_pad + k]);
      xv::PackMxRow(res.data(), k_pad, step_wcol.data(), w4 + (size_t)n * ldw4, w4_scale + (size_t)n * (k_pad / xv::kBK));
    }
    return XV_OK;
  });
}

xv_status xv_pack_mx_residual(const float* w, const uint16_t* w_hi_f16, int32_t n_pad, int32_t nseg, const int32_t* seg_src,
                              const int32_t* seg_shift, const int32_t* seg_klen, uint8_t* w4, uint8_t* w4_scale) {
  return PackMxResidualImpl(false, w, w_hi_f16, n_pad, nseg, seg_src, seg_shift, seg_klen, w4, w4_scale);
}

xv_status xv_pack_mx_residual64(const float* w, const uint16_t* w_hi_f16, int32_t n_pad, int32_t nseg, const int32_t* seg_src,
                                const int32_t* seg_shift, const int32_t* seg_klen, uint8_t* w4, uint8_t* w4_scale) {
  return PackMxResidualImpl(true, w, w_hi_f16, n_pad, nseg, seg_src, seg_shift, seg_klen, w4, w4_scale);
}

static xv_status PackMxWeightsImpl(bool walk64, const float* w, int32_t n_pad, int32_t nseg, const int32_t* seg_src,
                                   const int32_t* seg_shift, const int32_t* seg_klen, uint8_t* w4b, uint8_t* w4b_scale) {
  if (!w || !seg_src || !seg_shift || !seg_klen || !w4b || !w4b_scale || nseg < 1 || nseg > xv::kMaxSeg || n_pad < 1)
    return Fail(XV_ERR_ARG, "xv_pack_mx_weights: bad argument");
  return Guard([&] {
    long key[xv::kMaxSeg];
    int shift[xv::kMaxSeg], ksteps[xv::kMaxSeg], k_pad = 0;
    for (int j = 0; j < nseg; ++j) {
      if (seg_klen[j] % (walk64 ? 256 : 128)) return Fail(XV_ERR_ARG, "xv_pack_mx_weights: k_len must be a multiple of 128 (256 for the p8 order)");
      key[j] = seg_src[j];
      shift[j] = seg_shift[j];
      ksteps[j] = seg_klen[j] / xv::kBK;
      k_pad += seg_klen[j];
    }
    xv::WalkGroup wg[xv::kMaxSeg];
    const int ng = xv::PlanWalkGroups(nseg, key, shift, ksteps, wg);
    std::vector<int> lo_wcol(k_pad / 128);
    const int n_lo = walk64 ? xv::PlanWalkLoSteps64(ng, wg, lo_wcol.data(), (int)lo_wcol.size())
                            : xv::PlanWalkLoSteps(ng, wg, lo_wcol.data(), (int)lo_wcol.size());
    if (n_lo != (int)lo_wcol.size()) return Fail(XV_ERR_ARG, "xv_pack_mx_weights: inconsistent walk");
    const size_t pitch = walk64 ? (size_t)k_pad * 2 : (size_t)n_lo * 64;   // p8: rows of the fp16 plane's pitch
    if (walk64) memset(w4b, 0, pitch * n_pad);
    for (int n = 0; n < n_pad; ++n)
      xv::PackMxWeightsRow(w + (size_t)n * k_pad, lo_wcol.data(), n_lo, w4b + (size_t)n * pitch, w4b_scale + (size_t)n * n_lo * 4);
    return XV_OK;
  });
}

xv_status xv_pack_mx_weights(const float* w, int32_t n_pad, int32_t nseg, const int32_t* seg_src, const int32_t* seg_shift,
                             const int32_t* seg_klen, uint8_t* w4b, uint8_t* w4b_scale) {
  return PackMxWeightsImpl(false, w, n_pad, nseg, seg_src, seg_shift, seg_klen, w4b, w4b_scale);
}

xv_status xv_pack_mx_weights64(const float* w, int32_t n_pad, int32_t nseg, const int32_t* seg_src, const int32_t* seg_shift,
                               const int32_t* seg_klen, uint8_t* w4b, uint8_t* w4b_scale) {
  return PackMxWeightsImpl(true, w, n_pad, nseg, seg_src, seg_shift, seg_klen, w4b, w4b_scale);
}

xv_status xv_tile_mx_scales(const uint8_t* natural, int32_t n_pad, int32_t k_len, int32_t epilogue, uint8_t* tiled) {
  if (!natural || !tiled || n_pad < 1 || n_pad % xv::kBN || k_len < 128 || k_len % 128)
    return Fail(XV_ERR_ARG, "xv_tile_mx_scales: bad argument");
  return Guard([&] {
    xv::TileMxScales(natural, n_pad, k_len / xv::kBK, epilogue != xv::kEpiStats, tiled);
    return XV_OK;
  });
}

}  // extern "C"
