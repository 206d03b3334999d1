#include "backend.h"

#include <hip/hip_runtime.h>

#include <string>

#include "engine.h"
#include "kernels.h"

namespace xv {
namespace {

void Check(hipError_t e, const char* what) {
  if (e != hipSuccess) throw EngineError(std::string(what) + ": " + hipGetErrorString(e));
}

// Device buffer that frees itself.
struct DevBuf {
  void* p = nullptr;
  explicit DevBuf(size_t n) { Check(hipMalloc(&p, n ? n : 4), "hipMalloc"); }
  ~DevBuf() { if (p) (void)hipFree(p); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

void UseDevice(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1)
    throw EngineError("no HIP device available: the back-end kernels need a gfx950 GPU (there is no CPU path)");
  if (device < 0 || device >= n) throw EngineError("device index out of range");
  Check(hipSetDevice(device), "hipSetDevice");
}

}  // namespace

void BackendApply(int device, const float* x, int n, int dim, const BackendOptions& opt, float* out, float* ratio) {
  if (n < 0 || dim < 1) throw EngineError("BackendApply: bad shape");
  if (opt.transform && opt.t_cols != dim && opt.t_cols != dim + 1)
    throw EngineError("Dimension mismatch: input vector has dimension " + std::to_string(dim) + " and transform has " +
                      std::to_string(opt.t_cols) + " columns");
  UseDevice(device);
  if (n == 0) return;
  const int out_dim = opt.transform ? opt.t_rows : dim;
  DevBuf dx((size_t)n * dim * 4), dout((size_t)n * out_dim * 4), dratio((size_t)n * 4);
  DevBuf dmean((size_t)dim * 4), dt(opt.transform ? (size_t)opt.t_rows * opt.t_cols * 4 : 4);
  Check(hipMemcpy(dx.p, x, (size_t)n * dim * 4, hipMemcpyHostToDevice), "copy vectors");
  if (opt.mean) Check(hipMemcpy(dmean.p, opt.mean, (size_t)dim * 4, hipMemcpyHostToDevice), "copy mean");
  if (opt.transform) Check(hipMemcpy(dt.p, opt.transform, (size_t)opt.t_rows * opt.t_cols * 4, hipMemcpyHostToDevice), "copy transform");
  BackendArgs a;
  a.x = (const float*)dx.p;
  a.n = n;
  a.dim = dim;
  a.ldx = dim;
  a.mean = opt.mean ? (const float*)dmean.p : nullptr;
  a.t = opt.transform ? (const float*)dt.p : nullptr;
  a.t_rows = opt.t_rows;
  a.t_cols = opt.t_cols;
  a.normalize = opt.normalize ? 1 : 0;
  a.scaleup = opt.scaleup ? 1 : 0;
  a.out = (float*)dout.p;
  a.ldo = out_dim;
  a.ratio = ratio ? (float*)dratio.p : nullptr;
  Check(launch_backend(a, nullptr), "backend kernel launch");
  Check(hipMemcpy(out, dout.p, (size_t)n * out_dim * 4, hipMemcpyDeviceToHost), "copy result");
  if (ratio) Check(hipMemcpy(ratio, dratio.p, (size_t)n * 4, hipMemcpyDeviceToHost), "copy ratios");
}

void SegmentMean(int device, const float* x, int n, int dim, const int32_t* seg_off, const int32_t* idx, int n_seg,
                 bool acc64, float* out) {
  if (n < 0 || dim < 1 || n_seg < 0) throw EngineError("SegmentMean: bad shape");
  UseDevice(device);
  if (n_seg == 0) return;
  const int n_idx = seg_off[n_seg];
  for (int i = 0; i < n_idx; ++i)
    if (idx[i] < 0 || idx[i] >= n) throw EngineError("SegmentMean: row index out of range");
  DevBuf dx((size_t)n * dim * 4), doff((size_t)(n_seg + 1) * 4), didx((size_t)n_idx * 4), dout((size_t)n_seg * dim * 4);
  Check(hipMemcpy(dx.p, x, (size_t)n * dim * 4, hipMemcpyHostToDevice), "copy vectors");
  Check(hipMemcpy(doff.p, seg_off, (size_t)(n_seg + 1) * 4, hipMemcpyHostToDevice), "copy segment offsets");
  if (n_idx) Check(hipMemcpy(didx.p, idx, (size_t)n_idx * 4, hipMemcpyHostToDevice), "copy row indices");
  SegMeanArgs a;
  a.x = (const float*)dx.p;
  a.dim = dim;
  a.ldx = dim;
  a.seg_off = (const int32_t*)doff.p;
  a.idx = (const int32_t*)didx.p;
  a.n_seg = n_seg;
  a.acc64 = acc64 ? 1 : 0;
  a.out = (float*)dout.p;
  Check(launch_segment_mean(a, nullptr), "segment mean kernel launch");
  Check(hipMemcpy(out, dout.p, (size_t)n_seg * dim * 4, hipMemcpyDeviceToHost), "copy result");
}

}  // namespace xv
