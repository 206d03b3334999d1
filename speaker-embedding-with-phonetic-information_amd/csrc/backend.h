// Speaker-level back-end of the extraction path: the vector post-processing the reference
// runs right after nnet3-xvector-compute — ivector-mean (egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:106-107,
// egs/sre/v2/run_sre10.sh:219-221), ivector-subtract-global-mean | transform-vec | ivector-normalize-length
// (run_sre10.sh:229,233,238-241) — as two device kernels (kernels.h: launch_backend, launch_segment_mean) behind host
// buffers.  Everything here throws EngineError when there is no usable GPU: there is no CPU path.
#pragma once
#include <stdint.h>

#include <vector>

namespace xv {

struct BackendOptions {
  const float* mean = nullptr;       // [dim] subtracted first, or null
  const float* transform = nullptr;  // [t_rows][t_cols], t_cols == dim (linear) or dim + 1 (affine), or null
  int t_rows = 0, t_cols = 0;
  bool normalize = false;            // ivector-normalize-length --normalize
  bool scaleup = true;               // ivector-normalize-length --scaleup
};

// out: [n][out_dim], out_dim = transform ? t_rows : dim.  ratio (optional, [n]): |y| / sqrt(out_dim) (scaleup) or |y|.
void BackendApply(int device, const float* x, int n, int dim, const BackendOptions& opt, float* out, float* ratio);

// out[s] = mean of rows idx[seg_off[s] .. seg_off[s+1]) of x ([n][dim]); fp32 or fp64 accumulation in list order.
void SegmentMean(int device, const float* x, int n, int dim, const int32_t* seg_off, const int32_t* idx, int n_seg,
                 bool acc64, float* out);

}  // namespace xv
