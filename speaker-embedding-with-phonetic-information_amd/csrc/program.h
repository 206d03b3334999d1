// Graph lowering: nnet3 node graph -> "TDNN program".
//
// Replaces, for the graphs the reference defines, what nnet3-xvector-compute does between reading the
// model and running it: SetBatchnormTestMode(true) / SetDropoutTestMode(true) / CollapseModel() and
// the ComputationRequest -> CachingOptimizingCompiler::Compile step that decides which frames of
// which node are needed (App. B.4).  All topologies in the tree
// (run_xvector_new.sh:94-114, train_am.sh:30-38, train_xvector_with_am.sh:43-56,
// train_cvector_with_am.sh:65-89, prepare_nnet3_xconfig*.sh, run_xvector_pa_wo_pretrain.sh:94-122)
// fit one grammar: spliced affine [-> ReLU] [-> BatchNorm] layers over one or two source nodes, one
// mean+stddev pooling over every computable frame, affine layers after the pooling.
#pragma once
#include <string>
#include <vector>

#include "nnet3_raw.h"

namespace xv {

constexpr int kSrcInput = -1;   // source is the feature matrix
constexpr int kSrcPooled = -2;  // source is the [mean | stddev] vector of the pooling

struct LayerSource {
  int layer = kSrcInput;  // index into TdnnProgram::layers, or kSrcInput / kSrcPooled
  int offset = 0;         // time offset of this Append() term
  int dim = 0;
};

struct AffineLayer {
  std::string name;               // affine node name, e.g. "tdnn3.affine"
  std::string out_node;           // node whose value the layer materialises, e.g. "tdnn3.batchnorm"
  std::vector<LayerSource> src;   // Append() order == column blocks of w
  int in_dim = 0, out_dim = 0;
  std::vector<float> w;           // [out_dim][in_dim] row-major
  std::vector<float> bias;        // [out_dim]
  bool relu = false, bn = false;
  std::vector<float> bn_scale, bn_offset;  // test-mode BatchNorm: y = x*scale + offset
  bool bn_folded = false;         // FoldBatchNormIntoConsumers: bn_scale is a power of two, bn_offset 0; the rest is in the consumers' w / bias
  bool log_softmax = false;       // LogSoftmaxComponent after the affine (frame-level heads)
  bool segment_level = false;     // computed once per chunk (after the pooling)
  int left = 0, right = 0;        // frames not computable at the left / right edge (frame-level)
};

struct TdnnProgram {
  int input_dim = 0;
  std::vector<AffineLayer> layers;   // topological order
  int pooled_layer = -1;             // frame-level layer feeding the statistics pooling (-1: none)
  int pool_dim = 0;                  // its dimension; pooled vector is 2*pool_dim
  int pool_left = 0, pool_right = 0; // StatisticsPoolingComponent left/right context
  float variance_floor = 1e-10f;
  int output_layer = -1;
  int output_dim = 0;
  bool output_is_segment = true;     // one vector per chunk (x-vector) vs one row per frame
  int left_context = 0, right_context = 0;  // of the frame-level part that feeds the output
  int min_frames = 1;                // smallest chunk with >= 1 computable output
  std::string output_name;
  // Algorithmic multiply-accumulates of one chunk of T frames (unpadded dims, computable frames only)
  double Macs(int T) const;
  std::string Describe() const;
};

// Lowers the dependency cone of output node `output_name` ("output" in every recipe).  Throws KioError
// with a precise message when the graph is outside the supported grammar.
TdnnProgram LowerToProgram(const RawNnet& net, const std::string& output_name);
// Moves the BatchNorm of every frame-level relu + batchnorm layer that only feeds other layers into those consumers (program.cc);
// applied by LowerToProgram only when XVEC_DEBUG=bn_fold=1 (opt-in; measured in round 5: +3.4 % throughput, but the fast arithmetics'
// weight-side errors stop averaging out over the frames of a chunk - program.cc).
void FoldBatchNormIntoConsumers(TdnnProgram* p);

}  // namespace xv
