// nnet3 `.raw` model container: token-driven reader (binary + text) and writer.
//
// Replaces `ReadKaldiObject(rxfilename, &nnet)` of nnet3-xvector-compute for the models the reference
// defines (graphs: egs/sre/v2/local/nnet3/xvector/run_xvector_new.sh:94-114,
// egs/sre/v5/local/nnet3_cvector/cvector/train_am.sh:30-38, train_cvector_with_am.sh:65-89 ...) and
// the `nnet3-copy --nnet-config=extract.config final.raw -` edit that
// egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:58-59 puts in front of it.
// Layout: SURVEY.md App. B.3 (upstream Kaldi, not vendored).  The reader keeps only what the forward
// pass needs plus the raw byte span of every component so that the model can be re-emitted verbatim.
#pragma once
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "kio.h"

namespace xv {

struct RawComponent {
  std::string name;
  std::string type;                      // e.g. NaturalGradientAffineComponent
  std::map<std::string, double> scalar;  // every scalar field by token name, e.g. "<Dim>"
  Matrix linear;                         // <LinearParams> / <Params>
  std::vector<float> bias;               // <BiasParams>
  std::vector<float> stats_mean, stats_var;  // BatchNorm
  bool has_linear = false;
  size_t span_begin = 0, span_end = 0;   // byte span of "<Type> ... </Type>" in the source buffer
  double Get(const char* tok, double dflt) const {
    auto it = scalar.find(tok);
    return it == scalar.end() ? dflt : it->second;
  }
};

struct RawNode {
  std::string kind;       // input-node | component-node | output-node | dim-range-node
  std::string name;
  std::string component;  // component-node
  std::string input;      // descriptor text (component-node / output-node)
  int dim = 0;            // input-node
  std::string line;       // the config line as read
};

struct DescTerm {  // one Append() term after flattening
  std::string node;
  int offset = 0;
};

// Flattens  name | Offset(d,k) | Append(d,...) | Round(d,k)  into Append order terms.
// Throws KioError for operators outside that subset (Sum, Scale, IfDefined, ...).
std::vector<DescTerm> FlattenDescriptor(const std::string& text);

class RawNnet {
 public:
  std::vector<RawNode> nodes;
  std::vector<RawComponent> components;
  bool binary = true;

  // Parses a whole model held in memory.
  void Read(const std::string& bytes);
  // Reads the object found at an rxfilename (file, file:offset, -, "cmd |").
  void ReadFrom(const std::string& rxfilename);
  // `nnet3-copy --nnet-config=<text>` for node lines: same-kind same-name nodes are replaced, new ones
  // appended (what extract.config "output-node name=output input=tdnn6.affine" needs).
  void ApplyNnetConfig(const std::string& config_text);
  // Re-emits the model (components verbatim when the flavour is unchanged).
  void Write(Output& out, bool binary_out) const;

  const RawNode* FindNode(const std::string& name, const char* kind = nullptr) const;
  const RawComponent* FindComponent(const std::string& name) const;

 private:
  std::string source_;  // kept for verbatim component spans
};

RawNode ParseConfigLine(const std::string& line);  // kind empty for blank / comment lines

}  // namespace xv
