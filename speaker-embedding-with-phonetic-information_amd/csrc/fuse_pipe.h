// The feature pipeline every extraction script of the reference builds, recognised as text.
//
// egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:79 (and extract_xvectors.sh:73, extract_output_new.sh:75, the
// nnet3_cvector copies, extract_cvectors_with_am.sh:92, extract_cvectors_with_embedding.sh:81, extract_log_post.sh:68-70) hand
// the binary its features as ONE rspecifier string:
//   ark:apply-cmvn-sliding --norm-vars=false --center=true --cmn-window=300 scp:$sdata/feats.scp ark:- |
//       select-voiced-frames ark:- scp,s,cs:$sdata/vad.scp ark:- |
// Run as written, that is two CPU tools and two pipes per job in front of a GPU that could take fifty times what they deliver.
// The tool therefore reads the string before it starts the commands: when it is EXACTLY this pipeline - these two programs, only
// the options the device front-end implements with the values it implements them for, "ark:-" in the plumbing positions - the job
// reads the inner feature table itself and runs sliding CMN + voiced-frame selection on the device (engine.h FrontEndJob; the
// stored features, compressed as make_mfcc.sh leaves them, go up as they are).  Anything else - another option, another
// program, a third stage, quotes, redirections - is not this pipeline and is run as a command, as before.
#pragma once
#include <string>

namespace xv {

struct FusedPipeline {
  std::string feats_rspecifier;   // what apply-cmvn-sliding reads: "scp:..." or "ark:..."
  std::string vad_rspecifier;     // what select-voiced-frames takes its decisions from; empty: no selection stage
  int cmn_window = 600;           // apply-cmvn-sliding's defaults (sliding-window-cmn options)
  int min_cmn_window = 100;
  bool center = false;
};

// true: `rspecifier` is the pipeline above and *out describes it.  false: it is something else (out untouched).
bool RecognizeFeaturePipeline(const std::string& rspecifier, FusedPipeline* out);

}  // namespace xv
