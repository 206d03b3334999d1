// See fuse_pipe.h.
#include "fuse_pipe.h"

#include <stdlib.h>

#include <sstream>
#include <vector>

namespace xv {

namespace {

std::string Basename(const std::string& p) {
  const size_t s = p.rfind('/');
  return s == std::string::npos ? p : p.substr(s + 1);
}

bool ParseBoolValue(const std::string& v, bool* out) {
  if (v == "true" || v == "t" || v == "1") {
    *out = true;
    return true;
  }
  if (v == "false" || v == "f" || v == "0") {
    *out = false;
    return true;
  }
  return false;
}

bool ParseIntValue(const std::string& v, int* out) {
  if (v.empty()) return false;
  char* end = nullptr;
  const long x = strtol(v.c_str(), &end, 10);
  if (*end || x < 1 || x > 1000000) return false;
  *out = (int)x;
  return true;
}

std::vector<std::string> Words(const std::string& s) {
  std::istringstream in(s);
  std::vector<std::string> w;
  std::string t;
  while (in >> t) w.push_back(t);
  return w;
}

}  // namespace

bool RecognizeFeaturePipeline(const std::string& rspecifier, FusedPipeline* out) {
  // anything a shell would interpret beyond words and the pipe disqualifies the string: it is then a command line, not this pipeline
  for (char c : rspecifier)
    if (c == '\'' || c == '"' || c == '\\' || c == '`' || c == '$' || c == ';' || c == '&' || c == '<' || c == '>' || c == '(' ||
        c == ')' || c == '*' || c == '?' || c == '~' || c == '\n')
      return false;
  std::string s = rspecifier;
  while (!s.empty() && (s.back() == ' ' || s.back() == '\t')) s.pop_back();
  if (s.compare(0, 4, "ark:") != 0 || s.empty() || s.back() != '|') return false;
  s = s.substr(4, s.size() - 5);
  std::vector<std::string> stages;
  {
    size_t a = 0;
    for (;;) {
      const size_t b = s.find('|', a);
      stages.push_back(s.substr(a, b == std::string::npos ? std::string::npos : b - a));
      if (b == std::string::npos) break;
      a = b + 1;
    }
  }
  if (stages.size() < 1 || stages.size() > 2) return false;
  FusedPipeline p;
  {
    const std::vector<std::string> w = Words(stages[0]);
    if (w.size() < 3 || Basename(w[0]) != "apply-cmvn-sliding") return false;
    std::vector<std::string> pos;
    for (size_t i = 1; i < w.size(); ++i) {
      if (w[i].compare(0, 2, "--") != 0) {
        pos.push_back(w[i]);
        continue;
      }
      const size_t eq = w[i].find('=');
      if (eq == std::string::npos) return false;
      std::string name = w[i].substr(2, eq - 2);
      for (char& c : name)
        if (c == '_') c = '-';
      const std::string value = w[i].substr(eq + 1);
      bool b = false;
      if (name == "norm-vars") {
        if (!ParseBoolValue(value, &b) || b) return false;        // variance normalisation: not what the device front-end does
      } else if (name == "center") {
        if (!ParseBoolValue(value, &p.center)) return false;
      } else if (name == "cmn-window") {
        if (!ParseIntValue(value, &p.cmn_window)) return false;
      } else if (name == "min-cmn-window") {
        if (!ParseIntValue(value, &p.min_cmn_window)) return false;
      } else {
        return false;                                             // --max-warnings, --verbose, --config ...: run the command
      }
    }
    if (pos.size() != 2 || pos[1] != "ark:-") return false;
    const std::string& in = pos[0];
    // a table in a file: "scp:<file>" or "ark:<file>", with the option letters Kaldi allows in front of the colon
    const size_t colon = in.find(':');
    if (colon == std::string::npos || colon + 1 >= in.size()) return false;
    const std::string kind = in.substr(0, in.find_first_of(",:"));
    if (kind != "scp" && kind != "ark") return false;
    const std::string target = in.substr(colon + 1);
    if (target == "-" || target.back() == '|') return false;
    p.feats_rspecifier = in;
  }
  if (stages.size() == 2) {
    const std::vector<std::string> w = Words(stages[1]);
    if (w.size() != 4 || Basename(w[0]) != "select-voiced-frames" || w[1] != "ark:-" || w[3] != "ark:-") return false;
    const std::string& v = w[2];
    if (v.compare(0, 2, "--") == 0) return false;
    const size_t colon = v.find(':');
    if (colon == std::string::npos || colon + 1 >= v.size()) return false;
    const std::string kind = v.substr(0, v.find_first_of(",:"));
    if (kind != "scp" && kind != "ark") return false;
    if (v.substr(colon + 1) == "-" || v.back() == '|') return false;
    p.vad_rspecifier = v;
  }
  *out = p;
  return true;
}

}  // namespace xv
