// copy-feats / copy-vector - minimal equivalents of the Kaldi table-copy tools, built on kio.
//   copy-feats  [--binary=true|false] [--compress=...ignored] <matrix-rspecifier> <matrix-wspecifier>
//   copy-vector [--binary=true|false] <vector-rspecifier> <vector-wspecifier>
// They exist so that recipes and tests can move features / embeddings between ark, scp and text forms on a
// Kaldi-less box (the reference pipes features through such tools, extract_xvectors_new.sh:79), and they
// exercise every reader/writer path of kio (FM/DM/CM/CM2/CM3/text in; FM/FV/text, ark+scp out).
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "kio.h"

int main(int argc, char** argv) {
  const char* prog = strrchr(argv[0], '/') ? strrchr(argv[0], '/') + 1 : argv[0];
  xv::InstallMappedFileFaultHandler(prog);
  const bool vectors = strstr(prog, "vector") != nullptr;
  try {
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
      std::string a = argv[i];
      if (a.compare(0, 2, "--") == 0 && pos.empty()) continue;  // ark,t: / ark: in the wspecifier decides the form
      pos.push_back(a);
    }
    if (pos.size() != 2) {
      fprintf(stderr, "Usage: %s [options] <rspecifier> <wspecifier>\n", prog);
      return 1;
    }
    xv::TableWriter w(pos[1]);
    long n = 0, bad = 0;
    if (!vectors) {
      xv::SequentialMatrixReader r(pos[0]);
      std::string key, err;
      xv::Matrix m;
      while (r.Next(&key, &m, &err)) {
        if (!err.empty()) {
          fprintf(stderr, "WARNING (%s) %s: %s\n", prog, key.c_str(), err.c_str());
          ++bad;
          continue;
        }
        w.WriteMat(key, m);
        ++n;
      }
    } else {
      // vectors: a vector table is read through the matrix reader's text/binary object layer
      xv::RspecifierOptions o = xv::ParseRspecifier(pos[0]);
      if (o.is_scp) {
        xv::RandomAccessVectorReader rr(pos[0]);
        xv::Input in;
        in.Open(o.rxfilename);
        std::string line;
        int c;
        while ((c = in.Get()) >= 0) {
          if (c != '\n') {
            line.push_back((char)c);
            continue;
          }
          size_t sp = line.find_first_of(" \t");
          std::string key = line.substr(0, sp);
          if (!key.empty()) {
            const std::vector<float>& v = rr.Value(key);
            w.WriteVec(key, v.data(), (int)v.size());
            ++n;
          }
          line.clear();
        }
      } else {
        xv::Input in;
        in.Open(o.rxfilename);
        for (;;) {
          int c;
          while ((c = in.Peek()) >= 0 && isspace(c)) in.Get();
          if (c < 0) break;
          std::string key;
          while ((c = in.Peek()) >= 0 && !isspace(c)) key.push_back((char)in.Get());
          in.Get();
          bool binary = xv::ReadBinaryHeader(in);
          std::vector<float> v;
          xv::ReadVector(in, binary, &v);
          w.WriteVec(key, v.data(), (int)v.size());
          ++n;
        }
      }
    }
    w.Close();
    fprintf(stderr, "LOG (%s) Copied %ld %s%s\n", prog, n, vectors ? "vectors" : "feature matrices",
            bad ? " (some entries failed)" : "");
    return n > 0 ? 0 : 1;
  } catch (const std::exception& e) {
    fprintf(stderr, "ERROR (%s) %s\n", prog, e.what());
    return -1;
  }
}
