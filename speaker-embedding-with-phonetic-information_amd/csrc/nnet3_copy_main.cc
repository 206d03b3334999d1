// nnet3-copy shim - just enough of Kaldi's nnet3-copy for the one way the extraction scripts use it:
//   nnet="nnet3-copy --nnet-config=$dir/extract.config $srcdir/final.raw - |"
//   (egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:58-59)
// i.e. read a raw model, replace/add node lines (the output-node), write the model to a wxfilename.
// Components are re-emitted byte for byte.  Options outside that use (--edits, --learning-rate, ...) are
// rejected loudly rather than silently ignored, because they would change the model.
#include <stdio.h>
#include <string.h>

#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "kio.h"
#include "nnet3_raw.h"

int main(int argc, char** argv) {
  try {
    std::string nnet_config;
    bool binary = true, binary_set = false;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
      std::string a = argv[i];
      if (a.compare(0, 2, "--") == 0 && pos.empty()) {
        size_t eq = a.find('=');
        std::string name = a.substr(2, eq == std::string::npos ? std::string::npos : eq - 2);
        std::string val = eq == std::string::npos ? "" : a.substr(eq + 1);
        if (name == "nnet-config") nnet_config = val;
        else if (name == "binary") {
          binary = !(val == "false" || val == "f" || val == "0");
          binary_set = true;
        } else if (name == "print-args" || name == "verbose") {
        } else if (name == "help") {
          fputs("Usage: nnet3-copy [--nnet-config=<file>] [--binary=true|false] <raw-nnet-in> <raw-nnet-out>\n", stderr);
          return 0;
        } else {
          fprintf(stderr, "nnet3-copy (xvec-hip shim): option --%s is not supported by this shim\n", name.c_str());
          return 1;
        }
      } else {
        pos.push_back(a);
      }
    }
    if (pos.size() != 2) {
      fputs("Usage: nnet3-copy [--nnet-config=<file>] [--binary=true|false] <raw-nnet-in> <raw-nnet-out>\n", stderr);
      return 1;
    }
    xv::RawNnet net;
    net.ReadFrom(pos[0]);
    if (!nnet_config.empty()) {
      std::ifstream f(nnet_config);
      if (!f) throw xv::KioError("cannot open --nnet-config file " + nnet_config);
      std::stringstream ss;
      ss << f.rdbuf();
      net.ApplyNnetConfig(ss.str());
    }
    if (binary_set && binary != net.binary)
      throw xv::KioError("this shim cannot convert between binary and text models (input is " +
                         std::string(net.binary ? "binary" : "text") + ")");
    xv::Output out;
    out.Open(pos[1]);
    net.Write(out, net.binary);
    if (out.Close() != 0) throw xv::KioError("error closing output " + pos[1]);
    fprintf(stderr, "LOG (nnet3-copy[xvec-hip-0.1]:main()) Copied raw neural net from %s to %s\n", pos[0].c_str(), pos[1].c_str());
    return 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "ERROR (nnet3-copy[xvec-hip-0.1]:main()) %s\n", e.what());
    return -1;
  }
}
