// Host-only self test of the parsers that face untrusted input (model files, feature / vector archives), built with
// AddressSanitizer + UndefinedBehaviorSanitizer by `make sanitize` (plain g++, no HIP: kio.cc, nnet3_raw.cc,
// program.cc have no device dependency).  GPU AddressSanitizer is not available on the target pool, so this is where
// memory errors of the host layer are caught.
//
//   host_selftest <model.raw> <output-node> <features.ark> [variants per kind, default 200]
//
// 1. parses the model, lowers it, re-emits it and parses the copy again;
// 2. reads the archive and re-writes it through the table writer in binary and text form;
// 2b. binary archives: the index pass of the parallel table readers (MatrixTableIndexer + ReadIndexedMatrix) must deliver
//     exactly what the sequential reader delivers, through the archive itself and through a script file of its offsets;
// 3. replays both files truncated at `variants` positions and with 1.5 x `variants` single bits flipped: every variant must either parse or
//    throw KioError - never crash, hang or trip a sanitizer.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "kio.h"
#include "nnet3_raw.h"
#include "program.h"

namespace {

std::string Slurp(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  std::ostringstream o;
  o << f.rdbuf();
  return o.str();
}

// returns 0 parsed, 1 rejected with KioError
int TryModel(const std::string& bytes, const std::string& node) {
  try {
    xv::RawNnet net;
    net.Read(bytes);
    if (!node.empty()) net.ApplyNnetConfig("output-node name=output input=" + node + "\n");
    xv::TdnnProgram p = xv::LowerToProgram(net, "output");
    (void)p.Describe();
    (void)p.Macs(400);
    return 0;
  } catch (const xv::KioError&) {
    return 1;
  } catch (const std::bad_alloc&) {
    return 1;   // an absurd dimension in a corrupted header
  } catch (const std::length_error&) {
    return 1;
  }
}

int TryArchive(const std::string& bytes) {
  try {
    xv::Input in;
    in.OpenMemory(bytes.data(), bytes.size());
    for (int n = 0; n < 100000; ++n) {
      int c;
      while ((c = in.Peek()) >= 0 && isspace(c)) in.Get();
      if (c < 0) break;
      std::string key;
      while ((c = in.Peek()) >= 0 && !isspace(c)) key.push_back((char)in.Get());
      in.Get();
      const bool binary = xv::ReadBinaryHeader(in);
      xv::Matrix m;
      xv::ReadMatrix(in, binary, &m);
    }
    return 0;
  } catch (const xv::KioError&) {
    return 1;
  } catch (const std::bad_alloc&) {
    return 1;
  } catch (const std::length_error&) {
    return 1;
  }
}

// The indexed readers against the sequential one: same keys, same matrices, same order.  Returns 0 ok, 1 mismatch,
// 2 "not addressable" (text archive: the caller checks that this is what usable() says).
int CheckIndexer(const std::string& rspec, const std::string& seq_rspec, long* views) {
  xv::FileMapper mapper;
  xv::MatrixTableIndexer idx(rspec);
  if (!idx.usable()) return 2;
  xv::SequentialMatrixReader rd(seq_rspec);
  xv::Input in;
  std::string in_path, key, err;
  xv::Matrix want, got;
  xv::MatrixTableIndexer::Entry e;
  for (;;) {
    const bool a = idx.Next(&e), b = rd.Next(&key, &want, &err);
    if (a != b) return 1;
    if (!a) return 0;
    if (!e.error.empty() || !err.empty() || e.key != key) return 1;
    xv::ReadIndexedMatrix(e, &in, &in_path, &got);
    if (got.rows != want.rows || got.cols != want.cols || got.data != want.data) return 1;
    if (e.rows >= 0 && (e.rows != want.rows || e.cols != want.cols)) return 1;
    // the VIEW of the same object in the mapped file (what the reader threads of a table job hand out): a binary float matrix
    // must be viewable and hold the same bytes (compared with memcmp: the view has the archive's alignment, not a float's)
    xv::Matrix view;
    if (mapper.View(e, &view, true)) {
      ++*views;
      if (view.rows != want.rows || view.cols != want.cols || !view.data.empty()) return 1;
      if (view.cm) {   // a compressed view: what the device front-end uploads; expanded here it is what the reader delivers
        xv::Matrix full;
        xv::ExpandCompressedView(view, &full);
        if (full.rows != want.rows || full.cols != want.cols || full.data != want.data) return 1;
      } else if (memcmp(view.Data(), want.data.data(), want.data.size() * sizeof(float)) != 0) {
        return 1;
      }
    }
  }
}

// a damaged archive through the index pass: every variant either indexes + reads or throws KioError
int TryIndexedFile(const std::string& path) {
  try {
    xv::MatrixTableIndexer idx("ark:" + path);
    if (!idx.usable()) return 1;
    xv::Input in;
    std::string in_path;
    xv::MatrixTableIndexer::Entry e;
    xv::Matrix m;
    xv::FileMapper mapper;
    for (int n = 0; n < 100000 && idx.Next(&e); ++n) {
      xv::Matrix view;
      if (mapper.View(e, &view, true) && (long)view.rows * view.cols > 0) {   // a view of a damaged file must stay inside the file
        if (view.cm) {
          xv::Matrix full;
          xv::ExpandCompressedView(view, &full);   // touches every byte of the object
        } else {
          volatile unsigned char first = *(const unsigned char*)view.Data();
          volatile unsigned char last = ((const unsigned char*)view.Data())[(size_t)view.rows * view.cols * 4 - 1];
          (void)first;
          (void)last;
        }
      }
      xv::ReadIndexedMatrix(e, &in, &in_path, &m);
    }
    return 0;
  } catch (const xv::KioError&) {
    return 1;
  } catch (const std::bad_alloc&) {
    return 1;
  } catch (const std::length_error&) {
    return 1;
  }
}

}  // namespace

int main(int argc, char** argv) {
  if (argc != 4 && argc != 5) {
    fprintf(stderr, "usage: host_selftest <model.raw> <output-node> <features.ark> [variants]\n");
    return 2;
  }
  const int variants = argc == 5 ? atoi(argv[4]) : 200;
  const std::string model = Slurp(argv[1]), node = argv[2], ark = Slurp(argv[3]);
  if (model.empty() || ark.empty()) {
    fprintf(stderr, "host_selftest: empty input\n");
    return 2;
  }
  // 1. round trips
  xv::RawNnet net;
  net.Read(model);
  // (the writer re-emits components verbatim, so it keeps the flavour of its input)
  const int same = (model.size() >= 2 && model[0] == '\0' && model[1] == 'B') ? 1 : 0;
  for (int binary = same; binary <= same; ++binary) {
    const std::string tmp = std::string(argv[1]) + (binary ? ".selftest.bin" : ".selftest.txt");
    {
      xv::Output out;
      out.Open(tmp);
      net.Write(out, binary != 0);
      out.Close();
    }
    if (TryModel(Slurp(tmp), node) != 0) {
      fprintf(stderr, "host_selftest: re-emitted model (%s) does not parse\n", binary ? "binary" : "text");
      return 1;
    }
    remove(tmp.c_str());
  }
  if (TryModel(model, node) != 0 || TryArchive(ark) != 0) {
    fprintf(stderr, "host_selftest: the pristine inputs do not parse\n");
    return 1;
  }
  {
    xv::SequentialMatrixReader rd(std::string("ark:") + argv[3]);
    const std::string tb = std::string(argv[3]) + ".selftest.bin", tt = std::string(argv[3]) + ".selftest.txt";
    {
      xv::TableWriter wb("ark:" + tb), wt("ark,t:" + tt);
      std::string key, err;
      xv::Matrix m;
      while (rd.Next(&key, &m, &err)) {
        wb.WriteMat(key, m);
        wt.WriteMat(key, m);
      }
    }
    if (TryArchive(Slurp(tb)) != 0 || TryArchive(Slurp(tt)) != 0) {
      fprintf(stderr, "host_selftest: re-written archive does not parse\n");
      return 1;
    }
    // 2b. the index pass: the binary copy through the archive and through a script file of its offsets; the text copy
    // is not addressable and must say so
    {
      const std::string ts = std::string(argv[3]) + ".selftest.scp";
      {
        xv::TableWriter ws("ark,scp:" + tb + "," + ts);
        xv::SequentialMatrixReader rd2(std::string("ark:") + argv[3]);
        std::string key, err;
        xv::Matrix m;
        while (rd2.Next(&key, &m, &err)) ws.WriteMat(key, m);
      }
      long views = 0;
      if (CheckIndexer("ark:" + tb, "ark:" + tb, &views) != 0 || CheckIndexer("scp:" + ts, "ark:" + tb, &views) != 0 ||
          CheckIndexer("ark:" + tt, "ark:" + tt, &views) != 2) {
        fprintf(stderr, "host_selftest: the indexed table readers disagree with the sequential reader\n");
        return 1;
      }
      if (views == 0) {
        fprintf(stderr, "host_selftest: no object of the binary archive could be viewed in the mapped file\n");
        return 1;
      }
      // the input archive itself, when it is binary: it may hold COMPRESSED objects (the stored form of the recipes' features),
      // whose views the device front-end uploads as they are
      {
        xv::MatrixTableIndexer probe(std::string("ark:") + argv[3]);
        if (probe.usable()) {
          long v2 = 0;
          if (CheckIndexer(std::string("ark:") + argv[3], std::string("ark:") + argv[3], &v2) != 0) {
            fprintf(stderr, "host_selftest: the indexed readers / views disagree with the sequential reader on the input archive\n");
            return 1;
          }
          const std::string orig = Slurp(argv[3]), to = std::string(argv[3]) + ".selftest.dmg2";
          unsigned st2 = 4242;
          for (int i = 0; i < variants; ++i) {
            std::string bad = orig;
            st2 = st2 * 1664525u + 1013904223u;
            if (i % 2) bad.resize(orig.size() * (size_t)(i + 1) / (variants + 2));
            else bad[st2 % bad.size()] = (char)(bad[st2 % bad.size()] ^ (1u << ((st2 >> 20) & 7)));
            {
              std::ofstream f(to, std::ios::binary);
              f.write(bad.data(), (std::streamsize)bad.size());
            }
            (void)TryIndexedFile(to);
          }
          remove(to.c_str());
        }
      }
      // damaged copies of the binary archive through the index pass (truncations and bit flips, on disk: the indexer seeks)
      const std::string good = Slurp(tb), td = std::string(argv[3]) + ".selftest.dmg";
      unsigned state = 777;
      for (int i = 0; i < variants; ++i) {
        std::string bad = good;
        state = state * 1664525u + 1013904223u;
        if (i % 2) bad.resize(good.size() * (size_t)(i + 1) / (variants + 2));
        else bad[state % bad.size()] = (char)(bad[state % bad.size()] ^ (1u << ((state >> 20) & 7)));
        {
          std::ofstream f(td, std::ios::binary);
          f.write(bad.data(), (std::streamsize)bad.size());
        }
        (void)TryIndexedFile(td);
      }
      remove(td.c_str());
      remove(ts.c_str());
    }
    remove(tb.c_str());
    remove(tt.c_str());
  }
  // 2. truncations and byte flips
  long parsed = 0, rejected = 0;
  auto replay = [&](const std::string& good, bool is_model) {
    const size_t n = good.size();
    for (int i = 1; i <= variants; ++i) {
      const size_t cut = n * i / (variants + 1);
      const int r = is_model ? TryModel(good.substr(0, cut), node) : TryArchive(good.substr(0, cut));
      (r ? rejected : parsed)++;
    }
    unsigned state = 12345;
    for (int i = 0; i < variants * 3 / 2; ++i) {
      state = state * 1664525u + 1013904223u;
      std::string bad = good;
      // most flips in the first 4 KiB (tokens, dimensions), the rest anywhere
      const size_t pos = (i % 3 ? state % (n < 4096 ? n : 4096) : state % n);
      bad[pos] = (char)(bad[pos] ^ (1u << ((state >> 20) & 7)));
      const int r = is_model ? TryModel(bad, node) : TryArchive(bad);
      (r ? rejected : parsed)++;
    }
  };
  replay(model, true);
  replay(ark, false);
  printf("host_selftest: ok (%ld damaged variants parsed, %ld rejected cleanly)\n", parsed, rejected);
  return 0;
}
