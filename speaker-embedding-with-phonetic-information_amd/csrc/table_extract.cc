// Table-driven extraction loop.  See table_extract.h.
#include "table_extract.h"

#include <dirent.h>
#include <math.h>
#include <pthread.h>
#include <sys/resource.h>
#include <sys/time.h>
#include <memory>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <thread>
#include <vector>

#include "backend.h"
#include "calib_file.h"
#include "knobs.h"
#include "kio.h"

namespace xv {

namespace {
struct Utt {
  std::string key;
  Matrix feats;
};
struct Batch {
  std::vector<Utt> utts;
  bool last = false;
};
}  // namespace

Engine::Calibration CalibrateOnUtterances(Engine* engine, const ExtractOptions& opt, const std::vector<CalibUtt>& utts,
                                          const LogFn& log) {
  Engine::Calibration c;
  c.chosen = engine->fast_mode();
  if (!engine->can_switch_fast_mode()) return c;
  const int D = engine->info().input_dim;
  std::unique_ptr<RandomAccessVectorReader> vad;
  if (!opt.vad_rspecifier.empty()) vad.reset(new RandomAccessVectorReader(opt.vad_rspecifier));
  const bool use_frontend = opt.cmn_window > 0 || vad;
  std::vector<float> feats;
  std::vector<int32_t> offs(1, 0);
  for (size_t i = 0; i < utts.size() && (int)i < opt.calibrate_utts; ++i) {   // the first calibrate_utts utterances of the list, whoever calls
    const CalibUtt& u = utts[i];
    if (u.rows <= 0 || u.cols != D) continue;
    std::vector<float> tmp;
    const float* rows = u.data;
    int T = u.rows;
    if (use_frontend) {
      const std::vector<float>* v = nullptr;
      if (vad) {
        if (!vad->HasKey(*u.key)) continue;
        v = &vad->Value(*u.key);
        if ((int)v->size() != T) continue;
      }
      std::vector<int32_t> sel_row, sel_utt, raw_off = {0, T};
      for (int t = 0; t < T; ++t)
        if (!v || (*v)[t] != 0.f) {
          sel_row.push_back(t);
          sel_utt.push_back(0);
        }
      if (sel_row.empty()) continue;
      tmp.resize(sel_row.size() * (size_t)D);
      engine->FrontEndHost(u.data, raw_off.data(), 1, sel_row.data(), sel_utt.data(), (int)sel_row.size(), opt.cmn_window,
                           opt.cmn_center, opt.cmn_min_window, tmp.data());
      rows = tmp.data();
      T = (int)sel_row.size();
    }
    const int len = (opt.chunk_size > 0 && T > opt.chunk_size) ? opt.chunk_size : T;   // the utterance's first chunk
    if (len < engine->info().min_frames || (long)offs.back() + len > opt.max_batch_rows) continue;
    feats.insert(feats.end(), rows, rows + (size_t)len * D);
    offs.push_back(offs.back() + len);
  }
  const int n = (int)offs.size() - 1;
  const auto t_cal = std::chrono::steady_clock::now();
  // A measurement that fails - a device fault in a candidate arithmetic, a stream-K time-out, a device that does not reproduce
  // its own bits (Engine::Calibrate checks) - is REPEATED ONCE, with a WARNING that says why: the first forward passes of a
  // job's context are where four co-tenant jobs of a recipe meet on one GPU (GPUTEST r05, profiles/r06_cotenancy.md), and a
  // job of an hour should not die of - nor silently take another arithmetic from - one odd pass of its start-up measurement.
  // The second failure ends the job with its message.  (Errors of the EXTRACTION are never retried: a result is a result.)
  if (n > 0) {
    try {
      c = engine->Calibrate(feats.data(), offs.data(), n, opt.calibrate_tol);
    } catch (const EngineError& ex) {
      log("WARNING", std::string("calibration attempt failed (") + ex.what() + "); the measurement is repeated once");
      c = engine->Calibrate(feats.data(), offs.data(), n, opt.calibrate_tol);
    }
  }
  if (getenv("XVEC_TIMING")) {
    std::ostringstream t;
    t << "calibration: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_cal).count() << " s for " << n << " chunks";
    log("LOG", t.str());
  }
  std::ostringstream m;
  m.precision(3);
  if (c.checked == 0) {
    m << "calibration: none of the " << utts.size() << " sampled utterances has a chunk long enough for the fast arithmetic; keeping "
      << PrecisionName(c.chosen);
  } else {
    m << "calibration on " << c.checked << " chunks against the three-pass arithmetic: fp16mx " << c.err_mx << " (on the " << c.checked_mx
      << " chunks it runs fast; it takes " << Engine::kCalibMinChunks << " to choose it), fp16mx2 " << c.err_mx2 << " (tolerance "
      << opt.calibrate_tol << ") -> " << PrecisionName(c.chosen);
    if (c.lite_mask) {
      m << " with " << __builtin_popcountll(c.lite_mask) << " of its layers in 1.25 passes (";
      bool first = true;
      for (size_t i = 0; i < engine->info().layers.size() && i < 64; ++i)
        if ((c.lite_mask >> i) & 1) {
          m << (first ? "" : ", ") << engine->info().layers[i].name;
          first = false;
        }
      m << ": " << c.err_lite << "; chosen on one half of the sample, " << c.err_holdout << " on the " << c.checked_holdout
        << " held-out chunks)";
    }
    if (c.lite_dropped) m << "; " << c.lite_dropped << " layer(s) admitted by the selection half failed the held-out half and were taken out again";
    if (c.tail > 0.f) m << "; projected tail of what runs (mean + 6 sd of the per-chunk error) " << c.tail;
    if (c.chosen != kPrecFp16Mx && c.err_mx <= opt.calibrate_tol && c.checked_mx >= Engine::kCalibMinChunks)
      m << "; fp16mx was within the tolerance on every sampled chunk but projects a tail of " << c.tail_mx << " and was turned down";
  }
  log("LOG", m.str());
  return c;
}

// Calibration sample of a table.  Tables whose objects can be addressed (an archive in a regular file, a script file) are
// indexed once - headers only - and calibrate_utts utterances are drawn EVENLY over the whole list: the reference's lists are
// sorted by speaker (utils/data/split_data.sh:18-21 splits them per speaker), so the head of a list is one or two speakers,
// and the choice governs every utterance of the job.  Streams (the feature pipe of extract_xvectors_new.sh:79) can only be
// read front to back: their sample is the head, and the log says so.  *strided tells which one it was.
static void SampleTable(const ExtractOptions& opt, const std::string& feat_rspec, std::vector<std::string>* keys,
                        std::vector<Matrix>* mats, bool* strided, long* n_list, TableIndex* index) {
  *strided = false;
  *n_list = -1;
  const int want = std::max(1, opt.calibrate_utts);
  {
    MatrixTableIndexer ix(feat_rspec);
    if (ix.usable()) {
      // ONE index of the table serves the sample and, through `index`, the batching pass of the extraction (the
      // headers of a 1 M-line scp were visited twice).  An object the indexer cannot get past (a text object, a truncated
      // archive) ends the index there and is remembered: the sample comes from what was indexed, and the extraction writes
      // everything in front of the bad object before it reports it - like the sequential reader and like the reference.
      TableIndex local;
      TableIndex& all = index ? *index : local;
      try {
        MatrixTableIndexer::Entry e;
        while (ix.Next(&e)) all.entries.push_back(std::move(e));
      } catch (const std::exception& ex) {
        all.error = ex.what();
      }
      all.valid = true;
      std::vector<size_t> good;
      for (size_t i = 0; i < all.entries.size(); ++i)
        if (all.entries[i].error.empty()) good.push_back(i);
      *strided = true;
      *n_list = (long)good.size();
      const long n = (long)good.size();
      Input in;
      std::string in_path;
      long prev = -1;
      for (int i = 0; i < want && n > 0; ++i) {
        // centres of `want` equal slices of the list (all of it when it is shorter)
        const long k = n <= want ? i : (long)(((2 * (long long)i + 1) * n) / (2LL * want));
        if (k >= n || k == prev) continue;
        prev = k;
        Matrix m;
        try {
          ReadIndexedMatrix(all.entries[good[k]], &in, &in_path, &m);
        } catch (const std::exception&) {
          continue;   // the extraction itself will report it
        }
        keys->push_back(all.entries[good[k]].key);
        mats->push_back(std::move(m));
      }
      return;
    }
  }
  SequentialMatrixReader rd(feat_rspec);
  std::string key, e;
  Matrix m;
  while ((int)keys->size() < want && rd.Next(&key, &m, &e)) {
    if (!e.empty()) continue;
    keys->push_back(key);
    mats->push_back(std::move(m));
  }
  // (the reader is closed here; for a pipe that ends the producer early, which is what a "head" of the list wants)
}

Engine::Calibration CalibrateOnTable(Engine* engine, const ExtractOptions& opt, const std::string& feat_rspec, const LogFn& log,
                                     TableIndex* index) {
  std::vector<std::string> keys;
  std::vector<Matrix> mats;
  bool strided = false;
  long n_list = -1;
  SampleTable(opt, feat_rspec, &keys, &mats, &strided, &n_list, index);
  std::ostringstream m;
  if (strided) m << "calibration sample: " << keys.size() << " utterances spread evenly over the " << n_list << " of the table";
  else m << "calibration sample: the first " << keys.size() << " utterances of the stream (a stream cannot be sampled any other way)";
  log("LOG", m.str());
  std::vector<CalibUtt> utts;
  for (size_t i = 0; i < keys.size(); ++i) utts.push_back(CalibUtt{&keys[i], mats[i].data.data(), mats[i].rows, mats[i].cols});
  return CalibrateOnUtterances(engine, opt, utts, log);
}

TableExtractResult RunTableExtraction(Engine* engine, const ExtractOptions& opt, const std::string& feat_rspec,
                                      const std::string& vec_wspec, const LogFn& log) {
  return RunTableExtraction(std::vector<Engine*>(1, engine), opt, feat_rspec, vec_wspec, log);
}

// Several engines = several GPUs driven by this one process (nnet3-xvector-compute --devices=..., the `--nj 1` form of
// extract_xvectors_new.sh:83-93).  Whole batches are dealt to the engines round-robin and finalised in the order they were
// submitted, through the one writer: the output archive is byte-identical to a one-GPU run of the same job.  The arithmetic
// is chosen once, on engines[0], and applied to the others (like rank 0's choice in dist_extract.py).
TableExtractResult RunTableExtraction(const std::vector<Engine*>& engines, const ExtractOptions& opt, const std::string& feat_rspec,
                                      const std::string& vec_wspec, const LogFn& log) {
  if (engines.empty() || !engines[0]) throw EngineError("RunTableExtraction: no engine");
  Engine* const engine = engines[0];   // calibration, host-side front-end fallback, back-end
  const int NE = (int)engines.size();
  for (Engine* e : engines)
    if (!e || e->info().input_dim != engine->info().input_dim || e->info().output_dim != engine->info().output_dim ||
        e->can_switch_fast_mode() != engine->can_switch_fast_mode())
      throw EngineError("RunTableExtraction: the engines do not hold the same model");
  auto share_choice = [&] {   // what engines[0] was calibrated to, on every other engine
    for (int k = 1; k < NE; ++k) {
      if (!engines[k]->can_switch_fast_mode()) continue;
      engines[k]->SetFastMode(engine->fast_mode());
      if (engine->lite_mask()) engines[k]->SetLiteMask(engine->lite_mask());
    }
  };
  TableExtractResult res;
  const int D = engine->info().input_dim, E = engine->info().output_dim;
  // option errors are raised before the reader thread and the output files exist
  if (!opt.backend_mean.empty() && (int)opt.backend_mean.size() != E)
    throw KioError("--backend-mean has dimension " + std::to_string(opt.backend_mean.size()) + ", the embedding has " +
                   std::to_string(E));
  if (!opt.backend_transform.empty() && opt.backend_t_cols != E && opt.backend_t_cols != E + 1)
    throw KioError("Dimension mismatch: the embedding has dimension " + std::to_string(E) + " and --backend-transform has " +
                   std::to_string(opt.backend_t_cols) + " columns");
  // everything that can fail on a user error (unreadable VAD table, unwritable output) happens BEFORE the reader
  // thread exists; what can still throw afterwards unwinds through ReaderGuard, which stops and joins the thread
  // (a joinable std::thread destroyed by unwinding would call std::terminate: SIGABRT instead of "ERROR ..." + 255)
  const bool use_frontend = opt.cmn_window > 0 || !opt.vad_rspecifier.empty();
  std::unique_ptr<RandomAccessVectorReader> vad;
  if (!opt.vad_rspecifier.empty()) vad.reset(new RandomAccessVectorReader(opt.vad_rspecifier));
  TableWriter writer(vec_wspec);
  FileMapper mapper;   // declared before everything that can hold a view of its mappings (batches, work slots, reader threads)
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Batch> queue;
  // feature buffers of finished batches, handed back to the stream reader (under mu): a matrix read into a buffer that is
  // already large enough is not zero-filled first - for a pipe, whose single reader thread is the job's ceiling, that pass over
  // every byte was a fifth of the thread's time
  std::vector<std::vector<float>> buf_pool;
  std::string reader_error;
  long num_fail_read = 0;
  bool stop = false;   // under mu: the consumer is gone, the reader must not block on a full queue
  std::mutex warn_mu;   // readers and (with several engines) consumers warn from their own threads
  auto warn = [&](const std::string& m) {
    std::unique_lock<std::mutex> lk(warn_mu);
    log("WARNING", m);
  };

  // ---- feature ingestion --------------------------------------------------------------------------------------------
  // Tables whose objects can be addressed (binary archive in a regular file, script file of path:offset entries) are read by
  // several threads: one index pass forms the batches from the headers alone (MatrixTableIndexer: no data touched), the
  // readers fill whole batches, a sequencer hands them to the consumer in table order.  One thread parsing 7 GB/s of
  // archive was what the loop waited for 42 % of the time (round 2).  Streams (pipes, standard input - the feature
  // pipeline of extract_xvectors_new.sh:79) can only be read front to back: one reader thread, as before.
  // Everything that can fail on a user error happens before a thread exists (the indexer opens the table here).
  const bool opt_is_scp = ParseRspecifier(feat_rspec).is_scp;
  int n_readers = std::min(16, 4 * NE);
  n_readers = std::max(1, std::min(16, DebugKnobInt("readers", n_readers)));
  std::unique_ptr<MatrixTableIndexer> indexer;
  if (n_readers > 1) {
    indexer.reset(new MatrixTableIndexer(feat_rspec));
    if (!indexer->usable()) indexer.reset();
  }
  struct PlanBatch {
    long seq = 0;
    std::vector<MatrixTableIndexer::Entry> entries;
    bool last = false;
  };
  TableIndex pre_index;                 // the calibration's index of an addressable table, reused by the index pass
  std::deque<PlanBatch> plans;          // under mu: index pass -> readers
  std::map<long, Batch> filled;         // under mu: readers -> sequencer (by batch number)
  long next_plan = 0, next_out = 0;     // batches planned / moved to the consumer's queue
  long n_consumed = 0;                  // batches the consumer has taken (bounds what is held in memory)
  std::vector<std::thread> threads;

  auto sequential_reader = [&] {
    try {
      SequentialMatrixReader rd(feat_rspec);
      Batch cur;
      long rows = 0;
      std::string key, e;
      Matrix m;
      auto push = [&](bool last) {
        cur.last = last;
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return queue.size() < 2 || stop; });
        if (stop) return;
        queue.push_back(std::move(cur));
        cur = Batch();
        rows = 0;
        cv.notify_all();
      };
      auto take_buffer = [&] {
        std::unique_lock<std::mutex> lk(mu);
        if (m.data.capacity() == 0 && !buf_pool.empty()) {
          m.data = std::move(buf_pool.back());
          buf_pool.pop_back();
        }
      };
      take_buffer();
      while (rd.Next(&key, &m, &e)) {
        {
          std::unique_lock<std::mutex> lk(mu);
          if (stop) break;
        }
        if (!e.empty()) {
          warn("failed to read features for " + key + ": " + e);
          ++num_fail_read;
          continue;
        }
        // a batch never exceeds max_batch_rows (one forward pass) unless a single utterance does
        if (!cur.utts.empty() && (rows + m.rows > opt.max_batch_rows || (int)cur.utts.size() >= opt.max_batch_chunks)) push(false);
        Utt u;
        u.key = key;
        u.feats = std::move(m);
        rows += u.feats.rows;
        cur.utts.push_back(std::move(u));
        m = Matrix();
        take_buffer();
      }
      res.reader_status = rd.Close();
      push(true);
    } catch (const std::exception& ex) {
      std::unique_lock<std::mutex> lk(mu);
      reader_error = ex.what();
      Batch b;
      b.last = true;
      queue.push_back(std::move(b));
      cv.notify_all();
    }
  };
  // index pass: batches by the same rule as the sequential reader, from the headers (objects whose size is not known
  // without reading them - text, pipes - count as 0 rows: such batches are bounded by max_batch_chunks only, and the
  // extractor splits what does not fit one forward pass)
  auto index_pass_body = [&] {
    PlanBatch cur;
    long rows = 0;
    auto push = [&](bool last) {
      cur.last = last;
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return next_plan - n_consumed < 2 * n_readers + 2 || stop; });
      if (stop) return;
      cur.seq = next_plan++;
      plans.push_back(std::move(cur));
      cur = PlanBatch();
      rows = 0;
      cv.notify_all();
    };
    try {
      MatrixTableIndexer::Entry e;
      size_t pre_i = 0;
      // the calibration's index of the table, when there is one (it is consumed: entries are moved out), else the indexer
      auto next_entry = [&]() {
        if (!pre_index.valid) return indexer->Next(&e);
        if (pre_i < pre_index.entries.size()) {
          e = std::move(pre_index.entries[pre_i++]);
          return true;
        }
        if (!pre_index.error.empty()) throw KioError(pre_index.error);   // where that index stopped: reported after what precedes it
        return false;
      };
      while (next_entry()) {
        {
          std::unique_lock<std::mutex> lk(mu);
          if (stop) break;
        }
        if (!e.error.empty()) {
          warn("failed to read features for " + e.key + ": " + e.error);
          std::unique_lock<std::mutex> lk(mu);
          ++num_fail_read;
          continue;
        }
        const int r = std::max(e.rows, 0);
        if (!cur.entries.empty() && (rows + r > opt.max_batch_rows || (int)cur.entries.size() >= opt.max_batch_chunks)) push(false);
        rows += r;
        cur.entries.push_back(std::move(e));
      }
    } catch (const std::exception& ex) {
      std::unique_lock<std::mutex> lk(mu);
      if (reader_error.empty()) reader_error = ex.what();
    }
    push(true);
  };
  // anything thrown outside the guarded calls above (an allocation failing while a batch is assembled) must not leave the
  // thread: an exception escaping a std::thread is std::terminate.  It becomes the job's reader error, and a terminal batch
  // releases whoever waits.
  auto reader_failed = [&](const char* what) {
    std::unique_lock<std::mutex> lk(mu);
    if (reader_error.empty()) reader_error = what;
    Batch b;
    b.last = true;
    queue.push_back(std::move(b));
    PlanBatch again;
    again.seq = -1;
    again.last = true;
    plans.push_back(std::move(again));
    cv.notify_all();
  };
  auto index_pass = [&] {
    (void)pthread_setname_np(pthread_self(), "xv-index");
    try {
      index_pass_body();
    } catch (const std::exception& ex) {
      reader_failed(ex.what());
    } catch (...) {
      reader_failed("unknown error in the index pass");
    }
  };
  const bool use_views = DebugKnobInt("mmap", 1) != 0;
  const bool cm_on_device = DebugKnobInt("cm_on_device", 1) != 0;   // compressed matrices of a front-end job: expanded on the GPU
  auto parallel_reader_body = [&] {
    Input in;
    std::string in_path;
    for (;;) {
      PlanBatch pb;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !plans.empty() || stop; });
        if (stop) return;
        pb = std::move(plans.front());
        plans.pop_front();
        if (pb.last) {   // the other readers must see the end as well
          PlanBatch again;
          again.seq = -1;
          again.last = true;
          plans.push_back(std::move(again));
          cv.notify_all();
          if (pb.seq < 0) return;
        }
      }
      Batch b;
      b.last = pb.last;
      for (MatrixTableIndexer::Entry& e : pb.entries) {
        Utt u;
        try {
          // a float matrix in a regular file is not read at all: the utterance is a view of the mapped archive, and its bytes
          // are touched once, by the copy into the pinned staging buffer
          if (!use_views || !mapper.View(e, &u.feats, use_frontend && cm_on_device)) ReadIndexedMatrix(e, &in, &in_path, &u.feats);
        } catch (const std::exception& ex) {
          if (e.offset >= 0 && !opt_is_scp) {   // a damaged archive is fatal, as in the sequential reader
            std::unique_lock<std::mutex> lk(mu);
            if (reader_error.empty()) reader_error = ex.what();
            continue;
          }
          warn("failed to read features for " + e.key + ": " + ex.what());
          std::unique_lock<std::mutex> lk(mu);
          ++num_fail_read;
          continue;
        }
        u.key = std::move(e.key);
        b.utts.push_back(std::move(u));
      }
      {
        std::unique_lock<std::mutex> lk(mu);
        filled.emplace(pb.seq, std::move(b));
        // sequencer: whatever is next in table order goes to the consumer's queue
        for (auto it = filled.find(next_out); it != filled.end(); it = filled.find(next_out)) {
          queue.push_back(std::move(it->second));
          filled.erase(it);
          ++next_out;
        }
        cv.notify_all();
        if (pb.last) return;
      }
    }
  };
  auto parallel_reader = [&] {
    (void)pthread_setname_np(pthread_self(), "xv-read");
    try {
      parallel_reader_body();
    } catch (const std::exception& ex) {
      reader_failed(ex.what());
    } catch (...) {
      reader_failed("unknown error in a reader thread");
    }
  };
  // ---- the arithmetic of the job -----------------------------------------------------------------------------------
  // A SHARED choice first (calib_file.h): what the file holds is applied and nothing is measured; a missing file is measured
  // here, published, and read back - this job's choice or that of the job that published first.
  const bool shared = !opt.calibration_file.empty() && engine->can_switch_fast_mode();
  auto adopt = [&](const SharedChoice& sc, const std::string& how) {
    AdoptSharedChoice(engine, sc, opt.calibration_file);
    share_choice();
    std::ostringstream m;
    m << "arithmetic " << PrecisionName(sc.precision);
    if (sc.lite_mask) m << " with " << __builtin_popcountll(sc.lite_mask) << " of its layers in 1.25 passes (mask 0x" << std::hex << sc.lite_mask << std::dec << ")";
    m << ": " << how << " " << opt.calibration_file << (sc.note.empty() ? "" : " [" + sc.note + "]");
    log("LOG", m.str());
  };
  auto publish = [&](const Engine::Calibration* c, const std::string& sample) {   // after a measurement (or a failed one: c = null)
    if (!shared) return;
    SharedChoice mine, got;
    mine.model = engine->info().fingerprint;
    mine.precision = engine->fast_mode();
    mine.lite_mask = engine->lite_mask();
    mine.tolerance = opt.calibrate_tol;
    std::ostringstream n;
    n.precision(3);
    if (c && c->checked > 0)
      n << "measured on " << c->checked << " chunks (" << sample << "): fp16mx " << c->err_mx << ", fp16mx2 " << c->err_mx2 << ", projected tail "
        << c->tail;
    else
      n << "nothing could be measured (" << sample << "): the packed arithmetic";
    mine.note = n.str();
    const bool won = PublishCalibrationFile(opt.calibration_file, mine, &got);
    adopt(got, won ? "measured here and published as" : "another job published first; adopted from");
  };
  bool want_calibrate = opt.calibrate && engine->can_switch_fast_mode();
  if (shared) {
    SharedChoice sc;
    if (ReadCalibrationFile(opt.calibration_file, &sc)) {
      adopt(sc, "read from");
      want_calibrate = false;
    } else {
      want_calibrate = true;
    }
  }
  if (want_calibrate && !shared)
    log("LOG", "the arithmetic is measured for THIS job only (--calibrate without a --calibration file): jobs over other shards "
               "of the same list measure their own samples and may choose differently");
  bool calibrated = !want_calibrate;   // nothing to choose: no batch is held back
  if (!calibrated && indexer) {
    // addressable table: the sample is spread over the whole list (SampleTable), drawn before the readers start.  An unreadable
    // sample (KioError) is not the job's failure: the packed arithmetic stays and the extraction reports the bad object itself,
    // after writing the good ones.  Anything else - a device fault in a candidate arithmetic, a stream-K time-out, a failed
    // allocation - ends the job (swallowed, it let independent jobs over one table run different arithmetics
    // and exit 0).
    try {
      const Engine::Calibration c = CalibrateOnTable(engine, opt, feat_rspec, log, &pre_index);
      share_choice();
      publish(&c, "spread evenly over the table");
    } catch (const KioError& ex) {
      warn(std::string("calibration sample unreadable (") + ex.what() + "); keeping " + PrecisionName(engine->fast_mode()));
      share_choice();
      publish(nullptr, "sample unreadable");
    }
    calibrated = true;
  }
  if (indexer) {
    threads.emplace_back(index_pass);
    for (int i = 0; i < n_readers; ++i) threads.emplace_back(parallel_reader);
  } else {
    threads.emplace_back(sequential_reader);
  }
  struct ReaderGuard {
    std::vector<std::thread>& ts;
    std::mutex& mu;
    std::condition_variable& cv;
    bool& stop;
    ~ReaderGuard() {
      {
        std::unique_lock<std::mutex> lk(mu);
        stop = true;
        cv.notify_all();
      }
      for (std::thread& t : ts)
        if (t.joinable()) t.join();
    }
  } reader_guard{threads, mu, cv, stop};

  const auto t0 = std::chrono::steady_clock::now();
  struct rusage ru0;
  getrusage(RUSAGE_SELF, &ru0);
  const bool has_backend = !opt.backend_mean.empty() || !opt.backend_transform.empty() || opt.backend_normalize;
  std::string fatal;
  std::mutex fatal_mu;
  auto set_fatal = [&](const std::string& m) {
    std::unique_lock<std::mutex> lk(fatal_mu);
    if (fatal.empty()) fatal = m;
  };
  auto is_fatal = [&] {
    std::unique_lock<std::mutex> lk(fatal_mu);
    return !fatal.empty();
  };

  // Up to kNumHostSlots batches queued on a device (ExtractJob::Start returns at once): after submitting batch i a consumer
  // finalises batch i-2 (average, back-end, hand over to the writer) and packs batch i+1 while i-1 and i keep the GPU busy - with
  // only two, both lanes ran their batches side by side, finished together, and the GPU idled while the host caught up
  // (measured: 20 % idle).  Output order = input order.
  struct Work {
    Batch b;
    long g = 0;                  // number of the batch in table order
    std::vector<int> idx;        // utterances of b that entered the device batch
    std::vector<float> packed;   // front-end path only: the processed rows, back to back
    std::vector<int32_t> offs;
    std::vector<const float*> uptr;   // plain path: rows of the utterances where the reader left them
    std::vector<int32_t> urows;
    ExtractJob job;
  };
  // What a finalised batch hands to the writer: one record per utterance that reached the device, in table order.
  struct OutBatch {
    std::vector<std::string> keys;
    std::vector<float> vecs;   // [keys.size()][VE]
    int VE = 0;
    std::vector<char> okv;
    std::vector<std::string> whys;
    std::vector<int> rows;
  };
  // One consumer per engine (= per GPU): its own ring of host slots, scratch and counters.  With ONE engine the consumer runs on the
  // calling thread and writes as it goes (rounds 2-4's loop).  With several (nnet3-xvector-compute --devices) each consumer is a
  // thread of its own - the packing of a batch (one host copy per byte) and the wait for its device are what a single thread could
  // not do for more than one GPU - batches are dealt to them round-robin in table order, and the calling thread writes the
  // finalised batches in that order: the archive is byte-identical to the one-GPU job's.
  constexpr int NS = Engine::kNumHostSlots;
  struct Consumer {
    Engine* eng = nullptr;
    std::vector<Work> work;
    int cur = 0;
    long seq = 0;   // running batch number on this engine (selects its lane)
    std::vector<int32_t> sel_row, sel_utt, poffs;
    std::vector<float> processed, emb, post;
    std::vector<int32_t> ok;
    std::vector<std::string> why;
    long num_fail = 0;   // utterances rejected before they reached the device
    long num_cm_device = 0;   // utterances that went up compressed and were expanded on the device
    long num_fe_host = 0;     // batches of a front-end job that had to take the host round trip
    double t_pack = 0, t_start = 0, t_fin = 0;
  };
  std::vector<Consumer> cons(NE);
  for (int e = 0; e < NE; ++e) {
    cons[e].eng = engines[e];
    cons[e].work.resize(NS);
  }
  std::mutex vad_mu, out_mu;
  std::condition_variable out_cv;
  std::map<long, OutBatch> done;   // NE > 1: finalised batches waiting for their turn at the writer
  // stage timing (XVEC_TIMING=1 logs it): waiting for the reader, packing, submitting, finishing
  const bool timing = getenv("XVEC_TIMING") != nullptr;
  double t_wait = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double>(b - a).count();
  };

  // the writer (calling thread only)
  auto write_out = [&](const OutBatch& ob) {
    for (size_t k = 0; k < ob.keys.size(); ++k) {
      if (!ob.okv[k]) {
        warn(ob.whys[k] + ": " + ob.keys[k]);
        ++res.num_fail;
        continue;
      }
      writer.WriteVec(ob.keys[k], ob.vecs.data() + k * (size_t)ob.VE, ob.VE);
      res.frames += ob.rows[k];
      ++res.num_success;
    }
  };
  auto deliver = [&](long g, OutBatch&& ob) {
    if (NE == 1) {
      write_out(ob);   // same thread, already in table order
      return;
    }
    std::unique_lock<std::mutex> lk(out_mu);
    done.emplace(g, std::move(ob));
    out_cv.notify_all();
  };

  auto finalize = [&](Consumer& C, Work& w) {
    const int n = (int)w.idx.size();
    C.emb.resize((size_t)n * E);
    C.ok.assign(n, 0);
    C.why.assign(n, std::string());
    w.job.Finish(C.emb.data(), C.ok.data(), &C.why);
    const float* vec = C.emb.data();
    int VE = E;
    if (n && has_backend) {
      BackendOptions bo;
      bo.mean = opt.backend_mean.empty() ? nullptr : opt.backend_mean.data();
      bo.transform = opt.backend_transform.empty() ? nullptr : opt.backend_transform.data();
      bo.t_rows = opt.backend_t_rows;
      bo.t_cols = opt.backend_t_cols;
      bo.normalize = opt.backend_normalize;
      bo.scaleup = opt.backend_scaleup;
      VE = bo.transform ? bo.t_rows : E;
      C.post.resize((size_t)n * VE);
      for (int k = 0; k < n; ++k)
        if (!C.ok[k]) std::fill(C.emb.begin() + (size_t)k * E, C.emb.begin() + (size_t)(k + 1) * E, 0.f);
      BackendApply(C.eng->device(), C.emb.data(), n, E, bo, C.post.data(), nullptr);
      vec = C.post.data();
    }
    OutBatch ob;
    ob.VE = VE;
    ob.keys.reserve(n);
    ob.vecs.assign(vec, vec + (size_t)n * VE);
    ob.okv.resize(n);
    ob.whys.resize(n);
    ob.rows.resize(n);
    for (int k = 0; k < n; ++k) {
      Utt& u = w.b.utts[w.idx[k]];
      ob.keys.push_back(std::move(u.key));   // (the batch is dropped below)
      ob.okv[k] = C.ok[k] ? 1 : 0;
      if (!C.ok[k]) ob.whys[k] = C.why[k];
      ob.rows[k] = u.feats.rows;
    }
    if (!indexer) {   // stream reader: the batch's feature buffers go back to it
      std::unique_lock<std::mutex> lk(mu);
      for (Utt& u : w.b.utts)
        if (buf_pool.size() < 4096 && u.feats.data.capacity()) buf_pool.push_back(std::move(u.feats.data));
    }
    w.b = Batch();
    deliver(w.g, std::move(ob));
  };

  // One batch through one consumer: reject what cannot run, pack, submit; then finalise the consumer's oldest batch in flight.
  // Every batch number g is delivered exactly once (an empty record when nothing of the batch reaches the writer), so the ordered
  // writer never waits for a number that does not come.
  auto process = [&](Consumer& C, Batch&& b, long g) {
    bool submitted_batch = false;
    if (!b.utts.empty() && !is_fatal()) {
      try {
        const auto tp0 = now();
        Work& w = C.work[C.cur];
        w.b = std::move(b);
        w.g = g;
        std::vector<int32_t>& sel_row = C.sel_row;
        std::vector<int32_t>& sel_utt = C.sel_utt;
        std::vector<int32_t>& poffs = C.poffs;
        std::vector<float>& processed = C.processed;
        std::vector<float>& packed = w.packed;
        std::vector<int32_t>& offs = w.offs;
        std::vector<int>& idx = w.idx;
        offs.assign(1, 0);
        idx.clear();
        // the rows stay where the reader put them (w.b keeps the utterances alive until finalize): the chunks are copied
        // from there straight into the engine's pinned staging buffer (ExtractJob::StartPtrs) - one host copy per byte
        std::vector<const float*>& uptr = w.uptr;
        std::vector<int32_t>& urows = w.urows;
        uptr.clear();
        urows.clear();
        for (size_t i = 0; i < w.b.utts.size(); ++i) {
          const Utt& u = w.b.utts[i];
          if (u.feats.rows > 0 && u.feats.cols != D) {
            std::ostringstream m;
            m << "feature dimension " << u.feats.cols << " of utterance " << u.key << " does not match the model's " << D;
            warn(m.str());
            ++C.num_fail;
            continue;
          }
          uptr.push_back(u.feats.Data());
          urows.push_back(u.feats.rows);
          idx.push_back((int)i);
        }
        int n = (int)idx.size();
        bool submitted = false;
        if (use_frontend && n) {
          // sliding CMN over the WHOLE utterance first, then keep the voiced frames (order of the two pipe stages)
          std::vector<int> keep_idx;
          std::vector<const float*> rawp, vadp;
          std::vector<int32_t> rrows;
          std::vector<const uint8_t*> cmp;    // compressed views of a mapped archive: expanded on the device (or below, on the host)
          std::vector<size_t> cmb;
          bool any_cm = false;
          for (int k = 0; k < n; ++k) {
            const Utt& u = w.b.utts[idx[k]];
            const int T = u.feats.rows;
            const std::vector<float>* v = nullptr;
            if (vad) {
              std::unique_lock<std::mutex> vlk(vad_mu);   // (the reader caches what it loads: one consumer at a time)
              if (!vad->HasKey(u.key)) {
                warn("No VAD input found for utterance " + u.key);
                ++C.num_fail;
                continue;
              }
              v = &vad->Value(u.key);
              vlk.unlock();
              if ((int)v->size() != T) {
                std::ostringstream m;
                m << "Mismatch in number of frames " << T << " for features and VAD " << v->size() << ", for utterance " << u.key;
                warn(m.str());
                ++C.num_fail;
                continue;
              }
              bool any = false;
              for (int t = 0; t < T && !any; ++t) any = (*v)[t] != 0.f;
              if (T > 0 && !any) {
                warn("No features were judged as voiced for utterance " + u.key);
                ++C.num_fail;
                continue;
              }
            }
            keep_idx.push_back(idx[k]);
            rawp.push_back(u.feats.cm ? nullptr : u.feats.Data());
            cmp.push_back(u.feats.cm);
            cmb.push_back(u.feats.cm_bytes);
            any_cm = any_cm || u.feats.cm != nullptr;
            vadp.push_back(v ? v->data() : nullptr);
            rrows.push_back(T);
          }
          idx.swap(keep_idx);
          n = (int)idx.size();
          // device path: raw rows in, CMN + selection + network on the lane's stream, nothing comes back but embeddings
          if (n)
            submitted = w.job.StartFrontEnd(C.eng, opt, C.cur, C.seq, n, rawp.data(), rrows.data(), vadp.data(),
                                            any_cm ? cmp.data() : nullptr, any_cm ? cmb.data() : nullptr);
          if (submitted) ++C.seq;
          if (submitted && any_cm) C.num_cm_device += n;
          if (n && !submitted && any_cm) {
            // the batch cannot go to the device as it is (an utterance cut into several chunks, a mixed batch): the compressed
            // views are expanded here, into the utterances themselves, and the float paths below see what they always saw
            for (int k = 0; k < n; ++k) {
              Utt& u = w.b.utts[idx[k]];
              if (!u.feats.cm) continue;
              Matrix full;
              ExpandCompressedView(u.feats, &full);
              u.feats = std::move(full);
              rawp[k] = u.feats.Data();
            }
            any_cm = false;
            submitted = w.job.StartFrontEnd(C.eng, opt, C.cur, C.seq, n, rawp.data(), rrows.data(), vadp.data());
            if (submitted) ++C.seq;
          }
          if (n && !submitted) {
            ++C.num_fe_host;
            // a batch that does not fit one forward batch: front-end result back to the host, then Start()
            sel_row.clear();
            sel_utt.clear();
            poffs.assign(1, 0);
            std::vector<int32_t> raw_off2(1, 0);
            std::vector<float> raw2;
            for (int k = 0; k < n; ++k) {
              const int T = rrows[k];
              const int32_t base = raw_off2.back();
              for (int t = 0; t < T; ++t)
                if (!vadp[k] || vadp[k][t] != 0.f) {
                  sel_row.push_back(base + t);
                  sel_utt.push_back(k);
                }
              raw2.insert(raw2.end(), rawp[k], rawp[k] + (size_t)T * D);
              raw_off2.push_back(base + T);
              poffs.push_back((int32_t)sel_row.size());
            }
            processed.resize(sel_row.size() * (size_t)D);
            C.eng->FrontEndHost(raw2.data(), raw_off2.data(), n, sel_row.data(), sel_utt.data(), (int)sel_row.size(),
                                 opt.cmn_window, opt.cmn_center, opt.cmn_min_window, processed.data());
            packed.swap(processed);
            offs = poffs;
          }
        }
        if (vad && use_frontend) {
          // the VAD decisions of this batch have done their work (the selection tables are built, or the host path has run)
          std::unique_lock<std::mutex> vlk(vad_mu);
          for (const Utt& u : w.b.utts) vad->Forget(u.key);
        }
        const auto tp1 = now();
        C.t_pack += secs(tp0, tp1);
        if (n) {
          if (!submitted) {
            if (use_frontend) w.job.Start(C.eng, opt, C.cur, C.seq++, packed.data(), offs.data(), n);   // front-end result (host)
            else w.job.StartPtrs(C.eng, opt, C.cur, C.seq++, uptr.data(), urows.data(), n);
          }
          const auto tp2 = now();
          C.t_start += secs(tp1, tp2);
          submitted_batch = true;
          C.cur = (C.cur + 1) % NS;
          Work& oldest = C.work[C.cur];   // the slot the next batch will use
          if (oldest.job.active()) finalize(C, oldest);
          C.t_fin += secs(tp2, now());
        } else {
          w.b = Batch();
        }
      } catch (const std::exception& ex) {
        set_fatal(ex.what());  // keep draining the queue so the reader can finish
      }
    }
    if (!submitted_batch) deliver(g, OutBatch());
  };
  // the batches a consumer still has in flight, oldest first
  auto drain_consumer = [&](Consumer& C) {
    for (int k = 1; k <= NS; ++k) {
      Work& w = C.work[(C.cur + k) % NS];
      if (!w.job.active()) continue;
      if (is_fatal()) {   // still owed to the ordered writer
        deliver(w.g, OutBatch());
        continue;
      }
      try {
        finalize(C, w);
      } catch (const std::exception& ex) {
        set_fatal(ex.what());
        deliver(w.g, OutBatch());
      }
    }
  };

  // ---- several engines: one consumer thread each, the calling thread deals the batches out and writes ------------------------
  std::vector<std::deque<std::pair<long, Batch>>> inbox(NE);
  std::mutex in_mu;
  std::condition_variable in_cv;
  bool in_end = false;
  std::vector<std::thread> workers;
  struct WorkerGuard {   // an exception on the calling thread must not leave joinable threads behind
    std::vector<std::thread>& ts;
    std::mutex& m;
    std::condition_variable& c;
    bool& end;
    ~WorkerGuard() {
      {
        std::unique_lock<std::mutex> lk(m);
        end = true;
        c.notify_all();
      }
      for (std::thread& t : ts)
        if (t.joinable()) t.join();
    }
  } worker_guard{workers, in_mu, in_cv, in_end};
  if (NE > 1) {
    for (int e = 0; e < NE; ++e)
      workers.emplace_back([&, e] {
        Consumer& C = cons[e];
        for (;;) {
          std::pair<long, Batch> item;
          {
            std::unique_lock<std::mutex> lk(in_mu);
            in_cv.wait(lk, [&] { return !inbox[e].empty() || in_end; });
            if (inbox[e].empty()) break;
            item = std::move(inbox[e].front());
            inbox[e].pop_front();
            in_cv.notify_all();
          }
          try {
            process(C, std::move(item.second), item.first);
          } catch (const std::exception& ex) {   // (process catches what the extraction throws; this is for the unexpected)
            set_fatal(ex.what());
          }
        }
        drain_consumer(C);
      });
  }
  long next_g = 0, next_write = 0;
  auto write_ready = [&](bool all) {   // calling thread: whatever is next in table order (all: whatever is there, gaps included)
    for (;;) {
      OutBatch ob;
      {
        std::unique_lock<std::mutex> lk(out_mu);
        auto it = done.begin();
        if (it == done.end() || (!all && it->first != next_write)) return;
        next_write = it->first + 1;
        ob = std::move(it->second);
        done.erase(it);
      }
      write_out(ob);
    }
  };
  auto dispatch = [&](Batch&& b) {
    const long g = next_g++;
    if (NE == 1) {
      process(cons[0], std::move(b), g);
      return;
    }
    const int e = (int)(g % NE);
    {
      std::unique_lock<std::mutex> lk(in_mu);
      in_cv.wait(lk, [&] { return inbox[e].size() < 2; });
      inbox[e].emplace_back(g, std::move(b));
      in_cv.notify_all();
    }
    write_ready(false);
  };
  // With calibration the arithmetic of the whole job is chosen before anything is submitted.  Addressable tables were sampled
  // over their whole list above; a stream is calibrated on its first calibrate_utts utterances: batches are held back until
  // that many have arrived (whatever the batch size, so that the choice - and with it every embedding - does not depend on
  // --batch-frames), and the log line says that the sample is the head of the stream.
  std::deque<Batch> held;
  size_t held_utts = 0;
  for (;;) {
    Batch b;
    const auto tw0 = now();
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return !queue.empty(); });
      b = std::move(queue.front());
      queue.pop_front();
      ++n_consumed;
      cv.notify_all();
    }
    t_wait += secs(tw0, now());
    const bool last = b.last;
    if (!calibrated) {
      held_utts += b.utts.size();
      held.push_back(std::move(b));
      if (held_utts >= (size_t)opt.calibrate_utts || last) {
        calibrated = true;
        if (!is_fatal()) {
          try {
            std::vector<CalibUtt> cu;
            for (const Batch& hb : held)
              for (const Utt& u : hb.utts) cu.push_back(CalibUtt{&u.key, u.feats.Data(), u.feats.rows, u.feats.cols});
            if (!cu.empty()) {
              log("LOG", "calibration sample: the first " + std::to_string(std::min(cu.size(), (size_t)opt.calibrate_utts)) +
                             " utterances of the stream (a stream cannot be sampled any other way)");
              const Engine::Calibration c = CalibrateOnUtterances(engine, opt, cu, log);
              share_choice();
              publish(&c, "the head of a stream");
            } else {
              publish(nullptr, "empty stream");
            }
          } catch (const std::exception& ex) {
            set_fatal(ex.what());
          }
        }
        for (Batch& hb : held) dispatch(std::move(hb));
        held.clear();
      }
    } else {
      dispatch(std::move(b));
    }
    if (last) break;
  }
  if (NE == 1) {
    drain_consumer(cons[0]);
  } else {
    {
      std::unique_lock<std::mutex> lk(in_mu);
      in_end = true;
      in_cv.notify_all();
    }
    // write while the consumers finish; then whatever is left (gaps only after a fatal error)
    for (;;) {
      write_ready(false);
      std::unique_lock<std::mutex> lk(out_mu);
      if (next_write >= next_g) break;
      if (out_cv.wait_for(lk, std::chrono::milliseconds(50)) == std::cv_status::timeout && is_fatal()) break;
    }
    for (std::thread& t : workers) t.join();
    workers.clear();
    write_ready(true);
  }
  for (const Consumer& C : cons) res.num_fail += C.num_fail;
  if (timing) {
    double t_pack = 0, t_start = 0, t_fin = 0;
    for (const Consumer& C : cons) {
      t_pack += C.t_pack;
      t_start += C.t_start;
      t_fin += C.t_fin;
    }
    std::ostringstream m;
    m << "consumer stages" << (NE > 1 ? " (summed over the engines' threads)" : "") << ": wait for reader " << t_wait << " s, pack "
      << t_pack << " s, plan+submit " << t_start << " s, finish+write " << t_fin << " s";
    log("LOG", m.str());
    long n_cm = 0;
    for (const Consumer& C : cons) n_cm += C.num_cm_device;
    if (n_cm) log("LOG", "front-end: " + std::to_string(n_cm) + " utterances went to the device compressed (one byte per element) and were expanded there");
    long n_feh = 0;
    for (const Consumer& C : cons) n_feh += C.num_fe_host;
    if (n_feh) log("LOG", "front-end: " + std::to_string(n_feh) + " batches did not fit one forward batch and took the host round trip");
    // host budget of the table loop: CPU seconds of ALL threads of the process (readers, copy threads, consumers, writer, the
    // runtime's own) between the first batch and the last write - what eight ranks on one node have to share
    struct rusage ru1;
    getrusage(RUSAGE_SELF, &ru1);
    auto tv = [](const timeval& a, const timeval& b) { return (double)(b.tv_sec - a.tv_sec) + 1e-6 * (double)(b.tv_usec - a.tv_usec); };
    const double user = tv(ru0.ru_utime, ru1.ru_utime), sys = tv(ru0.ru_stime, ru1.ru_stime);
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const long n_all = res.num_success + res.num_fail;
    std::ostringstream h;
    h.precision(4);
    h << "host cost of the table loop: " << user << " s user + " << sys << " s system over " << wall << " s = " << (user + sys) / std::max(wall, 1e-9)
      << " cores; " << (n_all ? 1e6 * (user + sys) / n_all : 0.0) << " us of CPU per utterance (" << n_all << " utterances, "
      << (ru1.ru_minflt - ru0.ru_minflt) << " minor page faults)";
    log("LOG", h.str());
    // ... and who spent it: CPU seconds per thread NAME since the threads were created (/proc/self/task/*/schedstat; the HIP
    // runtime's own threads carry the process name)
    std::map<std::string, std::pair<double, int>> by_name;
    if (DIR* d = opendir("/proc/self/task")) {
      while (dirent* de = readdir(d)) {
        if (de->d_name[0] == '.') continue;
        const std::string base = std::string("/proc/self/task/") + de->d_name;
        std::ifstream c(base + "/comm"), st(base + "/schedstat");
        std::string name;
        unsigned long long ns = 0;
        if (std::getline(c, name) && (st >> ns)) {
          by_name[name].first += 1e-9 * (double)ns;
          by_name[name].second += 1;
        }
      }
      closedir(d);
    }
    std::ostringstream t;
    t.precision(3);
    t << "CPU seconds by thread name (threads):";
    for (const auto& kv : by_name) t << " " << kv.first << " " << kv.second.first << " (" << kv.second.second << ")";
    log("LOG", t.str());
  }
  {
    std::unique_lock<std::mutex> lk(mu);
    stop = true;   // every reader has delivered its last batch; the flag releases any that still waits for work
    cv.notify_all();
  }
  for (std::thread& t : threads) t.join();
  writer.Close();
  if (!fatal.empty()) throw std::runtime_error(fatal);
  if (!reader_error.empty()) throw KioError(reader_error);
  res.num_fail += num_fail_read;
  res.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return res;
}

TableExtractResult RunTableCompute(Engine* engine, int max_batch_rows, bool apply_exp, const std::string& feat_rspec,
                                   const std::string& mat_wspec, const LogFn& log) {
  TableExtractResult res;
  if (!engine->frame_mode()) throw EngineError("nnet3-compute needs a frame-level output node (this one follows the pooling)");
  const int D = engine->info().input_dim, E = engine->info().output_dim;
  TableWriter writer(mat_wspec);
  SequentialMatrixReader rd(feat_rspec);
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<Utt> batch;
  std::vector<float> packed, out;
  std::vector<int32_t> offs;
  long rows = 0;
  auto flush = [&]() {
    if (batch.empty()) return;
    packed.resize((size_t)rows * D);
    offs.assign(1, 0);
    size_t r = 0;
    for (const Utt& u : batch) {
      memcpy(&packed[r * D], u.feats.Data(), (size_t)u.feats.rows * D * 4);
      r += u.feats.rows;
      offs.push_back((int32_t)r);
    }
    out.resize((size_t)rows * E);
    engine->ForwardHost(packed.data(), offs.data(), (int)batch.size(), out.data());
    for (size_t i = 0; i < batch.size(); ++i) {
      Matrix m;
      m.rows = batch[i].feats.rows;
      m.cols = E;
      m.data.assign(out.begin() + (size_t)offs[i] * E, out.begin() + (size_t)offs[i + 1] * E);
      if (apply_exp)
        for (float& v : m.data) v = expf(v);
      writer.WriteMat(batch[i].key, m);
      res.frames += m.rows;
      ++res.num_success;
    }
    batch.clear();
    rows = 0;
  };
  std::string key, e;
  Matrix m;
  while (rd.Next(&key, &m, &e)) {
    if (!e.empty()) {
      log("WARNING", "failed to read features for " + key + ": " + e);
      ++res.num_fail;
      continue;
    }
    if (m.rows == 0) {
      log("WARNING", "Zero-length utterance: " + key);
      ++res.num_fail;
      continue;
    }
    if (m.cols != D) {
      std::ostringstream s;
      s << "feature dimension " << m.cols << " of utterance " << key << " does not match the model's " << D;
      log("WARNING", s.str());
      ++res.num_fail;
      continue;
    }
    Utt u;
    u.key = key;
    u.feats = std::move(m);
    rows += u.feats.rows;
    batch.push_back(std::move(u));
    if (rows >= max_batch_rows) flush();
  }
  flush();
  res.reader_status = rd.Close();
  writer.Close();
  res.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return res;
}

}  // namespace xv
