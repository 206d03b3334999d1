// Chunk loop + weighted average.  See extractor.h.
#include "extractor.h"
#include "knobs.h"

#include <pthread.h>
#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <functional>

#include <math.h>
#include <string.h>

#include <sstream>

namespace xv {

namespace {
// A few helper threads for the bulk host copies of table jobs (started on first use, alive until exit).  ParallelFor splits
// [0, n) into contiguous ranges, the caller takes one of them and returns when all are done.  XVEC_DEBUG=copy_threads=n (default 3
// helpers, 0 = the caller alone).
class CopyPool {
 public:
  static CopyPool& Get() {
    static CopyPool* p = new CopyPool();   // never destroyed: the threads may outlive static destructors otherwise
    return *p;
  }
  void Run(int n, const std::function<void(int, int)>& fn) {
    const int parts = std::min((int)workers_.size() + 1, n);
    if (parts <= 1 || n < 8) {
      fn(0, n);
      return;
    }
    std::unique_lock<std::mutex> run_lock(run_mu_);   // one ParallelFor at a time
    {
      std::unique_lock<std::mutex> lk(mu_);
      fn_ = &fn;
      n_ = n;
      parts_ = parts;
      next_ = 1;          // part 0 is the caller's
      pending_ = parts - 1;
      ++gen_;
    }
    cv_.notify_all();
    fn(0, n / parts);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  CopyPool() {
    int t = 3;
    t = std::max(0, std::min(15, DebugKnobInt("copy_threads", t)));
    for (int i = 0; i < t; ++i) workers_.emplace_back([this] { Loop(); });
    for (std::thread& w : workers_) w.detach();
  }
  void Loop() {
    (void)pthread_setname_np(pthread_self(), "xv-copy");
    unsigned long seen = 0;
    for (;;) {
      int part;
      const std::function<void(int, int)>* fn;
      int n, parts;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen && fn_ && next_ < parts_; });
        part = next_++;
        if (next_ >= parts_) seen = gen_;
        fn = fn_;
        n = n_;
        parts = parts_;
      }
      (*fn)((int)((long)n * part / parts), (int)((long)n * (part + 1) / parts));
      std::unique_lock<std::mutex> lk(mu_);
      if (--pending_ == 0) done_.notify_all();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_, run_mu_;
  std::condition_variable cv_, done_;
  const std::function<void(int, int)>* fn_ = nullptr;
  int n_ = 0, parts_ = 0, next_ = 0, pending_ = 0;
  unsigned long gen_ = 0;
};
void ParallelFor(int n, const std::function<void(int, int)>& fn) { CopyPool::Get().Run(n, fn); }
}  // namespace


bool PlanChunks(int utt, int num_rows, int chunk_size, int min_chunk_size, bool pad_input, int min_net_frames,
                std::vector<Chunk>* out, std::string* why) {
  if (num_rows == 0) {
    *why = "Zero-length utterance";
    return false;
  }
  int this_chunk = chunk_size;
  if (!pad_input && num_rows < min_chunk_size) {
    std::ostringstream m;
    m << "Minimum chunk size of " << min_chunk_size << " is greater than the number of rows in utterance";
    *why = m.str();
    return false;
  } else if (num_rows < chunk_size) {
    this_chunk = num_rows;
  } else if (chunk_size <= 0) {
    this_chunk = num_rows;
  }
  const int num_chunks = (int)ceil(num_rows / (double)this_chunk);
  const size_t before = out->size();
  for (int ci = 0; ci < num_chunks; ++ci) {
    const int len = std::min(this_chunk, num_rows - ci * this_chunk);
    if (!pad_input && len < min_chunk_size) continue;  // short tail: skipped, carries no weight
    Chunk c;
    c.utt = utt;
    c.start = ci * this_chunk;
    c.len = len;
    c.left_pad = c.right_pad = 0;
    if (pad_input && len < min_chunk_size) {
      c.left_pad = (min_chunk_size - len) / 2;
      c.right_pad = min_chunk_size - len - c.left_pad;
    }
    if (c.len + c.left_pad + c.right_pad < min_net_frames) {
      std::ostringstream m;
      m << "chunk of " << (c.len + c.left_pad + c.right_pad) << " frames is shorter than the network context ("
        << min_net_frames << " frames needed)";
      *why = m.str();
      out->resize(before);
      return false;
    }
    out->push_back(c);
  }
  if (out->size() == before) {
    *why = "no chunk of the utterance is long enough";
    return false;
  }
  return true;
}

void ExtractUtterances(Engine* eng, const ExtractOptions& opt, const float* feats, const int32_t* row_offsets, int n_utts,
                       float* out, int32_t* ok, std::vector<std::string>* why) {
  const int D = eng->info().input_dim, E = eng->info().output_dim;
  std::vector<Chunk> chunks;
  if (why) why->assign(n_utts, std::string());
  for (int u = 0; u < n_utts; ++u) {
    std::string reason;
    ok[u] = PlanChunks(u, row_offsets[u + 1] - row_offsets[u], opt.chunk_size, opt.min_chunk_size, opt.pad_input,
                       eng->info().min_frames, &chunks, &reason)
                ? 1
                : 0;
    if (!ok[u] && why) (*why)[u] = reason;
  }
  std::vector<float> tot(n_utts, 0.f);
  for (int u = 0; u < n_utts; ++u)
    if (ok[u]) memset(out + (size_t)u * E, 0, (size_t)E * 4);

  std::vector<float> pack;
  std::vector<int32_t> offs;
  std::vector<float> emb;
  size_t i = 0;
  while (i < chunks.size()) {
    // assemble one batch of chunks
    size_t j = i;
    long rows = 0;
    while (j < chunks.size() && (j == i || (rows + chunks[j].len + chunks[j].left_pad + chunks[j].right_pad <= opt.max_batch_rows &&
                                            (int)(j - i) < opt.max_batch_chunks))) {
      rows += chunks[j].len + chunks[j].left_pad + chunks[j].right_pad;
      ++j;
    }
    const int B = (int)(j - i);
    pack.resize((size_t)rows * D);
    offs.assign(1, 0);
    size_t r = 0;
    for (size_t k = i; k < j; ++k) {
      const Chunk& c = chunks[k];
      const float* src = feats + ((size_t)row_offsets[c.utt] + c.start) * D;
      for (int p = 0; p < c.left_pad; ++p, ++r) memcpy(&pack[r * D], src, (size_t)D * 4);
      memcpy(&pack[r * D], src, (size_t)c.len * D * 4);
      r += c.len;
      for (int p = 0; p < c.right_pad; ++p, ++r) memcpy(&pack[r * D], src + (size_t)(c.len - 1) * D, (size_t)D * 4);
      offs.push_back((int32_t)r);
    }
    emb.resize((size_t)B * E);
    eng->ForwardHost(pack.data(), offs.data(), B, emb.data());
    // xvector_avg.AddVec(len, xvector) in fp32, like the reference binary
    for (size_t k = i; k < j; ++k) {
      const Chunk& c = chunks[k];
      float* dst = out + (size_t)c.utt * E;
      const float* e = &emb[(k - i) * E];
      const float w = (float)c.len;
      for (int d = 0; d < E; ++d) dst[d] += w * e[d];
      tot[c.utt] += w;
    }
    i = j;
  }
  for (int u = 0; u < n_utts; ++u)
    if (ok[u]) {
      const float s = 1.0f / tot[u];
      float* dst = out + (size_t)u * E;
      for (int d = 0; d < E; ++d) dst[d] *= s;
    }
}

void ExtractJob::Start(Engine* eng, const ExtractOptions& opt, int slot, long seq, const float* feats,
                       const int32_t* row_offsets, int n_utts) {
  eng_ = eng;
  opt_ = opt;
  slot_ = slot;
  n_utts_ = n_utts;
  feats_ = feats;
  row_offsets_ = row_offsets;
  async_ = false;
  chunks_.clear();
  ok_.assign(n_utts, 0);
  why_.assign(n_utts, std::string());
  long rows = 0;
  for (int u = 0; u < n_utts; ++u) {
    std::string reason;
    const size_t before = chunks_.size();
    ok_[u] = PlanChunks(u, row_offsets[u + 1] - row_offsets[u], opt.chunk_size, opt.min_chunk_size, opt.pad_input,
                        eng->info().min_frames, &chunks_, &reason)
                 ? 1
                 : 0;
    if (!ok_[u]) why_[u] = reason;
    for (size_t k = before; k < chunks_.size(); ++k) rows += chunks_[k].len + chunks_[k].left_pad + chunks_[k].right_pad;
  }
  if (chunks_.empty() || rows > opt.max_batch_rows || (int)chunks_.size() > opt.max_batch_chunks) return;  // Finish() does it
  const int D = eng->info().input_dim;
  float* pack = eng->HostFeats(slot, (size_t)rows);
  std::vector<int32_t> offs(1, 0);
  size_t r = 0;
  for (const Chunk& c : chunks_) {
    const float* src = feats + ((size_t)row_offsets[c.utt] + c.start) * D;
    for (int p = 0; p < c.left_pad; ++p, ++r) memcpy(pack + r * D, src, (size_t)D * 4);
    memcpy(pack + r * D, src, (size_t)c.len * D * 4);
    r += c.len;
    for (int p = 0; p < c.right_pad; ++p, ++r) memcpy(pack + r * D, src + (size_t)(c.len - 1) * D, (size_t)D * 4);
    offs.push_back((int32_t)r);
  }
  eng->SubmitHost(slot, seq, offs.data(), (int)chunks_.size());
  async_ = true;
}

void ExtractJob::StartPtrs(Engine* eng, const ExtractOptions& opt, int slot, long seq, const float* const* utt,
                           const int32_t* rows, int n_utts) {
  eng_ = eng;
  opt_ = opt;
  slot_ = slot;
  n_utts_ = n_utts;
  feats_ = nullptr;
  row_offsets_ = nullptr;
  async_ = false;
  utt_ptr_.assign(utt, utt + n_utts);
  utt_rows_.assign(rows, rows + n_utts);
  chunks_.clear();
  ok_.assign(n_utts, 0);
  why_.assign(n_utts, std::string());
  long total = 0;
  for (int u = 0; u < n_utts; ++u) {
    std::string reason;
    const size_t before = chunks_.size();
    ok_[u] = PlanChunks(u, rows[u], opt.chunk_size, opt.min_chunk_size, opt.pad_input, eng->info().min_frames, &chunks_, &reason) ? 1 : 0;
    if (!ok_[u]) why_[u] = reason;
    for (size_t k = before; k < chunks_.size(); ++k) total += chunks_[k].len + chunks_[k].left_pad + chunks_[k].right_pad;
  }
  if (chunks_.empty() || total > opt.max_batch_rows || (int)chunks_.size() > opt.max_batch_chunks) return;  // Finish() does it
  const int D = eng->info().input_dim;
  float* pack = eng->HostFeats(slot, (size_t)total);
  std::vector<int32_t> offs(1, 0);
  for (const Chunk& c : chunks_) offs.push_back(offs.back() + c.left_pad + c.len + c.right_pad);
  // the one host copy per byte of a table job (reader's buffers -> pinned staging), spread over a few threads: done by the
  // consumer thread alone it was what the loop waited for once the readers were parallel (0.19 s of 0.34 s per 80 000
  // utterances)
  const int n_chunks = (int)chunks_.size();
  ParallelFor(n_chunks, [&](int k0, int k1) {
    for (int k = k0; k < k1; ++k) {
      const Chunk& c = chunks_[k];
      const float* src = utt[c.utt] + (size_t)c.start * D;
      size_t r = (size_t)offs[k];
      for (int p = 0; p < c.left_pad; ++p, ++r) memcpy(pack + r * D, src, (size_t)D * 4);
      memcpy(pack + r * D, src, (size_t)c.len * D * 4);
      r += c.len;
      for (int p = 0; p < c.right_pad; ++p, ++r) memcpy(pack + r * D, src + (size_t)(c.len - 1) * D, (size_t)D * 4);
    }
  });
  eng->SubmitHost(slot, seq, offs.data(), (int)chunks_.size());
  async_ = true;
}

bool ExtractJob::StartFrontEnd(Engine* eng, const ExtractOptions& opt, int slot, long seq, int n_utts, const float* const* raw,
                               const int32_t* raw_rows, const float* const* vad, const uint8_t* const* cm,
                               const size_t* cm_bytes) {
  const int D = eng->info().input_dim;
  std::vector<Chunk> chunks;
  std::vector<int32_t> ok(n_utts, 0), first_chunk(n_utts, 0), n_chunk(n_utts, 0);
  std::vector<std::string> why(n_utts);
  long total_raw = 0, total_kept = 0;   // total_kept: rows of all chunks, replicated edge rows of padded chunks included
  for (int u = 0; u < n_utts; ++u) {
    int k = raw_rows[u];
    if (vad[u]) {
      k = 0;
      for (int t = 0; t < raw_rows[u]; ++t) k += vad[u][t] != 0.f;
    }
    const size_t before = chunks.size();
    std::string reason;
    ok[u] = PlanChunks(u, k, opt.chunk_size, opt.min_chunk_size, opt.pad_input, eng->info().min_frames, &chunks, &reason) ? 1 : 0;
    if (!ok[u]) {
      why[u] = reason;
      continue;
    }
    // An utterance may be cut into several chunks (a five-minute recording at --chunk-size=10000: the recipes' own setting and
    // the recipes' own data) and a short chunk may be padded by edge replication: both are SELECTIONS of the utterance's kept
    // rows - consecutive ranges, an edge row listed several times - so the device front-end takes them as they are.  (Rounds
    // 2-5 sent such batches back through the host: 25 M raw frames/s where single-chunk utterances ran at 100 M.)
    first_chunk[u] = (int32_t)before;
    n_chunk[u] = (int32_t)(chunks.size() - before);
    for (size_t c = before; c < chunks.size(); ++c) total_kept += chunks[c].left_pad + chunks[c].len + chunks[c].right_pad;
    total_raw += raw_rows[u];
  }
  if (chunks.empty() || total_kept > opt.max_batch_rows || (int)chunks.size() > opt.max_batch_chunks) return false;
  // compressed input: all or nothing (a mixed batch goes the host way - decided before this job takes any state)
  bool all_cm = cm != nullptr && cm_bytes != nullptr;
  for (int u = 0; u < n_utts && all_cm; ++u)
    if (ok[u] && !cm[u]) all_cm = false;
  size_t cm_total = 0;
  int max_rows = 0;
  for (int u = 0; u < n_utts; ++u) {
    if (!ok[u]) continue;
    if (!all_cm && !raw[u]) return false;   // a compressed view without floats in a batch that cannot go compressed
    if (all_cm) cm_total += (cm_bytes[u] + 15) & ~(size_t)15;
    max_rows = std::max(max_rows, raw_rows[u]);
  }
  eng_ = eng;
  opt_ = opt;
  slot_ = slot;
  n_utts_ = n_utts;
  feats_ = nullptr;
  row_offsets_ = nullptr;
  chunks_ = chunks;
  ok_ = ok;
  why_ = why;
  float* buf = eng->HostFeats(slot, all_cm ? (cm_total + (size_t)D * 4 - 1) / ((size_t)D * 4) : (size_t)total_raw);
  std::vector<int32_t> raw_off(1, 0), sel_row, sel_utt, offs(1, 0);
  std::vector<int64_t> cm_off;
  std::vector<int32_t> kidx;
  sel_row.reserve(total_kept);
  sel_utt.reserve(total_kept);
  int j = 0;   // index among the utterances that enter the device batch
  size_t cm_pos = 0;
  for (int u = 0; u < n_utts; ++u) {
    if (!ok[u]) continue;
    const int base = raw_off.back();
    if (all_cm) {
      memcpy((uint8_t*)buf + cm_pos, cm[u], cm_bytes[u]);
      cm_off.push_back((int64_t)cm_pos);
      cm_pos += (cm_bytes[u] + 15) & ~(size_t)15;
    } else {
      memcpy(buf + (size_t)base * D, raw[u], (size_t)raw_rows[u] * D * 4);
    }
    // kept rows of the utterance in order (identity without a VAD table), then chunk by chunk
    kidx.clear();
    if (vad[u])
      for (int t = 0; t < raw_rows[u]; ++t)
        if (vad[u][t] != 0.f) kidx.push_back(t);
    auto row_of = [&](int i) { return base + (vad[u] ? kidx[i] : i); };
    for (int c = first_chunk[u]; c < first_chunk[u] + n_chunk[u]; ++c) {
      const Chunk& ch = chunks[c];
      for (int p = 0; p < ch.left_pad; ++p) sel_row.push_back(row_of(ch.start));
      for (int i = ch.start; i < ch.start + ch.len; ++i) sel_row.push_back(row_of(i));
      for (int p = 0; p < ch.right_pad; ++p) sel_row.push_back(row_of(ch.start + ch.len - 1));
      sel_utt.resize(sel_row.size(), j);
      offs.push_back((int32_t)sel_row.size());
    }
    raw_off.push_back(base + raw_rows[u]);
    ++j;
  }
  Engine::FrontEndJob fe;
  fe.raw_off = raw_off.data();
  fe.n_utts = j;
  fe.sel_row = sel_row.data();
  fe.sel_utt = sel_utt.data();
  fe.n_out = (int)sel_row.size();
  fe.cmn_window = opt.cmn_window;
  fe.center = opt.cmn_center;
  fe.min_window = opt.cmn_min_window;
  if (all_cm) {
    fe.cm_off = cm_off.data();
    fe.cm_bytes = cm_total;
    fe.max_rows = max_rows;
  }
  eng->SubmitHost(slot, seq, offs.data(), (int)chunks_.size(), &fe);
  async_ = true;
  return true;
}

void ExtractJob::Finish(float* out, int32_t* ok, std::vector<std::string>* why) {
  Engine* eng = eng_;
  eng_ = nullptr;
  if (!eng) throw EngineError("ExtractJob::Finish without Start");
  if (!async_) {
    if (!feats_ && !utt_ptr_.empty()) {   // StartPtrs: the synchronous path wants packed rows
      const int D = eng->info().input_dim;
      fallback_offs_.assign(1, 0);
      for (int u = 0; u < n_utts_; ++u) fallback_offs_.push_back(fallback_offs_.back() + utt_rows_[u]);
      fallback_pack_.resize((size_t)fallback_offs_.back() * D);
      for (int u = 0; u < n_utts_; ++u)
        if (utt_rows_[u] > 0)
          memcpy(&fallback_pack_[(size_t)fallback_offs_[u] * D], utt_ptr_[u], (size_t)utt_rows_[u] * D * 4);
      feats_ = fallback_pack_.data();
      row_offsets_ = fallback_offs_.data();
    }
    ExtractUtterances(eng, opt_, feats_, row_offsets_, n_utts_, out, ok, why);
    return;
  }
  const float* emb = eng->WaitHost(slot_);
  const int E = eng->info().output_dim;
  for (int u = 0; u < n_utts_; ++u) {
    ok[u] = ok_[u];
    if (ok[u]) memset(out + (size_t)u * E, 0, (size_t)E * 4);
  }
  if (why) *why = why_;
  std::vector<float> tot(n_utts_, 0.f);
  for (size_t k = 0; k < chunks_.size(); ++k) {
    // xvector_avg.AddVec(len, xvector) in fp32, like the reference binary
    const Chunk& c = chunks_[k];
    float* dst = out + (size_t)c.utt * E;
    const float* e = emb + k * E;
    const float w = (float)c.len;
    for (int d = 0; d < E; ++d) dst[d] += w * e[d];
    tot[c.utt] += w;
  }
  for (int u = 0; u < n_utts_; ++u)
    if (ok[u]) {
      const float s = 1.0f / tot[u];
      float* dst = out + (size_t)u * E;
      for (int d = 0; d < E; ++d) dst[d] *= s;
    }
}

}  // namespace xv
