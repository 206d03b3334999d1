// kio - Kaldi table / object I/O without Kaldi.  See kio.h for what it replaces and why.
#include "kio.h"
#include "knobs.h"

#include <fcntl.h>

#include <sys/mman.h>
#include <sys/stat.h>

#include <ctype.h>
#include <errno.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>

#include <poll.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <map>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace xv {

static std::string Trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && isspace((unsigned char)s[a])) ++a;
  while (b > a && isspace((unsigned char)s[b - 1])) --b;
  return s.substr(a, b - a);
}

// ------------------------------------------------------------------------------------- Input
// One producer (this thread: read(2) on the pipe, whole blocks), one consumer (the parser).  The consumer touches the lock only
// when it moves from one block to the next.
class PipeDrain {
 public:
  static constexpr size_t kBlock = 1 << 20;
  static constexpr long kBlocks = 8;
  explicit PipeDrain(int fd) : fd_(fd) {
    for (long i = 0; i < kBlocks; ++i) blk_[i].reset(new unsigned char[kBlock]);
    th_ = std::thread([this] { Run(); });
  }
  ~PipeDrain() {
    {
      std::unique_lock<std::mutex> lk(mu_);
      stop_ = true;
      cv_.notify_all();
    }
    if (th_.joinable()) th_.join();
  }
  int Peek() { return (pos_ < cur_len_ || Advance()) ? cur_[pos_] : -1; }
  int Get() { return (pos_ < cur_len_ || Advance()) ? cur_[pos_++] : -1; }
  size_t Read(void* dst, size_t n) {   // up to n bytes, fewer only at the end of the input
    size_t got = 0;
    unsigned char* d = (unsigned char*)dst;
    while (got < n) {
      if (pos_ >= cur_len_ && !Advance()) break;
      const size_t k = std::min(n - got, cur_len_ - pos_);
      memcpy(d + got, cur_ + pos_, k);
      pos_ += k;
      got += k;
    }
    return got;
  }

 private:
  bool Advance() {   // the current block is used up: release it, wait for the next
    std::unique_lock<std::mutex> lk(mu_);
    if (have_cur_) {
      ++tail_;
      have_cur_ = false;
      cv_.notify_all();
    }
    cv_.wait(lk, [&] { return tail_ < head_ || eof_; });
    if (tail_ >= head_) {
      cur_len_ = pos_ = 0;
      return false;
    }
    cur_ = blk_[tail_ % kBlocks].get();
    cur_len_ = len_[tail_ % kBlocks];
    pos_ = 0;
    have_cur_ = true;
    return true;
  }
  void Run() {
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return head_ - tail_ < kBlocks || stop_; });
        if (stop_) return;
      }
      unsigned char* b = blk_[head_ % kBlocks].get();
      size_t have = 0;
      bool end = false;
      // fill the block (a pipe hands over what the producer has written so far, at most its buffer); give up the wait every
      // 50 ms to see whether the reader was closed early (a calibration sample takes only the head of a stream)
      while (have < kBlock) {
        struct pollfd pf = {fd_, POLLIN, 0};
        const int pr = poll(&pf, 1, 50);
        if (pr == 0) {
          std::unique_lock<std::mutex> lk(mu_);
          if (stop_) return;
          if (have) break;   // hand over what is there rather than sit on it
          continue;
        }
        if (pr < 0 && errno == EINTR) continue;
        const ssize_t r = read(fd_, b + have, kBlock - have);
        if (r < 0 && (errno == EINTR || errno == EAGAIN)) continue;
        if (r <= 0) {
          end = true;
          break;
        }
        have += (size_t)r;
      }
      std::unique_lock<std::mutex> lk(mu_);
      if (have) {
        len_[head_ % kBlocks] = have;
        ++head_;
      }
      if (end) eof_ = true;
      cv_.notify_all();
      if (end || stop_) return;
    }
  }
  int fd_;
  std::thread th_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::unique_ptr<unsigned char[]> blk_[kBlocks];
  size_t len_[kBlocks] = {};
  long head_ = 0, tail_ = 0;   // blocks produced / released (under mu_)
  bool eof_ = false, stop_ = false;
  // consumer side
  const unsigned char* cur_ = nullptr;
  size_t cur_len_ = 0, pos_ = 0;
  bool have_cur_ = false;
};

Input::Input() = default;

Input::~Input() {
  try {
    Close();
  } catch (...) {
  }
}

void Input::Open(const std::string& rx_in, bool drain_pipe) {
  Close();
  std::string rx = Trim(rx_in);
  name_ = rx;
  if (rx.empty() || rx == "-") {
    f_ = stdin;
    is_stdin_ = true;
    return;
  }
  if (rx.back() == '|') {
    std::string cmd = Trim(rx.substr(0, rx.size() - 1));
    f_ = popen(cmd.c_str(), "r");
    if (!f_) throw KioError("failed to start input pipe: " + cmd);
    is_pipe_ = true;
    // a feature pipe (extract_xvectors_new.sh:79) carries gigabytes: the largest pipe buffer an unprivileged process may ask
    // for (1 MiB by default) instead of 64 KiB - fewer hand-overs between the producer and this reader
#ifdef F_SETPIPE_SZ
    (void)fcntl(fileno(f_), F_SETPIPE_SZ, 1 << 20);
#endif
    // (stdio buffer left small: an fread of a whole matrix then goes from the pipe straight into its destination - one copy per
    // byte - instead of through the FILE buffer)
    if (drain_pipe) {
      if (DebugKnobInt("pipe_drain", 1) != 0) drain_.reset(new PipeDrain(fileno(f_)));   // nothing was read through f_ yet
    }
    return;
  }
  // "file:offset"
  std::string path = rx;
  long offset = -1;
  size_t c = rx.rfind(':');
  if (c != std::string::npos && c + 1 < rx.size()) {
    bool digits = true;
    for (size_t i = c + 1; i < rx.size(); ++i) digits = digits && isdigit((unsigned char)rx[i]);
    if (digits) {
      path = rx.substr(0, c);
      offset = strtol(rx.c_str() + c + 1, nullptr, 10);
    }
  }
  // A regular file is MAPPED (read-only) and read through the memory backend: the index pass of a table job skips from header
  // to header - fseek + ftell + fstat and a refill of the stdio buffer per utterance, 3 us of a 16 us host budget - and a
  // sequential reader copies straight out of the page cache.  XVEC_DEBUG=mmap=0: stdio, as before.
  {
    const int fd = DebugKnobInt("mmap", 1) == 0 ? -1 : open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd >= 0) {
      struct stat st;
      if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
        if (p != MAP_FAILED) {
          close(fd);
          if (offset > (long)st.st_size) {
            (void)munmap(p, (size_t)st.st_size);
            throw KioError("cannot seek to offset in " + rx);
          }
          mem_ = (const unsigned char*)p;
          mem_n_ = (size_t)st.st_size;
          mem_pos_ = offset > 0 ? (size_t)offset : 0;
          mapped_ = true;
          return;
        }
      }
      close(fd);
    }
  }
  f_ = fopen(path.c_str(), "rb");
  if (!f_) throw KioError("cannot open " + path + " for reading: " + strerror(errno));
  if (offset > 0 && fseek(f_, offset, SEEK_SET) != 0) {
    fclose(f_);
    f_ = nullptr;
    throw KioError("cannot seek to offset in " + rx);
  }
}

void Input::Seek(long offset) {
  if (mem_) {
    if (mapped_ && (offset < 0 || (size_t)offset > mem_n_)) throw KioError("cannot seek in " + name_);
    mem_pos_ = (size_t)offset;
    return;
  }
  if (!f_ || is_pipe_ || is_stdin_ || fseek(f_, offset, SEEK_SET) != 0) throw KioError("cannot seek in " + name_);
}

bool Input::IsRegularFile() const {
  if (mapped_) return true;
  if (!f_ || is_pipe_ || is_stdin_) return false;
  struct stat st;
  return fstat(fileno(f_), &st) == 0 && S_ISREG(st.st_mode);
}

long Input::FileTell() {
  if (!IsRegularFile()) throw KioError("not a regular file: " + name_);
  if (mapped_) return (long)mem_pos_;
  return ftell(f_);
}

void Input::Skip(long n) {
  if (mapped_) {
    if (n < 0 || mem_pos_ + (size_t)n > mem_n_) throw KioError("unexpected end of file in " + name_);
    mem_pos_ += (size_t)n;
    return;
  }
  if (!IsRegularFile() || fseek(f_, n, SEEK_CUR) != 0) throw KioError("cannot seek in " + name_);
  // fseek moves past the end of a truncated file without a word: the index pass of a table must notice where the sequential
  // reader would have ("unexpected end of file"), not a whole job later
  struct stat st;
  const long pos = ftell(f_);
  if (pos >= 0 && fstat(fileno(f_), &st) == 0 && pos > (long)st.st_size) throw KioError("unexpected end of file in " + name_);
}

void Input::OpenMemory(const void* data, size_t n) {
  Close();
  mem_ = (const unsigned char*)data;
  mem_n_ = n;
  mem_pos_ = 0;
  name_ = "<memory>";
}

int Input::Close() {
  int status = 0;
  drain_.reset();   // stops and joins the drain thread before the pipe is closed
  if (f_) {
    if (is_pipe_) {
      int st = pclose(f_);
      status = (st == -1) ? -1 : (WIFEXITED(st) ? WEXITSTATUS(st) : 128);
    } else if (!is_stdin_) {
      fclose(f_);
    }
  }
  f_ = nullptr;
  is_pipe_ = is_stdin_ = false;
  if (mapped_ && mem_) (void)munmap((void*)mem_, mem_n_);
  mapped_ = false;
  mem_ = nullptr;
  mem_n_ = mem_pos_ = 0;
  return status;
}

int Input::Peek() {
  if (drain_) return drain_->Peek();
  if (mem_) return mem_pos_ < mem_n_ ? mem_[mem_pos_] : -1;
  if (!f_) return -1;
  int c = fgetc(f_);
  if (c == EOF) return -1;
  ungetc(c, f_);
  return c;
}

int Input::Get() {
  if (drain_) return drain_->Get();
  if (mem_) return mem_pos_ < mem_n_ ? mem_[mem_pos_++] : -1;
  if (!f_) return -1;
  int c = fgetc(f_);
  return c == EOF ? -1 : c;
}

void Input::Read(void* dst, size_t n) {
  if (n == 0) return;
  if (drain_) {
    if (drain_->Read(dst, n) != n) throw KioError("unexpected end of file in " + name_);
    return;
  }
  if (mem_) {
    if (mem_pos_ + n > mem_n_) throw KioError(std::string(mapped_ ? "unexpected end of file in " : "unexpected end of data in ") + name_);
    memcpy(dst, mem_ + mem_pos_, n);
    mem_pos_ += n;
    return;
  }
  if (!f_ || fread(dst, 1, n, f_) != n) throw KioError("unexpected end of file in " + name_);
}

size_t Input::ReadUpTo(void* dst, size_t n) {
  if (drain_) return drain_->Read(dst, n);
  if (mem_) {
    const size_t got = std::min(n, mem_n_ - mem_pos_);
    memcpy(dst, mem_ + mem_pos_, got);
    mem_pos_ += got;
    return got;
  }
  return f_ ? fread(dst, 1, n, f_) : 0;
}

// ------------------------------------------------------------------------------------- Output
Output::~Output() {
  try {
    Close();
  } catch (...) {
  }
}

void Output::Open(const std::string& wx_in) {
  Close();
  std::string wx = Trim(wx_in);
  name_ = wx;
  pos_ = 0;
  if (wx.empty() || wx == "-") {
    f_ = stdout;
    is_stdout_ = true;
    return;
  }
  if (wx[0] == '|') {
    std::string cmd = Trim(wx.substr(1));
    f_ = popen(cmd.c_str(), "w");
    if (!f_) throw KioError("failed to start output pipe: " + cmd);
    is_pipe_ = true;
    return;
  }
  f_ = fopen(wx.c_str(), "wb");
  if (!f_) throw KioError("cannot open " + wx + " for writing: " + strerror(errno));
}

int Output::Close() {
  int status = 0;
  if (f_) {
    if (is_pipe_) {
      int st = pclose(f_);
      status = (st == -1) ? -1 : (WIFEXITED(st) ? WEXITSTATUS(st) : 128);
    } else if (is_stdout_) {
      fflush(f_);
    } else {
      if (fclose(f_) != 0) status = -1;
    }
  }
  f_ = nullptr;
  is_pipe_ = is_stdout_ = false;
  return status;
}

void Output::Write(const void* src, size_t n) {
  if (!f_) throw KioError("write to closed output " + name_);
  if (n && fwrite(src, 1, n, f_) != n) throw KioError("write failed on " + name_);
  pos_ += (int64_t)n;
}

void Output::Flush() {
  if (f_) fflush(f_);
}

// ------------------------------------------------------------------------------------- objects
bool ReadBinaryHeader(Input& in) {
  if (in.Peek() == 0) {
    in.Get();
    if (in.Get() != 'B') throw KioError("bad binary header in " + in.Name());
    return true;
  }
  return false;
}

void ReadToken(Input& in, bool binary, std::string* tok) {
  (void)binary;
  tok->clear();
  int c;
  while ((c = in.Peek()) >= 0 && isspace(c)) in.Get();
  while ((c = in.Peek()) >= 0 && !isspace(c)) tok->push_back((char)in.Get());
  if (tok->empty()) throw KioError("expected a token, got end of input in " + in.Name());
  if (c >= 0) in.Get();  // exactly one separator is consumed (the binary payload may follow)
}

void ExpectToken(Input& in, bool binary, const char* tok) {
  std::string t;
  ReadToken(in, binary, &t);
  if (t != tok) throw KioError(std::string("expected token ") + tok + ", got " + t);
}

static std::string ReadTextWord(Input& in) {
  std::string w;
  int c;
  while ((c = in.Peek()) >= 0 && isspace(c)) in.Get();
  while ((c = in.Peek()) >= 0 && !isspace(c)) w.push_back((char)in.Get());
  if (w.empty()) throw KioError("expected a value, got end of input in " + in.Name());
  return w;
}

int32_t ReadInt32(Input& in, bool binary) {
  if (binary) {
    int sz = in.Get();
    if (sz != 4) throw KioError("expected int32 (size byte 4) in " + in.Name());
    int32_t v;
    in.Read(&v, 4);
    return v;
  }
  std::string w = ReadTextWord(in);
  return (int32_t)strtol(w.c_str(), nullptr, 10);
}

double ReadFloatOrDouble(Input& in, bool binary) {
  if (binary) {
    int sz = in.Get();
    if (sz == 4) {
      float v;
      in.Read(&v, 4);
      return v;
    }
    if (sz == 8) {
      double v;
      in.Read(&v, 8);
      return v;
    }
    throw KioError("expected float/double size byte in " + in.Name());
  }
  std::string w = ReadTextWord(in);
  return strtod(w.c_str(), nullptr);
}

bool ReadBool(Input& in, bool binary) {
  int c;
  if (!binary)
    while ((c = in.Peek()) >= 0 && isspace(c)) in.Get();
  c = in.Get();
  if (c != 'T' && c != 'F') throw KioError("expected T or F in " + in.Name());
  return c == 'T';
}

void SkipScalar(Input& in, bool binary) {
  if (!binary) {
    ReadTextWord(in);
    return;
  }
  int c = in.Peek();
  if (c == 'T' || c == 'F') {
    in.Get();
  } else if (c == 4 || c == 8) {
    char buf[8];
    in.Get();
    in.Read(buf, (size_t)c);
  } else {
    throw KioError("cannot skip unknown field payload in " + in.Name());
  }
}

static void ReadTextNumbers(Input& in, std::vector<float>* vals, std::vector<int>* row_ends) {
  // grammar: ws* '[' (number | newline)* ']' ; a newline (or ']') after >=1 numbers closes a row
  int c;
  while ((c = in.Peek()) >= 0 && isspace(c)) in.Get();
  if (in.Get() != '[') throw KioError("expected '[' in text matrix/vector in " + in.Name());
  std::string tok;
  size_t row_start = 0;
  for (;;) {
    c = in.Get();
    if (c < 0) throw KioError("end of input inside text matrix in " + in.Name());
    if (isspace(c) || c == ']') {
      if (!tok.empty()) {
        const char* s = tok.c_str();
        char* end = nullptr;
        float v = strtof(s, &end);
        if (end == s) throw KioError("bad number '" + tok + "' in text matrix");
        vals->push_back(v);
        tok.clear();
      }
      if (c == '\n' || c == ']') {
        if (vals->size() > row_start) {
          row_ends->push_back((int)vals->size());
          row_start = vals->size();
        }
      }
      if (c == ']') break;
    } else {
      tok.push_back((char)c);
    }
  }
  // rest of the line
  while ((c = in.Peek()) >= 0 && c != '\n' && isspace(c)) in.Get();
  if (in.Peek() == '\n') in.Get();
}

// Reads n elements growing the vector block by block: a corrupted dimension in a header then fails at the end of the
// input after at most one block of extra allocation, instead of zero-filling gigabytes first.
template <typename T>
static void ReadGrow(Input& in, std::vector<T>* v, size_t n) {
  constexpr size_t kBlock = (size_t)1 << 22;   // elements
  if (v->size() >= n) {   // a recycled buffer that is large enough: nothing to allocate, nothing to zero-fill
    v->resize(n);
    in.Read(v->data(), n * sizeof(T));
    return;
  }
  v->clear();
  size_t done = 0;
  while (done < n) {
    const size_t take = n - done < kBlock ? n - done : kBlock;
    v->resize(done + take);
    in.Read(v->data() + done, take * sizeof(T));
    done += take;
  }
}

void ReadVector(Input& in, bool binary, std::vector<float>* v) {
  v->clear();
  if (binary) {
    std::string tok;
    ReadToken(in, true, &tok);
    int32_t n = ReadInt32(in, true);
    if (n < 0) throw KioError("negative vector dimension");
    if (tok == "FV") {
      ReadGrow(in, v, (size_t)n);
    } else if (tok == "DV") {
      std::vector<double> d;
      ReadGrow(in, &d, (size_t)n);
      v->resize((size_t)n);
      for (int i = 0; i < n; ++i) (*v)[i] = (float)d[i];
    } else {
      throw KioError("expected FV or DV, got " + tok);
    }
    return;
  }
  std::vector<int> ends;
  ReadTextNumbers(in, v, &ends);
}

static inline float U16ToFloat(float mn, float range, uint16_t v) { return mn + range * 1.52590218966964e-05F * v; }

// "CM": per-column percentiles + one byte per element, column-major
static void ExpandCm(float min_value, float range, int rows, int cols, const uint16_t* hdr, const uint8_t* bytes, Matrix* m) {
  const size_t total = (size_t)rows * cols;
  m->rows = rows;
  m->cols = cols;
  m->ext = nullptr;
  m->cm = nullptr;
  m->cm_bytes = 0;
  // One table of the 256 values a byte can stand for per COLUMN (same expressions as Kaldi's CharToFloat, evaluated once per
  // value instead of once per element: the same floats), then the transposition column-major bytes -> row-major floats in
  // blocks of rows that stay in the L1.  The raw feats.scp of the recipes is compressed (make_mfcc.sh --compress true), and
  // with the device front-end it is what the reader threads of a table job see: 4.5 ns per element before, the decompression
  // was 41 us of host time per 400-frame utterance - six times the whole remaining host budget (profiles/r06_host_cost.md).
  m->data.resize(total);
  std::vector<float> lut((size_t)cols * 256);
  for (int c = 0; c < cols; ++c) {
    uint16_t q[4];
    memcpy(q, (const uint8_t*)hdr + (size_t)c * 8, 8);   // (a view's header has the archive's alignment)
    const float p0 = U16ToFloat(min_value, range, q[0]);
    const float p25 = U16ToFloat(min_value, range, q[1]);
    const float p75 = U16ToFloat(min_value, range, q[2]);
    const float p100 = U16ToFloat(min_value, range, q[3]);
    float* t = lut.data() + (size_t)c * 256;
    for (int v = 0; v < 256; ++v) {
      float x;
      if (v <= 64) x = p0 + (p25 - p0) * v * (1 / 64.0f);
      else if (v <= 192) x = p25 + (p75 - p25) * (v - 64) * (1 / 128.0f);
      else x = p75 + (p100 - p75) * (v - 192) * (1 / 63.0f);
      t[v] = x;
    }
  }
  constexpr int kRowBlock = 64;
  float* out = m->data.data();
  for (int r0 = 0; r0 < rows; r0 += kRowBlock) {
    const int r1 = std::min(rows, r0 + kRowBlock);
    for (int c = 0; c < cols; ++c) {
      const uint8_t* col = bytes + (size_t)c * rows;  // column-major payload
      const float* t = lut.data() + (size_t)c * 256;
      for (int r = r0; r < r1; ++r) out[(size_t)r * cols + c] = t[col[r]];
    }
  }
}

void ExpandCompressedView(const Matrix& view, Matrix* out) {
  if (!view.cm || view.cm_bytes < 16) throw KioError("ExpandCompressedView: not a compressed view");
  float mr[2];
  int32_t rc[2];
  memcpy(mr, view.cm, 8);
  memcpy(rc, view.cm + 8, 8);
  if (rc[0] != view.rows || rc[1] != view.cols || view.cm_bytes != 16 + (size_t)rc[1] * 8 + (size_t)rc[0] * rc[1])
    throw KioError("ExpandCompressedView: the view does not match its header");
  ExpandCm(mr[0], mr[1], rc[0], rc[1], (const uint16_t*)(view.cm + 16), view.cm + 16 + (size_t)rc[1] * 8, out);
}

static void ReadCompressed(Input& in, const std::string& tok, Matrix* m) {
  struct {
    float min_value, range;
    int32_t rows, cols;
  } h;
  in.Read(&h, 16);
  if (h.rows < 0 || h.cols < 0) throw KioError("bad compressed-matrix header");
  m->rows = h.rows;
  m->cols = h.cols;
  const size_t total = (size_t)h.rows * h.cols;
  if (tok == "CM") {
    std::vector<uint16_t> hdr;
    ReadGrow(in, &hdr, (size_t)h.cols * 4);
    std::vector<uint8_t> bytes;
    ReadGrow(in, &bytes, total);
    ExpandCm(h.min_value, h.range, h.rows, h.cols, hdr.data(), bytes.data(), m);
  } else if (tok == "CM2") {
    std::vector<uint16_t> d;
    ReadGrow(in, &d, total);
    m->data.assign(total, 0.f);
    const float inc = h.range * (1.0f / 65535.0f);
    for (size_t i = 0; i < d.size(); ++i) m->data[i] = d[i] * inc + h.min_value;
  } else {  // CM3
    std::vector<uint8_t> d;
    ReadGrow(in, &d, total);
    m->data.assign(total, 0.f);
    const float inc = h.range * (1.0f / 255.0f);
    for (size_t i = 0; i < d.size(); ++i) m->data[i] = d[i] * inc + h.min_value;
  }
}

void SkipBinaryMatrix(Input& in, int* rows, int* cols) {
  std::string tok;
  ReadToken(in, true, &tok);
  if (tok == "CM" || tok == "CM2" || tok == "CM3") {
    struct {
      float min_value, range;
      int32_t rows, cols;
    } h;
    in.Read(&h, 16);
    if (h.rows < 0 || h.cols < 0) throw KioError("bad compressed-matrix header");
    *rows = h.rows;
    *cols = h.cols;
    const long total = (long)h.rows * h.cols;
    in.Skip(tok == "CM" ? (long)h.cols * 8 + total : tok == "CM2" ? total * 2 : total);
    return;
  }
  if (tok != "FM" && tok != "DM") throw KioError("expected FM/DM/CM, got " + tok + " in " + in.Name());
  const int32_t r = ReadInt32(in, true), c = ReadInt32(in, true);
  if (r < 0 || c < 0) throw KioError("negative matrix dimension");
  *rows = r;
  *cols = c;
  in.Skip((long)r * c * (tok == "FM" ? 4 : 8));
}

void ReadMatrix(Input& in, bool binary, Matrix* m) {
  m->ext = nullptr;   // (a matrix object that was a view before owns its floats from here on)
  m->cm = nullptr;
  m->cm_bytes = 0;
  if (binary) {
    std::string tok;
    ReadToken(in, true, &tok);
    if (tok == "CM" || tok == "CM2" || tok == "CM3") {
      ReadCompressed(in, tok, m);
      return;
    }
    if (tok != "FM" && tok != "DM") throw KioError("expected FM/DM/CM, got " + tok + " in " + in.Name());
    int32_t r = ReadInt32(in, true), c = ReadInt32(in, true);
    if (r < 0 || c < 0) throw KioError("negative matrix dimension");
    m->rows = r;
    m->cols = c;
    if (tok == "FM") {
      ReadGrow(in, &m->data, (size_t)r * c);
    } else {
      std::vector<double> d;
      ReadGrow(in, &d, (size_t)r * c);
      m->data.resize(d.size());
      for (size_t i = 0; i < d.size(); ++i) m->data[i] = (float)d[i];
    }
    return;
  }
  std::vector<int> ends;
  m->data.clear();
  ReadTextNumbers(in, &m->data, &ends);
  m->rows = (int)ends.size();
  m->cols = m->rows ? ends[0] : 0;
  for (int r = 0; r < m->rows; ++r)
    if (ends[r] != (r + 1) * m->cols) throw KioError("ragged text matrix in " + in.Name());
}

void WriteToken(Output& out, bool binary, const char* tok) {
  (void)binary;
  out.Puts(tok);
  out.Put(' ');
}

void WriteInt32(Output& out, bool binary, int32_t v) {
  if (binary) {
    out.Put((char)4);
    out.Write(&v, 4);
  } else {
    char buf[32];
    snprintf(buf, sizeof buf, "%d ", v);
    out.Puts(buf);
  }
}

void WriteFloat(Output& out, bool binary, float v) {
  if (binary) {
    out.Put((char)4);
    out.Write(&v, 4);
  } else {
    char buf[48];
    snprintf(buf, sizeof buf, "%.9g ", (double)v);
    out.Puts(buf);
  }
}

void WriteDouble(Output& out, bool binary, double v) {
  if (binary) {
    out.Put((char)8);
    out.Write(&v, 8);
  } else {
    char buf[48];
    snprintf(buf, sizeof buf, "%.17g ", v);
    out.Puts(buf);
  }
}

void WriteBool(Output& out, bool binary, bool v) {
  out.Put(v ? 'T' : 'F');
  if (!binary) out.Put(' ');
}

static void PutFloatText(Output& out, float v) {
  char buf[48];
  if (isnan(v)) snprintf(buf, sizeof buf, "nan");
  else if (isinf(v)) snprintf(buf, sizeof buf, v > 0 ? "inf" : "-inf");
  else snprintf(buf, sizeof buf, "%.9g", (double)v);
  out.Puts(buf);
}

void WriteVector(Output& out, bool binary, const float* v, int n) {
  if (binary) {
    out.Puts("FV ");
    WriteInt32(out, true, n);
    out.Write(v, (size_t)n * 4);
  } else {
    out.Puts(" [ ");
    for (int i = 0; i < n; ++i) {
      PutFloatText(out, v[i]);
      out.Put(' ');
    }
    out.Puts("]\n");
  }
}

void WriteMatrix(Output& out, bool binary, const Matrix& m) {
  if (binary) {
    out.Puts("FM ");
    WriteInt32(out, true, m.rows);
    WriteInt32(out, true, m.cols);
    out.Write(m.data.data(), m.data.size() * 4);
  } else {
    if (m.rows == 0) {
      out.Puts(" [ ]\n");
      return;
    }
    out.Puts(" [");
    for (int r = 0; r < m.rows; ++r) {
      out.Puts("\n  ");
      for (int c = 0; c < m.cols; ++c) {
        PutFloatText(out, m.data[(size_t)r * m.cols + c]);
        out.Put(' ');
      }
    }
    out.Puts("]\n");
  }
}

// ------------------------------------------------------------------------------------- specifiers
static void SplitFirstColon(const std::string& spec, std::string* opts, std::string* rest) {
  size_t c = spec.find(':');
  if (c == std::string::npos) throw KioError("invalid table specifier (no ':'): " + spec);
  *opts = spec.substr(0, c);
  *rest = spec.substr(c + 1);
}

static std::vector<std::string> SplitComma(const std::string& s) {
  std::vector<std::string> out;
  size_t a = 0;
  for (;;) {
    size_t b = s.find(',', a);
    out.push_back(s.substr(a, b == std::string::npos ? std::string::npos : b - a));
    if (b == std::string::npos) break;
    a = b + 1;
  }
  return out;
}

RspecifierOptions ParseRspecifier(const std::string& rspecifier) {
  RspecifierOptions o;
  std::string opts, rest;
  SplitFirstColon(Trim(rspecifier), &opts, &rest);
  bool kind = false;
  for (const std::string& t : SplitComma(opts)) {
    if (t == "ark") kind = true;
    else if (t == "scp") { kind = true; o.is_scp = true; }
    else if (t == "s") o.sorted = true;
    else if (t == "ns") o.sorted = false;
    else if (t == "cs") o.called_sorted = true;
    else if (t == "ncs") o.called_sorted = false;
    else if (t == "p") o.permissive = true;
    else if (t == "np") o.permissive = false;
    else if (t == "o") o.once = true;
    else if (t == "no") o.once = false;
    else if (t == "bg") o.background = true;
    else if (t == "t" || t == "b") {}  // accepted, meaningless for reading
    else throw KioError("invalid option '" + t + "' in rspecifier " + rspecifier);
  }
  if (!kind) throw KioError("rspecifier must contain ark or scp: " + rspecifier);
  o.rxfilename = Trim(rest);
  return o;
}

WspecifierOptions ParseWspecifier(const std::string& wspecifier) {
  WspecifierOptions o;
  std::string opts, rest;
  SplitFirstColon(Trim(wspecifier), &opts, &rest);
  std::vector<std::string> order;
  for (const std::string& t : SplitComma(opts)) {
    if (t == "ark") { o.has_ark = true; order.push_back(t); }
    else if (t == "scp") { o.has_scp = true; order.push_back(t); }
    else if (t == "t") o.binary = false;
    else if (t == "b") o.binary = true;
    else if (t == "f") o.flush = true;
    else if (t == "nf") o.flush = false;
    else if (t == "p") o.permissive = true;
    else throw KioError("invalid option '" + t + "' in wspecifier " + wspecifier);
  }
  if (order.empty()) throw KioError("wspecifier must contain ark and/or scp: " + wspecifier);
  if (order.size() == 1) {
    if (o.has_scp) throw KioError("scp-only wspecifiers are not supported (Kaldi needs ark,scp): " + wspecifier);
    o.ark_wxfilename = Trim(rest);
  } else {
    size_t c = rest.find(',');
    if (c == std::string::npos) throw KioError("ark,scp wspecifier needs two file names: " + wspecifier);
    std::string a = Trim(rest.substr(0, c)), b = Trim(rest.substr(c + 1));
    if (order[0] == "ark") { o.ark_wxfilename = a; o.scp_wxfilename = b; }
    else { o.scp_wxfilename = a; o.ark_wxfilename = b; }
  }
  return o;
}

// ------------------------------------------------------------------------------------- readers
SequentialMatrixReader::SequentialMatrixReader(const std::string& rspecifier) {
  opts_ = ParseRspecifier(rspecifier);
  in_.Open(opts_.rxfilename, /*drain_pipe=*/!opts_.is_scp);   // an archive through a pipe: the feature pipe of extract_xvectors_new.sh:79
}

SequentialMatrixReader::~SequentialMatrixReader() {}

static bool ReadKey(Input& in, std::string* key) {
  key->clear();
  int c;
  while ((c = in.Peek()) >= 0 && isspace(c)) in.Get();
  if (c < 0) return false;
  while ((c = in.Peek()) >= 0 && !isspace(c)) key->push_back((char)in.Get());
  if (c >= 0 && c != '\n') in.Get();  // the single separator; keep '\n' for scp line logic
  return true;
}

bool SequentialMatrixReader::Next(std::string* key, Matrix* m, std::string* error) {
  error->clear();
  if (!opts_.is_scp) {
    if (!ReadKey(in_, key)) return false;
    bool binary = ReadBinaryHeader(in_);
    ReadMatrix(in_, binary, m);  // a corrupt archive is fatal, as in Kaldi
    return true;
  }
  // scp: "key rxfilename\n"
  if (!ReadKey(in_, key)) return false;
  std::string rx;
  int c;
  while ((c = in_.Get()) >= 0 && c != '\n') rx.push_back((char)c);
  rx = Trim(rx);
  if (rx.empty()) {
    *error = "empty rxfilename for key " + *key;
    return true;
  }
  try {
    // fast path: consecutive entries of the same archive
    std::string path = rx;
    long offset = -1;
    size_t colon = rx.rfind(':');
    if (rx.back() != '|' && colon != std::string::npos && colon + 1 < rx.size() &&
        std::all_of(rx.begin() + colon + 1, rx.end(), [](char ch) { return isdigit((unsigned char)ch); })) {
      path = rx.substr(0, colon);
      offset = strtol(rx.c_str() + colon + 1, nullptr, 10);
    }
    if (offset >= 0 && path == data_path_ && data_in_.IsOpen()) {
      data_in_.Seek(offset);
    } else {
      data_in_.Open(rx);
      data_path_ = offset >= 0 ? path : std::string();
    }
    bool binary = ReadBinaryHeader(data_in_);
    ReadMatrix(data_in_, binary, m);
    if (data_path_.empty()) data_in_.Close();
  } catch (const KioError& e) {
    *error = e.what();
  }
  return true;
}

int SequentialMatrixReader::Close() { return in_.Close(); }

// "path:123" -> (path, 123); anything else -> offset -1
static long SplitOffset(const std::string& rx, std::string* path) {
  *path = rx;
  size_t colon = rx.rfind(':');
  if (!rx.empty() && rx.back() != '|' && colon != std::string::npos && colon + 1 < rx.size() &&
      std::all_of(rx.begin() + colon + 1, rx.end(), [](char ch) { return isdigit((unsigned char)ch); })) {
    *path = rx.substr(0, colon);
    return strtol(rx.c_str() + colon + 1, nullptr, 10);
  }
  return -1;
}

MatrixTableIndexer::MatrixTableIndexer(const std::string& rspecifier) {
  opts_ = ParseRspecifier(rspecifier);
  const std::string rx = Trim(opts_.rxfilename);
  if (rx.empty() || rx == "-" || rx.back() == '|') return;   // a stream: not addressable
  in_.Open(opts_.rxfilename);
  if (!in_.IsRegularFile()) return;
  if (opts_.is_scp) {
    usable_ = true;
    return;
  }
  // archive: binary objects only (the first one decides; a text object later on is an error of the index pass)
  const long start = in_.FileTell();
  int c;
  while ((c = in_.Peek()) >= 0 && isspace(c)) in_.Get();
  while ((c = in_.Peek()) >= 0 && !isspace(c)) in_.Get();
  if (c >= 0) in_.Get();
  usable_ = in_.Peek() == 0 || in_.Peek() < 0;   // "\0B" follows the first key (or the archive is empty)
  in_.Seek(start);
}

bool MatrixTableIndexer::Next(Entry* e) {
  *e = Entry();
  if (!ReadKey(in_, &e->key)) return false;
  if (!opts_.is_scp) {
    e->rx = in_.Name();
    {
      std::string p;
      if (SplitOffset(e->rx, &p) >= 0) e->rx = p;   // "ark:file:offset" was opened at that offset
    }
    e->offset = in_.FileTell();
    if (!ReadBinaryHeader(in_)) throw KioError("text object in a binary archive (key " + e->key + ") in " + in_.Name());
    SkipBinaryMatrix(in_, &e->rows, &e->cols);
    return true;
  }
  std::string rx;
  int c;
  while ((c = in_.Get()) >= 0 && c != '\n') rx.push_back((char)c);
  rx = Trim(rx);
  if (rx.empty()) {
    e->error = "empty rxfilename for key " + e->key;
    return true;
  }
  std::string path;
  const long offset = SplitOffset(rx, &path);
  e->rx = rx;
  if (offset < 0) return true;   // pipe or whole file: the reading thread opens it
  try {
    if (!(path == data_path_ && data_in_.IsOpen())) {
      data_in_.Open(path);
      data_path_ = path;
    }
    data_in_.Seek(offset);
    e->rx = path;
    e->offset = offset;
    if (ReadBinaryHeader(data_in_)) SkipBinaryMatrix(data_in_, &e->rows, &e->cols);   // text objects keep rows = -1
  } catch (const KioError& ex) {
    e->error = ex.what();
    data_in_.Close();
    data_path_.clear();
  }
  return true;
}

void ReadIndexedMatrix(const MatrixTableIndexer::Entry& e, Input* in, std::string* in_path, Matrix* m) {
  if (e.offset < 0) {
    Input one;
    one.Open(e.rx);
    const bool binary = ReadBinaryHeader(one);
    ReadMatrix(one, binary, m);
    one.Close();
    return;
  }
  if (!(in->IsOpen() && *in_path == e.rx)) {
    in->Open(e.rx);
    *in_path = e.rx;
  }
  in->Seek(e.offset);
  const bool binary = ReadBinaryHeader(*in);
  ReadMatrix(*in, binary, m);
}

namespace {
char g_fault_prog[64] = "xvec-hip";
void MappedFileFault(int) {
  // async-signal-safe: write(2) and _exit only
  const char a[] = "ERROR (", b[] = ") an input file changed (was truncated) while it was being read through its mapping\n";
  (void)!write(2, a, sizeof a - 1);
  (void)!write(2, g_fault_prog, strlen(g_fault_prog));
  (void)!write(2, b, sizeof b - 1);
  _exit(255);
}
}  // namespace

void InstallMappedFileFaultHandler(const char* program) {
  if (program) {
    strncpy(g_fault_prog, program, sizeof g_fault_prog - 1);
    g_fault_prog[sizeof g_fault_prog - 1] = 0;
  }
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = MappedFileFault;
  sigemptyset(&sa.sa_mask);
  (void)sigaction(SIGBUS, &sa, nullptr);
}

FileMapper::Mapped FileMapper::Map(const std::string& path) {
  std::lock_guard<std::mutex> lock(mu_);
  auto it = maps_.find(path);
  if (it != maps_.end()) return it->second;
  Mapped f;
  const int fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
  if (fd >= 0) {
    struct stat st;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
      void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
      if (p != MAP_FAILED) {
        f.base = (const uint8_t*)p;
        f.size = (size_t)st.st_size;
      }
    }
    close(fd);
  }
  maps_.emplace(path, f);
  return f;
}

bool FileMapper::View(const MatrixTableIndexer::Entry& e, Matrix* m, bool allow_compressed) {
  if (e.offset < 0 || e.rows < 0 || e.cols < 0) return false;
  const Mapped f = Map(e.rx);
  if (!f.base || (size_t)e.offset + 5 > f.size) return false;
  const uint8_t* p = f.base + e.offset;
  if (p[0] != 0 || p[1] != 'B') return false;
  if (p[2] == 'F' && p[3] == 'M' && p[4] == ' ') {
    // "\0B" "FM " '\4' <int32 rows> '\4' <int32 cols> <rows * cols floats>
    const size_t head = 2 + 3 + 5 + 5;
    if ((size_t)e.offset + head > f.size || p[5] != 4 || p[10] != 4) return false;
    int32_t r, c;
    memcpy(&r, p + 6, 4);
    memcpy(&c, p + 11, 4);
    if (r != e.rows || c != e.cols || r < 0 || c < 0) return false;
    const size_t bytes = (size_t)r * c * 4;
    if ((size_t)e.offset + head + bytes > f.size) return false;
    m->rows = r;
    m->cols = c;
    m->data.clear();
    m->cm = nullptr;
    m->cm_bytes = 0;
    m->ext = (const float*)(p + head);
    return true;
  }
  if (allow_compressed && p[2] == 'C' && p[3] == 'M' && p[4] == ' ') {
    // "\0B" "CM " {float min_value, range; int32 rows, cols} <cols x 4 uint16> <cols x rows bytes>
    const size_t head = 2 + 3;
    if ((size_t)e.offset + head + 16 > f.size) return false;
    int32_t rc[2];
    memcpy(rc, p + head + 8, 8);
    if (rc[0] != e.rows || rc[1] != e.cols || rc[0] < 0 || rc[1] < 0) return false;
    const size_t bytes = 16 + (size_t)rc[1] * 8 + (size_t)rc[0] * rc[1];
    if ((size_t)e.offset + head + bytes > f.size) return false;
    m->rows = rc[0];
    m->cols = rc[1];
    m->data.clear();
    m->ext = nullptr;
    m->cm = p + head;
    m->cm_bytes = bytes;
    return true;
  }
  return false;
}

FileMapper::~FileMapper() {
  for (auto& kv : maps_)
    if (kv.second.base) (void)munmap((void*)kv.second.base, kv.second.size);
}

RandomAccessVectorReader::RandomAccessVectorReader(const std::string& rspecifier) {
  RspecifierOptions o = ParseRspecifier(rspecifier);
  Input in;
  in.Open(o.rxfilename);
  std::string key;
  if (o.is_scp) {
    while (ReadKey(in, &key)) {
      std::string rx;
      int c;
      while ((c = in.Get()) >= 0 && c != '\n') rx.push_back((char)c);
      Entry e;
      e.key = key;
      e.rx = Trim(rx);
      entries_.push_back(std::move(e));
    }
  } else {
    while (ReadKey(in, &key)) {
      Entry e;
      e.key = key;
      bool binary = ReadBinaryHeader(in);
      ReadVector(in, binary, &e.v);
      e.loaded = true;
      entries_.push_back(std::move(e));
    }
  }
}

int RandomAccessVectorReader::Find(const std::string& key) {
  if (index_.size() != entries_.size()) {
    index_.clear();
    for (size_t i = 0; i < entries_.size(); ++i) index_.emplace(entries_[i].key, (int)i);   // first entry of a key wins
  }
  auto it = index_.find(key);
  return it == index_.end() ? -1 : it->second;
}

SequentialVectorReader::SequentialVectorReader(const std::string& rspecifier) {
  opts_ = ParseRspecifier(rspecifier);
  in_.Open(opts_.rxfilename);
}

bool SequentialVectorReader::Next(std::string* key, std::vector<float>* v, std::string* error) {
  error->clear();
  if (!ReadKey(in_, key)) return false;
  if (!opts_.is_scp) {
    bool binary = ReadBinaryHeader(in_);
    ReadVector(in_, binary, v);
    return true;
  }
  std::string rx;
  int c;
  while ((c = in_.Get()) >= 0 && c != '\n') rx.push_back((char)c);
  rx = Trim(rx);
  try {
    if (rx.empty()) throw KioError("empty rxfilename for key " + *key);
    Input data;
    data.Open(rx);
    bool binary = ReadBinaryHeader(data);
    ReadVector(data, binary, v);
  } catch (const KioError& e) {
    *error = e.what();
  }
  return true;
}

int SequentialVectorReader::Close() { return in_.Close(); }

std::vector<TokenList> ReadTokenVectorTable(const std::string& rspecifier) {
  RspecifierOptions o = ParseRspecifier(rspecifier);
  if (o.is_scp) throw KioError("token-vector tables are read from archives (ark:...), not scp: " + rspecifier);
  Input in;
  in.Open(o.rxfilename);
  std::vector<TokenList> out;
  std::string line;
  auto flush = [&]() {
    std::string t = Trim(line);
    line.clear();
    if (t.empty()) return;
    TokenList e;
    size_t i = 0;
    while (i < t.size()) {
      while (i < t.size() && isspace((unsigned char)t[i])) ++i;
      size_t j = i;
      while (j < t.size() && !isspace((unsigned char)t[j])) ++j;
      if (j > i) {
        if (e.key.empty()) e.key = t.substr(i, j - i);
        else e.tokens.push_back(t.substr(i, j - i));
      }
      i = j;
    }
    out.push_back(std::move(e));
  };
  int c;
  while ((c = in.Get()) >= 0) {
    if (c == '\n') flush();
    else line.push_back((char)c);
  }
  flush();
  in.Close();
  return out;
}

void ReadVectorObject(const std::string& rxfilename, std::vector<float>* v) {
  Input in;
  in.Open(rxfilename);
  bool binary = ReadBinaryHeader(in);
  ReadVector(in, binary, v);
  in.Close();
}

void ReadMatrixObject(const std::string& rxfilename, Matrix* m) {
  Input in;
  in.Open(rxfilename);
  bool binary = ReadBinaryHeader(in);
  ReadMatrix(in, binary, m);
  in.Close();
}

void WriteVectorObject(const std::string& wxfilename, bool binary, const float* v, int n) {
  Output out;
  out.Open(wxfilename);
  if (binary) out.Write("\0B", 2);
  WriteVector(out, binary, v, n);
  out.Close();
}

bool RandomAccessVectorReader::HasKey(const std::string& key) { return Find(key) >= 0; }

const std::vector<float>& RandomAccessVectorReader::Value(const std::string& key) {
  int i = Find(key);
  if (i < 0) throw KioError("key not found in table: " + key);
  Entry& e = entries_[i];
  if (!e.loaded) {
    // "file:offset" entries (what copy-vector / compute-vad-decision write): the archive is opened - mapped - once and every
    // lookup is a seek.  Opening it per key was 8 us per utterance in the consumer thread of a table job with a VAD table, the
    // wall of the recipes' own pipeline once it ran on the device (100 k utt/s; profiles/r06_host_cost.md).
    std::string path;
    const long off = SplitOffset(e.rx, &path);
    if (off >= 0) {
      if (!(data_in_.IsOpen() && data_path_ == path)) {
        data_in_.Open(path);
        data_path_ = path;
      }
      data_in_.Seek(off);
      const bool binary = ReadBinaryHeader(data_in_);
      ReadVector(data_in_, binary, &e.v);
    } else {
      Input in;
      in.Open(e.rx);
      const bool binary = ReadBinaryHeader(in);
      ReadVector(in, binary, &e.v);
    }
    e.loaded = true;
  }
  return e.v;
}

void RandomAccessVectorReader::Forget(const std::string& key) {
  const int i = Find(key);
  if (i < 0) return;
  Entry& e = entries_[i];
  if (e.rx.empty() || !e.loaded) return;   // an archive loaded as a whole has no way to read the value again
  std::vector<float>().swap(e.v);
  e.loaded = false;
}

// ------------------------------------------------------------------------------------- writer
TableWriter::TableWriter(const std::string& wspecifier) {
  opts_ = ParseWspecifier(wspecifier);
  ark_.Open(opts_.ark_wxfilename);
  if (opts_.has_scp) scp_.Open(opts_.scp_wxfilename);
}

TableWriter::~TableWriter() {
  try {
    Close();
  } catch (...) {
  }
}

void TableWriter::Begin(const std::string& key) {
  if (key.empty() || key.find_first_of(" \t\n") != std::string::npos)
    throw KioError("invalid table key '" + key + "'");
  ark_.Puts(key);
  ark_.Put(' ');
  if (opts_.has_scp) {
    char buf[64];
    snprintf(buf, sizeof buf, ":%lld\n", (long long)ark_.Tell());
    pending_scp_line_ = key + " " + opts_.ark_wxfilename + buf;
  }
  if (opts_.binary) ark_.Write("\0B", 2);
}

void TableWriter::End() {
  if (opts_.flush) ark_.Flush();
  if (opts_.has_scp) {
    // the ark bytes are flushed before the scp line becomes visible (extract_xvectors_new.sh:99 cats the scp)
    if (!opts_.flush) ark_.Flush();
    scp_.Puts(pending_scp_line_);
    if (opts_.flush) scp_.Flush();
  }
}

void TableWriter::WriteVec(const std::string& key, const float* v, int n) {
  Begin(key);
  WriteVector(ark_, opts_.binary, v, n);
  End();
}

void TableWriter::WriteMat(const std::string& key, const Matrix& m) {
  Begin(key);
  WriteMatrix(ark_, opts_.binary, m);
  End();
}

void TableWriter::WriteInt32(const std::string& key, int32_t v) {
  Begin(key);
  xv::WriteInt32(ark_, opts_.binary, v);
  if (!opts_.binary) ark_.Put('\n');
  End();
}

void TableWriter::Close() {
  if (ark_.IsOpen()) ark_.Close();
  if (scp_.IsOpen()) scp_.Close();
}

}  // namespace xv
