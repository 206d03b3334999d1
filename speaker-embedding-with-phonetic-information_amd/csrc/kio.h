// kio - Kaldi table / object I/O without Kaldi.
//
// Replaces, for the one hot path, what the reference gets from Kaldi's util/ and matrix/ libraries
// (not vendored under /root/reference): rxfilename / wxfilename handling, rspecifier / wspecifier
// parsing, `SequentialBaseFloatMatrixReader` (features, extract_xvectors_new.sh:79) and
// `BaseFloatVectorWriter` (ark,scp output, extract_xvectors_new.sh:93).  The text matrix form is the
// one the reference's own Python reads/writes (egs/sre/v2/steps/libs/common.py:354-470).
// Formats: SURVEY.md App. B.1 / B.2.
#pragma once
#include <stdint.h>
#include <stdio.h>

#include <memory>
#include <stdexcept>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace xv {

struct KioError : public std::runtime_error {
  explicit KioError(const std::string& m) : std::runtime_error(m) {}
};

// ---------------------------------------------------------------------------------------------
// Byte source: regular file (optionally at an offset), stdin, or the stdout of `cmd |`.
class PipeDrain;   // kio.cc: a thread that keeps reading an input pipe into a ring of blocks

class Input {
 public:
  Input();
  ~Input();
  Input(const Input&) = delete;
  Input& operator=(const Input&) = delete;
  // rxfilename forms: "file", "file:123", "-", "cmd args |"
  // drain_pipe: for "cmd |" inputs, a second thread does the read(2) calls and the parser consumes from a ring of 1 MiB
  // blocks (feature archives through a pipe, extract_xvectors_new.sh:79: the syscalls and the kernel's copy out of the pipe
  // buffer then overlap the parsing and the copy into the batch; XVEC_DEBUG=pipe_drain=0 turns it off)
  void Open(const std::string& rxfilename, bool drain_pipe = false);
  void OpenMemory(const void* data, size_t n);
  void Seek(long offset);           // regular files / memory only
  bool IsRegularFile() const;       // an fopen'ed regular file (seekable): not a pipe, FIFO, device or standard input
  long FileTell();                  // regular files: current offset
  void Skip(long n);                // regular files: move n bytes ahead without reading them
  bool IsOpen() const { return f_ != nullptr || mem_ != nullptr; }
  // Close; for pipes returns the child's exit status (0 otherwise).
  int Close();
  int Peek();                       // next byte or EOF (-1), not consumed
  int Get();                        // next byte or EOF
  void Read(void* dst, size_t n);   // throws on short read
  size_t ReadUpTo(void* dst, size_t n);   // up to n bytes, fewer only at the end of the input
  bool Eof() { return Peek() < 0; }
  // memory sources only: look-ahead and position (used by the model parser)
  int PeekAt(size_t k) const { return (mem_ && mem_pos_ + k < mem_n_) ? mem_[mem_pos_ + k] : -1; }
  size_t Tell() const { return mem_pos_; }
  const std::string& Name() const { return name_; }

 private:
  FILE* f_ = nullptr;
  bool is_pipe_ = false, is_stdin_ = false;
  const unsigned char* mem_ = nullptr;
  size_t mem_n_ = 0, mem_pos_ = 0;
  bool mapped_ = false;   // mem_ is this object's read-only mapping of a regular file (Open): unmapped by Close
  std::string name_;
  std::unique_ptr<PipeDrain> drain_;
};

// Byte sink: file, stdout ("-"), or the stdin of `| cmd`.
class Output {
 public:
  Output() = default;
  ~Output();
  Output(const Output&) = delete;
  Output& operator=(const Output&) = delete;
  void Open(const std::string& wxfilename);
  bool IsOpen() const { return f_ != nullptr; }
  int Close();
  void Write(const void* src, size_t n);
  void Put(char c) { Write(&c, 1); }
  void Puts(const std::string& s) { Write(s.data(), s.size()); }
  void Flush();
  int64_t Tell() const { return pos_; }  // bytes written so far (offset for scp lines)
  const std::string& Name() const { return name_; }

 private:
  FILE* f_ = nullptr;
  bool is_pipe_ = false, is_stdout_ = false;
  int64_t pos_ = 0;
  std::string name_;
};

// ---------------------------------------------------------------------------------------------
// Object-level helpers (binary = the "\0B" flavour).
bool ReadBinaryHeader(Input& in);  // consumes "\0B" if present, returns whether it was
void ReadToken(Input& in, bool binary, std::string* tok);
void ExpectToken(Input& in, bool binary, const char* tok);
int32_t ReadInt32(Input& in, bool binary);
// float or double (binary: size byte 4 or 8), returned as double
double ReadFloatOrDouble(Input& in, bool binary);
bool ReadBool(Input& in, bool binary);
// Skips one self-describing scalar (int32/float/double/bool) - used for unknown optional fields.
void SkipScalar(Input& in, bool binary);

struct Matrix {
  int rows = 0, cols = 0;
  std::vector<float> data;  // row-major, no padding
  // A VIEW instead of a copy (ViewIndexedMatrix): rows x cols floats that live in a mapped archive file.  The address has the
  // alignment the archive gave it (a key of any length precedes the object): memcpy from it, never dereference it as float.
  const float* ext = nullptr;
  // ... or of a COMPRESSED matrix ("CM": one byte per element, Kaldi's default for stored features): cm points behind the "CM "
  // token of the object - {float min_value, range; int32 rows, cols; uint16 percentiles[cols][4]; uint8 data[cols][rows]},
  // cm_bytes of them, any alignment.  rows / cols are set, data is empty, ext is null: such a matrix can only be handed to the
  // device front-end (which expands it on the GPU, kernels.h CmExpandArgs) or expanded with ExpandCompressedView.
  const uint8_t* cm = nullptr;
  size_t cm_bytes = 0;
  const float* Data() const { return ext ? ext : data.data(); }
  float* Row(int r) { return data.data() + (size_t)r * cols; }
  const float* Row(int r) const { return Data() + (size_t)r * cols; }
};

// Reads FM / DM / CM / CM2 / CM3 (binary) or " [ ... ]" (text).
void ReadMatrix(Input& in, bool binary, Matrix* m);
// Regular files are read through read-only mappings (Input, FileMapper).  A file that is truncated by someone else WHILE a tool
// reads it turns a read into SIGBUS where read(2) would have returned short; the command-line tools call this once so that
// the process then ends like every other input error - "ERROR ... changed while it was being read", exit status 255 - instead
// of a bare signal.  (Not installed by the library itself: a host process owns its signal handlers.)
void InstallMappedFileFaultHandler(const char* program);
// Expands a compressed view (Matrix::cm) into floats on the host: *out owns them (the same floats ReadMatrix delivers).
void ExpandCompressedView(const Matrix& view, Matrix* out);
// Regular files, positioned behind the "\0B" of a binary matrix (FM / DM / CM / CM2 / CM3): reads the header only, reports the
// dimensions and moves to the end of the object (the index pass of the parallel table readers).
void SkipBinaryMatrix(Input& in, int* rows, int* cols);
// Reads FV / DV (binary) or " [ ... ]" (text).
void ReadVector(Input& in, bool binary, std::vector<float>* v);
void WriteToken(Output& out, bool binary, const char* tok);
void WriteInt32(Output& out, bool binary, int32_t v);
void WriteFloat(Output& out, bool binary, float v);
void WriteDouble(Output& out, bool binary, double v);
void WriteBool(Output& out, bool binary, bool v);
void WriteVector(Output& out, bool binary, const float* v, int n);
void WriteMatrix(Output& out, bool binary, const Matrix& m);

// ---------------------------------------------------------------------------------------------
// Table specifiers.
struct RspecifierOptions {
  bool is_scp = false;
  bool sorted = false, called_sorted = false, permissive = false, once = false, background = false;
  std::string rxfilename;
};
RspecifierOptions ParseRspecifier(const std::string& rspecifier);

struct WspecifierOptions {
  bool has_ark = false, has_scp = false;
  bool binary = true, flush = true, permissive = false;
  std::string ark_wxfilename, scp_wxfilename;
};
WspecifierOptions ParseWspecifier(const std::string& wspecifier);

// Sequential reader of a table of float matrices ("ark:..." or "scp:...").
class SequentialMatrixReader {
 public:
  explicit SequentialMatrixReader(const std::string& rspecifier);
  ~SequentialMatrixReader();
  // Advances to the next entry; returns false at the end.  Per-entry problems in scp mode
  // (unreadable file) are reported through `error` (non-empty) with key set, and reading continues.
  bool Next(std::string* key, Matrix* m, std::string* error);
  // Exit status of an input pipe (valid after the table is exhausted); 0 if not a pipe.
  int Close();

 private:
  RspecifierOptions opts_;
  Input in_;         // the ark stream, or the scp file
  Input data_in_;    // scp mode: currently open data file
  std::string data_path_;
};

// Index pass over a matrix table whose objects can be addressed individually: binary archives in a regular file ("ark:file")
// and script files whose entries are "path:offset" (or plain files).  Next() yields one entry without reading its data - key,
// where the object starts, and (binary objects) its dimensions - so that several threads can read a table's matrices in
// parallel and batches can be formed before the data is touched (table_extract.cc).  usable() is false for what can only
// be read front to back (pipes, standard input, text archives): use SequentialMatrixReader there.
class MatrixTableIndexer {
 public:
  struct Entry {
    std::string key;
    std::string rx;      // what to open: "path" (+ offset below) or a whole rxfilename (pipe, plain file) when offset < 0
    long offset = -1;    // byte offset of the object (its "\0B") inside rx, or -1
    int rows = -1, cols = -1;   // -1: not known without reading (text object, pipe)
    std::string error;   // the entry could not be located (scp mode): skip it with a warning
  };
  explicit MatrixTableIndexer(const std::string& rspecifier);
  bool usable() const { return usable_; }
  bool Next(Entry* e);

 private:
  RspecifierOptions opts_;
  Input in_;          // the archive, or the script file
  Input data_in_;     // scp mode: the data file of the previous entry (consecutive entries usually share it)
  std::string data_path_;
  bool usable_ = false;
};
// Reads the matrix an index entry points to.  `in` / `in_path` cache the open data file between calls of one thread.
void ReadIndexedMatrix(const MatrixTableIndexer::Entry& e, Input* in, std::string* in_path, Matrix* m);
// The same without touching the data: for a binary FLOAT matrix ("FM") at a known offset of a regular file, *m becomes a view of
// the file's pages (the file is mapped read-only once per FileMapper and path, and stays mapped while the FileMapper lives - a
// job owns one, so every view dies before the mapping does and a file rewritten between two jobs of one process is mapped
// anew).  The one host copy of such an utterance is then the one into the device's pinned staging buffer (table_extract.cc) -
// read(2) into a buffer of its own first was half of the host time of a table job.  false: anything else
// (compressed or double matrices, text, pipes, a header that does not check out): the caller reads it with ReadIndexedMatrix,
// which also words the error.  Thread-safe.
class FileMapper {
 public:
  FileMapper() = default;
  ~FileMapper();
  FileMapper(const FileMapper&) = delete;
  FileMapper& operator=(const FileMapper&) = delete;
  // allow_compressed: a "CM" object becomes a compressed view (Matrix::cm) instead of `false`
  bool View(const MatrixTableIndexer::Entry& e, Matrix* m, bool allow_compressed = false);

 private:
  struct Mapped {
    const uint8_t* base = nullptr;
    size_t size = 0;
  };
  Mapped Map(const std::string& path);
  std::mutex mu_;
  std::map<std::string, Mapped> maps_;   // by path; a file that could not be mapped is remembered with base == nullptr
};

// Sequential reader of a table of float vectors ("ark:..." or "scp:...").  A corrupt archive is fatal (KioError),
// an unreadable scp entry is reported through `error` and reading continues.
class SequentialVectorReader {
 public:
  explicit SequentialVectorReader(const std::string& rspecifier);
  bool Next(std::string* key, std::vector<float>* v, std::string* error);
  int Close();

 private:
  RspecifierOptions opts_;
  Input in_;
};

// Random-access reader over an scp (used by the front-end for vad.scp: "scp,s,cs:...").
// Also accepts "ark:" by loading the whole archive.
class RandomAccessVectorReader {
 public:
  explicit RandomAccessVectorReader(const std::string& rspecifier);
  bool HasKey(const std::string& key);
  const std::vector<float>& Value(const std::string& key);
  // scp tables: gives the memory of a loaded value back (it is read again if the key is asked for once more).  A table job
  // forgets the VAD decisions of a batch once the batch is on the device: kept, a hundred thousand four-minute recordings are
  // 10 GB of them by the end of the job.  (What Kaldi's "scp,s,cs:" options promise its reader it may do.)
  void Forget(const std::string& key);

 private:
  struct Entry { std::string key, rx; std::vector<float> v; bool loaded = false; };
  std::vector<Entry> entries_;
  std::unordered_map<std::string, int> index_;
  int Find(const std::string& key);
  // scp tables: the data file of the previous lookup stays open (mapped); consecutive keys of a job point into the same archive
  Input data_in_;
  std::string data_path_;
};

// A text table of token lists, "key tok1 tok2 ...\n" per entry (spk2utt, "ark:$data/spk2utt").
struct TokenList {
  std::string key;
  std::vector<std::string> tokens;
};
std::vector<TokenList> ReadTokenVectorTable(const std::string& rspecifier);

// Whole-file Kaldi objects (with the optional binary header): mean.vec, transform.mat.
void ReadVectorObject(const std::string& rxfilename, std::vector<float>* v);
void ReadMatrixObject(const std::string& rxfilename, Matrix* m);
void WriteVectorObject(const std::string& wxfilename, bool binary, const float* v, int n);

// Writer of a table of float vectors / matrices: "ark:", "ark,t:", "ark,scp:a,b", "scp,ark:b,a".
class TableWriter {
 public:
  explicit TableWriter(const std::string& wspecifier);
  ~TableWriter();
  void WriteVec(const std::string& key, const float* v, int n);
  void WriteMat(const std::string& key, const Matrix& m);
  void WriteInt32(const std::string& key, int32_t v);   // Int32Writer (num_utts.ark)
  void Close();

 private:
  void Begin(const std::string& key);
  void End();
  WspecifierOptions opts_;
  Output ark_, scp_;
  std::string pending_scp_line_;
};

}  // namespace xv
