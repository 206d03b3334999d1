// nnet3 `.raw` reader / writer.  See nnet3_raw.h.
#include "nnet3_raw.h"

#include <ctype.h>
#include <stdlib.h>
#include <string.h>

namespace xv {

static std::string TrimS(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && isspace((unsigned char)s[a])) ++a;
  while (b > a && isspace((unsigned char)s[b - 1])) --b;
  return s.substr(a, b - a);
}

// ------------------------------------------------------------------------------- descriptors
namespace {
struct DescParser {
  const std::string& s;
  size_t p = 0;
  explicit DescParser(const std::string& str) : s(str) {}
  void Ws() {
    while (p < s.size() && isspace((unsigned char)s[p])) ++p;
  }
  std::string Ident() {
    Ws();
    size_t a = p;
    while (p < s.size() && (isalnum((unsigned char)s[p]) || s[p] == '_' || s[p] == '.' || s[p] == '-')) ++p;
    if (a == p) throw KioError("descriptor parse error in '" + s + "'");
    return s.substr(a, p - a);
  }
  void Expect(char c) {
    Ws();
    if (p >= s.size() || s[p] != c) throw KioError(std::string("descriptor: expected '") + c + "' in '" + s + "'");
    ++p;
  }
  bool Peek(char c) {
    Ws();
    return p < s.size() && s[p] == c;
  }
  int Int() {
    Ws();
    char* end = nullptr;
    long v = strtol(s.c_str() + p, &end, 10);
    if (end == s.c_str() + p) throw KioError("descriptor: expected integer in '" + s + "'");
    p = (size_t)(end - s.c_str());
    return (int)v;
  }
  std::vector<DescTerm> Expr() {
    std::string id = Ident();
    if (!Peek('(')) return {DescTerm{id, 0}};
    Expect('(');
    std::vector<DescTerm> out;
    if (id == "Append") {
      for (;;) {
        std::vector<DescTerm> a = Expr();
        out.insert(out.end(), a.begin(), a.end());
        if (Peek(',')) Expect(',');
        else break;
      }
    } else if (id == "Offset") {
      out = Expr();
      Expect(',');
      int k = Int();
      if (Peek(',')) {  // Offset(d, t, x): the x offset is not used by any graph here
        Expect(',');
        if (Int() != 0) throw KioError("descriptor: Offset with x-offset is not supported");
      }
      for (DescTerm& t : out) t.offset += k;
    } else if (id == "Round") {
      out = Expr();
      Expect(',');
      (void)Int();  // Round(x, 1) is the identity; other moduli only matter for subsampled stats
    } else {
      throw KioError("descriptor operator '" + id + "' is not supported (only Append/Offset/Round) in '" + s + "'");
    }
    Expect(')');
    return out;
  }
};
}  // namespace

std::vector<DescTerm> FlattenDescriptor(const std::string& text) {
  DescParser dp(text);
  std::vector<DescTerm> out = dp.Expr();
  dp.Ws();
  if (dp.p != text.size()) throw KioError("trailing characters in descriptor '" + text + "'");
  return out;
}

RawNode ParseConfigLine(const std::string& raw) {
  RawNode n;
  std::string line = raw;
  size_t hash = line.find('#');
  if (hash != std::string::npos) line = line.substr(0, hash);
  line = TrimS(line);
  if (line.empty()) return n;
  n.line = line;
  size_t sp = line.find_first_of(" \t");
  n.kind = line.substr(0, sp);
  std::string rest = sp == std::string::npos ? "" : line.substr(sp + 1);
  // key=value pairs; values may contain spaces inside parentheses -> split at " key=" boundaries
  size_t p = 0;
  while (p < rest.size()) {
    while (p < rest.size() && isspace((unsigned char)rest[p])) ++p;
    size_t eq = rest.find('=', p);
    if (eq == std::string::npos) break;
    std::string key = rest.substr(p, eq - p);
    size_t v0 = eq + 1, q = v0;
    int depth = 0;
    while (q < rest.size()) {
      if (rest[q] == '(') ++depth;
      else if (rest[q] == ')') --depth;
      else if (isspace((unsigned char)rest[q]) && depth == 0) {
        // a value ends at whitespace that is followed by "key="
        size_t r = q;
        while (r < rest.size() && isspace((unsigned char)rest[r])) ++r;
        size_t e = r;
        while (e < rest.size() && (isalnum((unsigned char)rest[e]) || rest[e] == '-' || rest[e] == '_')) ++e;
        if (r == rest.size() || (e > r && e < rest.size() && rest[e] == '=')) break;
      }
      ++q;
    }
    std::string val = TrimS(rest.substr(v0, q - v0));
    if (key == "name") n.name = val;
    else if (key == "component") n.component = val;
    else if (key == "input") n.input = val;
    else if (key == "dim") n.dim = atoi(val.c_str());
    p = q;
  }
  return n;
}

// ------------------------------------------------------------------------------- reader
static void ReadComponentBody(Input& in, bool binary, RawComponent* c) {
  const std::string close = "</" + c->type + ">";
  std::string tok;
  for (;;) {
    ReadToken(in, binary, &tok);
    if (tok == close) break;
    if (tok.size() < 2 || tok[0] != '<') throw KioError("component " + c->name + ": unexpected token " + tok);
    // classify the payload
    enum { kScalar, kVec, kMat } kind = kScalar;
    if (binary) {
      int c0 = in.PeekAt(0), c1 = in.PeekAt(1), c2 = in.PeekAt(2);
      if ((c0 == 'F' || c0 == 'D') && c1 == 'V' && c2 == ' ') kind = kVec;
      else if ((c0 == 'F' || c0 == 'D') && c1 == 'M' && c2 == ' ') kind = kMat;
      else if (c0 == 'C' && c1 == 'M') kind = kMat;
    } else {
      size_t k = 0;
      while (in.PeekAt(k) >= 0 && isspace(in.PeekAt(k))) ++k;
      if (in.PeekAt(k) == '[') kind = kMat;  // a text vector is a one-row matrix
    }
    if (kind == kScalar) {
      if (binary) {
        int c0 = in.PeekAt(0);
        if (c0 == 'T' || c0 == 'F') c->scalar[tok] = ReadBool(in, true) ? 1.0 : 0.0;
        else if (c0 == 4) {
          // int32 and float share the size byte; keep both readings apart by token knowledge where it
          // matters (dims are ints), otherwise store the float interpretation
          int32_t iv;
          float fv;
          in.Get();
          char b[4];
          in.Read(b, 4);
          memcpy(&iv, b, 4);
          memcpy(&fv, b, 4);
          static const char* kIntTokens[] = {"<Dim>", "<BlockDim>", "<InputDim>", "<OutputDim>", "<InputPeriod>",
                                             "<OutputPeriod>", "<LeftContext>", "<RightContext>",
                                             "<NumLogCountFeatures>", "<RankIn>", "<RankOut>", "<UpdatePeriod>",
                                             "<Rank>", "<TimePeriod>", "<DropoutPerFrame>"};
          bool is_int = false;
          for (const char* t : kIntTokens) is_int = is_int || tok == t;
          c->scalar[tok] = is_int ? (double)iv : (double)fv;
        } else if (c0 == 8) {
          c->scalar[tok] = ReadFloatOrDouble(in, true);
        } else {
          throw KioError("component " + c->name + ": cannot decode field " + tok);
        }
      } else {
        size_t k = 0;
        while (in.PeekAt(k) >= 0 && isspace(in.PeekAt(k))) ++k;
        int c0 = in.PeekAt(k), c1 = in.PeekAt(k + 1);
        if ((c0 == 'T' || c0 == 'F') && (c1 < 0 || isspace(c1))) c->scalar[tok] = ReadBool(in, false) ? 1.0 : 0.0;
        else c->scalar[tok] = (double)(float)ReadFloatOrDouble(in, false);  // BaseFloat fields: same rounding as the binary flavour
      }
      continue;
    }
    if (binary && kind == kVec) {
      std::vector<float> v;
      ReadVector(in, true, &v);
      if (tok == "<BiasParams>") c->bias = std::move(v);
      else if (tok == "<StatsMean>") c->stats_mean = std::move(v);
      else if (tok == "<StatsVar>") c->stats_var = std::move(v);
      continue;
    }
    Matrix m;
    ReadMatrix(in, binary, &m);
    if (tok == "<LinearParams>" || tok == "<Params>") {
      c->linear = std::move(m);
      c->has_linear = true;
    } else if (tok == "<BiasParams>") {
      c->bias = std::move(m.data);
    } else if (tok == "<StatsMean>") {
      c->stats_mean = std::move(m.data);
    } else if (tok == "<StatsVar>") {
      c->stats_var = std::move(m.data);
    }
  }
}

void RawNnet::Read(const std::string& bytes) {
  source_ = bytes;
  nodes.clear();
  components.clear();
  Input in;
  in.OpenMemory(source_.data(), source_.size());
  binary = ReadBinaryHeader(in);
  ExpectToken(in, binary, "<Nnet3>");
  // config section: lines up to the first blank line
  std::string line;
  bool first = true;
  for (;;) {
    line.clear();
    int c;
    while ((c = in.Get()) >= 0 && c != '\n') line.push_back((char)c);
    if (c < 0) throw KioError("end of input inside the nnet3 config section");
    if (TrimS(line).empty()) {
      if (first) {  // the remainder of the "<Nnet3>" line
        first = false;
        continue;
      }
      break;
    }
    first = false;
    RawNode n = ParseConfigLine(line);
    if (!n.kind.empty()) nodes.push_back(n);
  }
  ExpectToken(in, binary, "<NumComponents>");
  int32_t n = ReadInt32(in, binary);
  if (n < 0 || n > 100000) throw KioError("implausible <NumComponents>");
  for (int i = 0; i < n; ++i) {
    ExpectToken(in, binary, "<ComponentName>");
    RawComponent c;
    ReadToken(in, binary, &c.name);
    // skip whitespace before the opening tag (text mode) to get a clean span
    while (in.PeekAt(0) >= 0 && isspace(in.PeekAt(0))) in.Get();
    c.span_begin = in.Tell();
    std::string open;
    ReadToken(in, binary, &open);
    if (open.size() < 3 || open.front() != '<' || open.back() != '>')
      throw KioError("bad component opening tag " + open);
    c.type = open.substr(1, open.size() - 2);
    ReadComponentBody(in, binary, &c);
    c.span_end = in.Tell();
    components.push_back(std::move(c));
  }
  ExpectToken(in, binary, "</Nnet3>");
}

void RawNnet::ReadFrom(const std::string& rxfilename) {
  Input in;
  in.Open(rxfilename);
  std::string bytes;
  // slurp: the model is the only object behind this rxfilename in every call site of the reference.  (Block reads: byte by
  // byte through fgetc the 28 MB of the x-vector model took 40 ms - 0.5 s on a slow host - of a job's start-up.)
  for (;;) {
    const size_t have = bytes.size(), want = std::max<size_t>(1 << 20, have);
    bytes.resize(have + want);
    const size_t got = in.ReadUpTo(&bytes[have], want);
    bytes.resize(have + got);
    if (got < want) break;
  }
  int status = in.Close();
  if (bytes.empty()) throw KioError("no model data read from '" + rxfilename + "'" +
                                    (status ? " (input command exited with status " + std::to_string(status) + ")" : ""));
  Read(bytes);
}

void RawNnet::ApplyNnetConfig(const std::string& text) {
  size_t a = 0;
  while (a <= text.size()) {
    size_t b = text.find('\n', a);
    std::string line = text.substr(a, b == std::string::npos ? std::string::npos : b - a);
    a = (b == std::string::npos) ? text.size() + 1 : b + 1;
    RawNode n = ParseConfigLine(line);
    if (n.kind.empty()) continue;
    if (n.kind == "component") throw KioError("--nnet-config lines that add components are not supported");
    bool replaced = false;
    for (RawNode& o : nodes)
      if (o.kind == n.kind && o.name == n.name) {
        o = n;
        replaced = true;
      }
    if (!replaced) nodes.push_back(n);
  }
}

const RawNode* RawNnet::FindNode(const std::string& name, const char* kind) const {
  for (const RawNode& n : nodes)
    if (n.name == name && (!kind || n.kind == kind)) return &n;
  return nullptr;
}

const RawComponent* RawNnet::FindComponent(const std::string& name) const {
  for (const RawComponent& c : components)
    if (c.name == name) return &c;
  return nullptr;
}

void RawNnet::Write(Output& out, bool binary_out) const {
  if (binary_out != binary)
    throw KioError("changing the binary/text flavour of a model is not supported by this writer");
  if (binary_out) out.Write("\0B", 2);
  WriteToken(out, binary_out, "<Nnet3>");
  out.Put('\n');
  for (const RawNode& n : nodes) {
    out.Puts(n.line);
    out.Put('\n');
  }
  out.Put('\n');
  WriteToken(out, binary_out, "<NumComponents>");
  WriteInt32(out, binary_out, (int32_t)components.size());
  if (!binary_out) out.Put('\n');
  for (const RawComponent& c : components) {
    WriteToken(out, binary_out, "<ComponentName>");
    WriteToken(out, binary_out, c.name.c_str());
    out.Write(source_.data() + c.span_begin, c.span_end - c.span_begin);
    if (!binary_out) out.Put('\n');
  }
  WriteToken(out, binary_out, "</Nnet3>");
  if (!binary_out) out.Put('\n');
}

}  // namespace xv
