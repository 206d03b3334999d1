// See calib_file.h.
#include "calib_file.h"

#include <errno.h>
#include <inttypes.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <fstream>
#include <sstream>

#include "engine.h"
#include "kio.h"

namespace xv {

namespace {
int PrecisionFromName(const std::string& n) {
  for (int p = 0; p <= 9; ++p)
    if (n == PrecisionName(p)) return p;
  return -1;
}
}  // namespace

bool ReadCalibrationFile(const std::string& path, SharedChoice* out) {
  std::ifstream f(path);
  if (!f.good()) {
    if (access(path.c_str(), F_OK) != 0) return false;
    throw KioError("calibration file " + path + " exists but cannot be read");
  }
  SharedChoice c;
  std::string line, magic;
  int version = 0;
  bool have_model = false, have_prec = false;
  if (!std::getline(f, line)) throw KioError("calibration file " + path + " is empty");
  {
    std::istringstream h(line);
    h >> magic >> version;
    if (magic != "xvec-calibration" || version != 1)
      throw KioError("calibration file " + path + ": expected a first line 'xvec-calibration 1', found '" + line + "'");
  }
  while (std::getline(f, line)) {
    std::istringstream l(line);
    std::string key;
    if (!(l >> key)) continue;
    if (key == "model") {
      std::string v;
      l >> v;
      char* end = nullptr;
      c.model = strtoull(v.c_str(), &end, 16);
      have_model = end && *end == 0 && !v.empty();
    } else if (key == "precision") {
      std::string v;
      l >> v;
      c.precision = PrecisionFromName(v);
      have_prec = c.precision == kPrecFp16Mx || c.precision == kPrecFp16Mx2 || c.precision == kPrecFp16x3;
    } else if (key == "lite-mask") {
      std::string v;
      l >> v;
      c.lite_mask = strtoull(v.c_str(), nullptr, 16);
    } else if (key == "tolerance") {
      l >> c.tolerance;
    } else if (key == "note") {
      std::getline(l, c.note);
      if (!c.note.empty() && c.note[0] == ' ') c.note.erase(0, 1);
    }   // unknown keys: a later version's information, ignored
  }
  if (!have_model || !have_prec) throw KioError("calibration file " + path + ": no valid 'model' / 'precision' line");
  if (c.lite_mask && c.precision != kPrecFp16Mx2)
    throw KioError("calibration file " + path + ": a lite-mask goes with precision fp16mx2 only");
  *out = c;
  return true;
}

bool PublishCalibrationFile(const std::string& path, const SharedChoice& mine, SharedChoice* adopted) {
  bool published = false;
  if (access(path.c_str(), F_OK) != 0) {
    char host[64] = "host";
    (void)gethostname(host, sizeof host - 1);
    static std::atomic<unsigned> serial{0};   // (several contexts of one process may publish at once)
    std::ostringstream tmp;
    tmp << path << ".tmp." << host << "." << (long)getpid() << "." << serial.fetch_add(1);
    {
      std::ofstream f(tmp.str(), std::ios::trunc);
      char model[32];
      snprintf(model, sizeof model, "%016" PRIx64, mine.model);
      char mask[32];
      snprintf(mask, sizeof mask, "%" PRIx64, mine.lite_mask);
      f << "xvec-calibration 1\n"
        << "model " << model << "\n"
        << "precision " << PrecisionName(mine.precision) << "\n"
        << "lite-mask " << mask << "\n"
        << "tolerance " << mine.tolerance << "\n"
        << "note " << mine.note << "\n";
      f.flush();
      if (!f.good()) {
        (void)unlink(tmp.str().c_str());
        throw KioError("cannot write the calibration file " + tmp.str() + " (the choice of arithmetic must be shared: give a writable --calibration path)");
      }
    }
    // link(2) fails with EEXIST when another job published first: atomic on local file systems and on NFS
    if (link(tmp.str().c_str(), path.c_str()) == 0) {
      published = true;
    } else if (errno != EEXIST) {
      const std::string why = strerror(errno);
      (void)unlink(tmp.str().c_str());
      throw KioError("cannot publish the calibration file " + path + ": " + why);
    }
    (void)unlink(tmp.str().c_str());
  }
  if (!ReadCalibrationFile(path, adopted)) throw KioError("calibration file " + path + " vanished after it was published");
  return published;
}

void AdoptSharedChoice(Engine* engine, const SharedChoice& sc, const std::string& path) {
  if (!engine->can_switch_fast_mode())
    throw KioError("calibration file " + path + ": this context cannot switch its arithmetic (it must be packed as the default, fp16mx2)");
  if (sc.model != engine->info().fingerprint) {
    char a[32], b[32];
    snprintf(a, sizeof a, "%016" PRIx64, sc.model);
    snprintf(b, sizeof b, "%016" PRIx64, engine->info().fingerprint);
    throw KioError("calibration file " + path + " was measured on another model image (" + a + ", this one is " + b +
                   "): remove it, or point --calibration / XVEC_CALIBRATION at this model's file");
  }
  engine->SetFastMode(sc.precision);
  if (sc.lite_mask) {
    engine->SetLiteMask(sc.lite_mask);
    if (engine->lite_mask() != sc.lite_mask)
      throw KioError("calibration file " + path + ": its lite-mask names layers this model cannot run in 1.25 passes");
  }
}

}  // namespace xv
