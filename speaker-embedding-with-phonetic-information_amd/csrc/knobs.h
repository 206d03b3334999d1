// Developer and diagnostic switches: ONE environment variable, XVEC_DEBUG="name=value,name=value".
//
// What a user sets are command-line options and the handful of XVEC_* variables INTEGRATION.md lists (XVEC_DEVICE / XVEC_DEVICES,
// XVEC_CALIBRATION, XVEC_LANES, XVEC_TIMING, XVEC_BCAST_TIMEOUT, XVEC_FAST_MIN_POOLED, and XVEC_CMN_WINDOW / XVEC_VAD_RSPECIFIER
// for the C ABI's table entry points).  Everything else - A/B switches between kernel families, test knobs, study knobs of the
// tools - lives in this one list, so that the shipped path has one behaviour and the test matrix knows what it is (VERDICT r05
// item 9: nineteen separate variables, several read through statics inside PackModel).  Names (value, default):
//   gemm_variant (0 | 1 | 2 | 4, 0)   kernel family of the 32-column GEMMs: 2 = per-tile kernels instead of the persistent grid
//   sk_mf (4 | 8, 8)  sk_lanes (1 | 2 | 4, 4)  gemm_stagger (percent, 85)   geometry of those kernels
//   p8 (0 | 1, 1)                     0: never run tdnn_gemm_kernel_p8 (results differ in the last bits: another K order)
//   p8_whole (0 | 1 | 2, 0)           how that kernel deals its tiles out (engine.cc); read per context
//   first_kernel (0 | 1, 1)           0: layers on the network input go through prep_input + the generic GEMM
//   readers (1..16, 4 per engine)  copy_threads (0..15, 3)  pipe_drain (0 | 1, 1)  mmap (0 | 1, 1)  spin_wait (0 | 1, 0)
//   cm_on_device (0 | 1, 1)
//                                     host side of a table job: reader / copy threads, the pipe's drain thread, mapped archives,
//                                     hipEventSynchronize instead of the sleeping wait, compressed matrices of a front-end
//                                     job expanded on the GPU (0: by the reader threads)
//   engines_on_one_device (n, 0)      test knob: n engines on ONE device behind the several-engine table loop
//   bn_fold (0 | 1, 0)  bn_fold_mask (bits)   the opt-in lowering of DESIGN.md section 3.5
//   tail_over_tol (factor, 1.10)      study knob of tools/tail_error.py: the calibration's projected-tail condition
//   fuse_pipe (0 | 1, 1)              0: the reference's feature pipeline (fuse_pipe.h) is run as commands, not on the device
//   calib_fail (0 | 1 | 2 | 3, 0)     fault injection: the first / every Engine::Calibrate of the process throws; 3: the first one
//                                     sees one element of a pass move by an ulp (a device that does not reproduce its bits)
// The list is parsed on every call (callers that must not change their mind keep the answer in a static).
#pragma once
#include <stdlib.h>
#include <string.h>

#include <string>

namespace xv {

// The value of `name` in XVEC_DEBUG, or "" when it is not listed.
inline std::string DebugKnob(const char* name) {
  const char* e = getenv("XVEC_DEBUG");
  if (!e) return std::string();
  const size_t n = strlen(name);
  for (const char* p = e; *p;) {
    const char* end = strchr(p, ',');
    const size_t len = end ? (size_t)(end - p) : strlen(p);
    if (len > n && strncmp(p, name, n) == 0 && p[n] == '=') return std::string(p + n + 1, len - n - 1);
    p += len + (end ? 1 : 0);
  }
  return std::string();
}
inline int DebugKnobInt(const char* name, int dflt) {
  const std::string v = DebugKnob(name);
  return v.empty() ? dflt : atoi(v.c_str());
}

}  // namespace xv
