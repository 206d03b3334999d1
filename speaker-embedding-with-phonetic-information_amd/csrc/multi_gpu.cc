#include "multi_gpu.h"

#include <dlfcn.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <sstream>
#include <thread>

#include <hip/hip_runtime.h>

namespace xv {

namespace {

// RCCL is bound lazily so that the library loads (and every non-collective entry works) on hosts where librccl cannot
// initialise.
struct Rccl {
  typedef void* comm_t;
  typedef int (*CommInitAll_t)(comm_t*, int, const int*);
  typedef int (*Broadcast_t)(const void*, void*, size_t, int, int, comm_t, hipStream_t);
  typedef int (*Group_t)(void);
  typedef int (*CommDestroy_t)(comm_t);
  typedef int (*CommAbort_t)(comm_t);
  CommInitAll_t comm_init_all = nullptr;
  Broadcast_t bcast = nullptr;
  Group_t gstart = nullptr, gend = nullptr;
  CommDestroy_t cdestroy = nullptr;
  CommAbort_t cabort = nullptr;
  Rccl() {
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) throw EngineError(std::string("cannot load librccl: ") + dlerror());
    comm_init_all = (CommInitAll_t)dlsym(lib, "ncclCommInitAll");
    bcast = (Broadcast_t)dlsym(lib, "ncclBroadcast");
    gstart = (Group_t)dlsym(lib, "ncclGroupStart");
    gend = (Group_t)dlsym(lib, "ncclGroupEnd");
    cdestroy = (CommDestroy_t)dlsym(lib, "ncclCommDestroy");
    cabort = (CommAbort_t)dlsym(lib, "ncclCommAbort");
    if (!comm_init_all || !bcast || !gstart || !gend || !cdestroy) throw EngineError("librccl lacks expected symbols");
  }
};

}  // namespace

std::vector<int> ParseDeviceList(const std::string& spec, int n_visible) {
  std::vector<int> d;
  if (spec.empty()) return d;
  if (spec == "all") {
    for (int i = 0; i < n_visible; ++i) d.push_back(i);
    return d;
  }
  std::stringstream ss(spec);
  std::string tok;
  while (std::getline(ss, tok, ',')) {
    char* end = nullptr;
    const long v = strtol(tok.c_str(), &end, 10);
    if (tok.empty() || *end || v < 0 || v >= n_visible)
      throw EngineError("bad device list \"" + spec + "\": \"" + tok + "\" is not one of the " + std::to_string(n_visible) + " visible devices");
    if (std::find(d.begin(), d.end(), (int)v) != d.end()) throw EngineError("bad device list \"" + spec + "\": device " + tok + " twice");
    d.push_back((int)v);
  }
  if (d.empty()) throw EngineError("bad device list \"" + spec + "\"");
  return d;
}

std::vector<std::unique_ptr<Engine>> CreateEnginesBroadcast(const std::vector<uint8_t>& blob, const std::vector<int>& devices,
                                                            double timeout_s) {
  const int n = (int)devices.size();
  if (n < 1) throw EngineError("CreateEnginesBroadcast: no device");
  if (timeout_s <= 0.0) {
    const char* e = getenv("XVEC_BCAST_TIMEOUT");
    timeout_s = (e && *e && atof(e) > 0.0) ? atof(e) : 120.0;
  }
  static Rccl rccl;   // throws when librccl cannot be bound
  std::vector<Rccl::comm_t> comms(n, nullptr);
  if (rccl.comm_init_all(comms.data(), n, devices.data()) != 0) throw EngineError("ncclCommInitAll failed");
  std::vector<void*> dbuf(n, nullptr);
  std::vector<hipStream_t> st(n, nullptr);
  std::vector<std::unique_ptr<Engine>> engines;
  std::string err;
  bool stalled = false;
  for (int i = 0; i < n && err.empty(); ++i) {
    if (hipSetDevice(devices[i]) != hipSuccess || hipMalloc(&dbuf[i], blob.size()) != hipSuccess ||
        hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking) != hipSuccess)
      err = "device allocation for the weight broadcast failed on device " + std::to_string(devices[i]);
  }
  if (err.empty()) {
    (void)hipSetDevice(devices[0]);
    if (hipMemcpy(dbuf[0], blob.data(), blob.size(), hipMemcpyHostToDevice) != hipSuccess) err = "upload of the weight blob failed";
    // the receivers' buffers start without a valid header: a broadcast that was never enqueued (a failed group call) must not
    // leave a recycled allocation's stale - and valid - image of an earlier call behind for ReadBlobHead to accept
    for (int i = 1; i < n && err.empty(); ++i) {
      (void)hipSetDevice(devices[i]);
      if (hipMemset(dbuf[i], 0, std::min<size_t>(blob.size(), 256)) != hipSuccess) err = "clearing the receive buffer failed";
    }
  }
  if (err.empty()) {
    // ONE broadcast of the packed image, ncclChar elements (type id 0), root = rank 0
    // (errors of grouped calls surface at ncclGroupEnd: nothing is enqueued then, and the stream queries below would pass at once)
    if (rccl.gstart() != 0) err = "ncclGroupStart failed";
    for (int i = 0; i < n && err.empty(); ++i) {
      (void)hipSetDevice(devices[i]);
      if (rccl.bcast(dbuf[i], dbuf[i], blob.size(), /*ncclChar*/ 0, 0, comms[i], st[i]) != 0) err = "ncclBroadcast failed";
    }
    if (rccl.gend() != 0 && err.empty()) err = "ncclGroupEnd failed (the weight broadcast was not enqueued)";
  }
  if (err.empty()) {
    // bounded wait: a broadcast that does not complete becomes an error message and a non-zero exit of
    // the caller instead of a silent hang of the job
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n && err.empty(); ++i) {
      (void)hipSetDevice(devices[i]);
      for (;;) {
        const hipError_t q = hipStreamQuery(st[i]);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) {
          err = std::string("the weight broadcast failed on device ") + std::to_string(devices[i]) + ": " + hipGetErrorString(q);
          break;
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
          std::ostringstream m;
          m << "the weight broadcast over RCCL did not complete within " << timeout_s << " s (device " << devices[i]
            << " of " << n << "; XVEC_BCAST_TIMEOUT sets the limit)";
          err = m.str();
          stalled = true;
          break;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
      }
    }
  }
  // every engine is built from the bytes *its device received* (device to device), so a bad broadcast cannot go unnoticed:
  // header and layer table are read back from that copy and validated like any blob
  for (int i = 0; i < n && err.empty(); ++i) {
    (void)hipSetDevice(devices[i]);
    try {
      std::vector<uint8_t> head = ReadBlobHead(dbuf[i], blob.size());
      engines.emplace_back(new Engine(head.data(), blob.size(), devices[i], dbuf[i]));
    } catch (const std::exception& e) {
      err = std::string("context from the broadcast image on device ") + std::to_string(devices[i]) + ": " + e.what();
    }
  }
  if (stalled) {
    // the collective is still pending on the streams: freeing its buffers or destroying the communicators would block or fault.
    // Abort the communicators where the library allows it and leave the rest to process exit (the caller is about to fail).
    for (int i = 0; i < n; ++i)
      if (comms[i] && rccl.cabort) rccl.cabort(comms[i]);
    throw EngineError(err);
  }
  for (int i = 0; i < n; ++i) {
    (void)hipSetDevice(devices[i]);
    if (dbuf[i]) (void)hipFree(dbuf[i]);
    if (st[i]) (void)hipStreamDestroy(st[i]);
    if (comms[i]) rccl.cdestroy(comms[i]);
  }
  if (!err.empty()) throw EngineError(err);
  return engines;
}

}  // namespace xv
