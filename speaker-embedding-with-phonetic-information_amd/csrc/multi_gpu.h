// One process, several GPUs: the packed weight image is read / packed once on the host, uploaded to the
// first device and sent to the others with ONE ncclBroadcast (RCCL over xGMI); every engine is built from the bytes ITS
// device received, device to device.  The reference's counterpart is `nj` processes that each re-read the model
// (egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:72,83-93).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "engine.h"

namespace xv {

// devices: distinct HIP device ordinals, devices[0] is the root.  timeout_s: how long the broadcast may take before the call
// gives up with an EngineError (a rank that never joins would otherwise hang the job without a word); <= 0: XVEC_BCAST_TIMEOUT
// or 120 s.  Throws EngineError on any failure; nothing is left allocated then - except after a TIME-OUT: the collective is still
// pending on its streams, so buffers, streams and communicators are left to process exit (the communicators are aborted where
// the library allows it); the caller is expected to end the job.
std::vector<std::unique_ptr<Engine>> CreateEnginesBroadcast(const std::vector<uint8_t>& blob, const std::vector<int>& devices,
                                                            double timeout_s = 0.0);

// "--devices" / XVEC_DEVICES: "all", or a comma-separated list of ordinals ("0,1,2,3"; duplicates and ordinals >= n_visible
// are errors).  Empty string: empty list (the caller's single-device rule applies).
std::vector<int> ParseDeviceList(const std::string& spec, int n_visible);

}  // namespace xv
