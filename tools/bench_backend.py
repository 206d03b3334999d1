#!/usr/bin/env python3
"""Times the speaker-level back-end (subtract mean -> LDA 512x150 -> length normalisation) through the C ABI:
host-buffer rate (PCIe-inclusive) here; the kernel time itself comes from rocprofv3 --kernel-trace on this script
(profiles/r01c_backend_kernel_stats.csv).  Checks the result against the oracle on a sample."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from oracle import backend as B  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
    P = H.pkg()
    rng = np.random.default_rng(1)
    x = rng.standard_normal((n, 512), dtype=np.float32)
    mean = rng.standard_normal(512, dtype=np.float32)
    lda = (rng.standard_normal((150, 512), dtype=np.float32) / 20).astype(np.float32)
    y = P.backend_apply(x, mean, lda, normalize=True)
    ref, _ = B.backend_chain(x[:256], mean, lda, normalize=True)
    err = H.rel_err(y[:256], ref)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        P.backend_apply(x, mean, lda, normalize=True)
    dt = (time.perf_counter() - t0) / reps
    segs = [list(range(i, min(i + 8, n))) for i in range(0, min(n, 65536), 8)]
    t0 = time.perf_counter()
    P.segment_mean(x, segs)
    dts = time.perf_counter() - t0
    print(json.dumps({"vectors": n, "backend_host_ms": dt * 1e3, "backend_host_vectors_per_s": n / dt,
                      "alg_bytes_per_vector": 512 * 4 + 150 * 4, "segment_mean_host_ms": dts * 1e3, "segments": len(segs),
                      "rel_err_vs_oracle": err}))


if __name__ == "__main__":
    main()
