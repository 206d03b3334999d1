#!/usr/bin/env python3
"""One single-threaded CPU-oracle worker of bench.py's cpu_baseline leg (TEST INFRASTRUCTURE: imports oracle/).

Mirrors how the reference runs extraction on CPU: `nj` independent single-threaded processes, one utterance at a
time (egs/sre/v2/run_sre10.sh:24,200: --nj 32 --use-gpu false).  Usage: worker.py <topology> <frames> <seconds> <seed>
Prints "<utterances> <seconds>"."""
import os
import sys
import time

for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[v] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import helpers as H  # noqa: E402


def main():
    topo, frames, seconds, seed = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
    net, line = H.synth_model(topo)
    n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
    n2.apply_nnet_config(line)
    ev = H.xo.GraphEvaluator(n2, np.float32)
    x = [H.features(seed * 4 + i, frames) for i in range(4)]
    ev.compute(x[0])
    n, t0 = 0, time.time()
    while time.time() - t0 < seconds:
        ev.compute(x[n % 4])
        n += 1
    print(n, time.time() - t0)


if __name__ == "__main__":
    main()
