"""GPU box: randomized regression of fp16mx2 - random batch sizes, chunk lengths and topologies; every embedding against the
three-pass arithmetic of the same model (fp32-grade), and a random chunk of every batch alone == inside the batch.
usage: python tools/fuzz_mx2.py [batches] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402

P = H.pkg()
n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = 0.0
bad = []
for topo in ("v2_xvector", "v5_cvector", "v3_multitask"):
    net, line = H.synth_model(topo)
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    fast = P.Context(model, precision=P.PRECISIONS["fp16mx2"])
    slow = P.Context(model, precision=P.PRECISIONS["fp16x3"])
    for b in range(n_batches):
        n = int(rng.choice([1, 2, 3, 7, 20, 60, 150, 300]))
        kind = rng.integers(0, 3)
        lens = (rng.integers(25, 1000, n) if kind == 0 else np.full(n, int(rng.choice([100, 137, 400, 777]))) if kind == 1
                else rng.choice([25, 60, 113, 114, 115, 400], n))
        if int(np.sum(lens)) > 120000:
            lens = lens[: max(1, int(120000 // max(lens)))]
        utts = [H.features(int(rng.integers(1, 10 ** 6)), int(T)) for T in lens]
        f, o = H.pack(utts)
        a = fast.forward_batch(f, o)
        r = slow.forward_batch(f, o)
        err = max(H.rel_err(a[i:i + 1], r[i:i + 1]) for i in range(len(utts)))
        worst = max(worst, err)
        k = int(rng.integers(0, len(utts)))
        solo = fast.forward_batch(*H.pack(utts[k:k + 1]))
        same = np.array_equal(solo[0], a[k])
        if not (err < 1e-4 and same and np.isfinite(a).all()):
            bad.append((topo, b, len(utts), float(err), bool(same)))
    print(topo, "worst so far %.2e" % worst, "bad:", len(bad))
    sys.stdout.flush()
print("batches per topology:", n_batches, "worst relative difference to fp16x3: %.2e" % worst, "failures:", bad[:5])
sys.exit(1 if bad else 0)
