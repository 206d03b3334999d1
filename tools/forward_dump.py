"""Runs one forward pass of a synthetic model and saves the embeddings (GPU box).  Used by tests/test_gpu_variants.py
to compare GEMM variants (XVEC_GEMM_VARIANT / XVEC_SK_MF are read once per process) bit for bit.
usage: forward_dump.py <out.npy> <topology> <precision name> <n chunks> <frames> [ragged]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402

out, topo, prec, n, T = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
ragged = len(sys.argv) > 6
P = H.pkg()
net, line = H.synth_model(topo)
model = P.Model(raw=net.to_bytes(True), nnet_config=line)
ctx = P.Context(model, precision=P.PRECISIONS[prec])
rng = np.random.default_rng(11)
lens = rng.integers(T // 2, T + T // 2, n) if ragged else np.full(n, T)
pool = [H.features(2000 + i, int(t)) for i, t in enumerate(lens[:16])]
utts = [pool[i % 16] for i in range(n)]
feats, offs = H.pack(utts)
np.save(out, ctx.forward_batch(feats, offs))
