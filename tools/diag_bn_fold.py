"""GPU box: which layer's folded BatchNorm moves the error of the fast arithmetics?  One model, 32 chunks of 400 frames, each
fast mode against the fp64 oracle with XVEC_DEBUG=bn_fold_mask = nothing / one layer at a time / everything (a fresh process per
setting: the mask is read when the model is lowered).  usage: diag_bn_fold.py [v2|v5] [trained seed | init]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 3 and sys.argv[3] == "--worker":
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import helpers as H
    topo = "v5_cvector" if sys.argv[1] == "v5" else "v2_xvector"
    net, line = H.synth_model(topo) if sys.argv[2] == "init" else H.trained_like_model(topo, int(sys.argv[2]))
    P = H.pkg()
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    utts = [H.features(20000 + i, 400) for i in range(32)]
    feats, offs = H.pack(utts)
    n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
    n2.apply_nnet_config(line)
    ev = H.xo.GraphEvaluator(n2, np.float64)
    ref = np.stack([ev.compute(u)[0] for u in utts])
    res = []
    for mode in ("fp16x3", "fp16mx2", "auto", "fp16x2"):
        out = P.Context(model, precision=P.PRECISIONS[mode]).forward_batch(feats, offs)
        e = np.abs(out.astype(np.float64) - ref).max(axis=1) / np.abs(ref).max(axis=1)
        res.append("%s worst %.2e mean %.2e" % (mode, e.max(), e.mean()))
    print("   ".join(res), flush=True)
    sys.exit(0)
which, seed = (sys.argv[1] if len(sys.argv) > 1 else "v2"), (sys.argv[2] if len(sys.argv) > 2 else "11")
masks = [("none", "0"), ("all", str((1 << 63) - 1))] + [("layer %d" % i, str(1 << i)) for i in range(4 if which == "v2" else 9)]
for name, m in masks:
    r = subprocess.run([sys.executable, os.path.abspath(__file__), which, seed, "--worker"], env=dict(os.environ, XVEC_DEBUG="bn_fold=1,bn_fold_mask=%s" % m),
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    print("%s %s fold %-8s %s" % (which, seed, name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED"), flush=True)
