"""Diagnostic (GPU box): embedding error of each precision mode vs the fp32 / fp64 oracle, per chunk length."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
P = H.pkg()
net, line = H.synth_model("v2_xvector")
model = P.Model(raw=net.to_bytes(True), nnet_config=line)
n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True)); n2.apply_nnet_config(line)
ev32 = H.xo.GraphEvaluator(n2, np.float32); ev64 = H.xo.GraphEvaluator(n2, np.float64)
for prec in (0, 3, 4, 5, 1, 2):
    ctx = P.Context(model, precision=prec)
    for T in (400, 330, 314, 200, 57, 15, 16, 25, 137, 1000):
        x = H.features(T, T)
        out = ctx.forward_batch(x, [0, T])
        r32 = ev32.compute(x); r64 = ev64.compute(x)
        l2 = np.linalg.norm(out - r64) / np.linalg.norm(r64)
        print("prec %d T %4d  maxrel vs fp32 %.3e  vs fp64 %.3e  l2rel %.3e  (fp32 oracle vs fp64 %.3e) |e|max %.2f" % (
            prec, T, H.rel_err(out, r32), H.rel_err(out, r64), l2, H.rel_err(r32, r64), np.abs(r64).max()))
