"""Diagnostic (GPU box): pooled [mean|std] vector through an identity embedding layer; event profile sanity."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
P = H.pkg()
net = H.nm.synthesize(H.tiny_config(pool=16, emb=32), seed=5)
net.components["tdnn6.affine"].f["linear"] = np.eye(32, dtype=np.float32)
net.components["tdnn6.affine"].f["bias"] = np.zeros(32, np.float32)
net.apply_nnet_config("output-node name=output input=tdnn6.affine")
model = P.Model(raw=net.to_bytes(True))
ctx = P.Context(model, precision=0)
ev = H.xo.GraphEvaluator(net, np.float32)
for T in (15, 16):
    x = H.features(3, T, 5)
    out = ctx.forward_batch(x, [0, T])[0]; ref = ev.compute(x)[0]
    np.set_printoptions(precision=6, suppress=False, linewidth=200)
    print("T", T, "mean gpu", out[:6], "\n      mean ref", ref[:6], "\n      std gpu ", out[16:22], "\n      std ref ", ref[16:22])
# profile sanity on the v2 net
import torch
net2, line = H.synth_model("v2_xvector")
m2 = P.Model(raw=net2.to_bytes(True), nnet_config=line)
for prec in (0, 1, 2, 1):
    c = P.Context(m2, precision=prec)
    B, T = 256, 400
    feats = torch.randn(B*T, 23, device="cuda") * 3
    out = torch.empty(B, 512, device="cuda")
    offs = np.arange(B+1, dtype=np.int32)*T
    for _ in range(5): c.forward_batch_device(feats.data_ptr(), offs, out.data_ptr(), 512, None)
    torch.cuda.synchronize()
    c.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(20): c.forward_batch_device(feats.data_ptr(), offs, out.data_ptr(), 512, None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c.set_profiling(False)
    rep = c.profile_report()
    print("prec", prec, "ms/step %.3f" % (dt/20*1e3), [(l.split(':')[-1], round(ms/n, 4)) for l, n, ms in rep])
