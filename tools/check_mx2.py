"""GPU box: embedding error of fp16mx2 / fp16mx / fp16x3 against the fp64 oracle on three models (Kaldi-initialisation-like,
two heavy-tailed BatchNorm-calibrated ones) over chunk lengths 25 .. 400, and solo == batched for the first chunk.
usage: python tools/check_mx2.py"""
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import helpers as H
P = H.pkg()
for name, (net, line) in {"init 123": H.synth_model("v2_xvector", 123), "trained 11": H.trained_like_model("v2_xvector", 11), "trained 12": H.trained_like_model("v2_xvector", 12)}.items():
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True)); n2.apply_nnet_config(line)
    ev64 = H.xo.GraphEvaluator(n2, np.float64)
    utts = [H.features(9100 + i, T) for i, T in enumerate([400, 137, 400, 314, 60, 400, 25, 200] * 2)]
    feats, offs = H.pack(utts)
    ref = [ev64.compute(u) for u in utts]
    for pn in ("fp16mx2", "fp16mx", "fp16x3"):
        ctx = P.Context(model, precision=P.PRECISIONS[pn])
        out = ctx.forward_batch(feats, offs)
        errs = [H.rel_err(out[i:i + 1], ref[i]) for i in range(len(utts))]
        solo = ctx.forward_batch(*H.pack(utts[:1]))
        print("%-12s %-8s max %.2e mean %.2e  (T=400: %.2e, T=137: %.2e, T=25: %.2e)  solo==batched: %s" % (name, pn, max(errs), np.mean(errs), errs[0], errs[1], errs[6], np.array_equal(solo[0], out[0])))
