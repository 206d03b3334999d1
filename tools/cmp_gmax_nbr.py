import os, sys, json, importlib
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
import helpers as H
P = H.pkg()
def run(tag, net, line, topo):
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    utts = [H.features(500 + i, T) for i, T in enumerate([400]*24 + [314]*8)]
    feats, offs = H.pack(utts)
    ctx = P.Context(model)
    cal = ctx.calibrate(feats, offs, 7.5e-5)
    # mean errors too
    ctx.set_fast_mode("fp16x3"); ref = ctx.forward_batch(feats, offs)
    out = {}
    for m in ("fp16mx", "fp16mx2"):
        ctx.set_fast_mode(m); o = ctx.forward_batch(feats, offs)
        e = np.abs(o - ref).max(1) / np.abs(ref).max(1)
        out[m] = (float(e.mean()), float(e.max()))
    print(os.environ.get("XVEC_LIB","cur").split("/")[-1], tag, {k: "%.2e / %.2e" % v for k, v in out.items()})
for tag, (net, line), topo in (("init123", H.synth_model("v2_xvector", 123), "v2"), ("init7", H.synth_model("v2_xvector", 7), "v2"),
                          ("trained11", H.trained_like_model("v2_xvector", 11), "v2"), ("trained12", H.trained_like_model("v2_xvector", 12), "v2"),
                          ("v5init", H.synth_model("v5_cvector"), "v5"), ("v5trained", H.trained_like_model("v5_cvector", 11), "v5")):
    run(tag, net, line, topo)
