"""GPU: the fp16mx forward pass of one big batch (264 chunks) dumped to .npy, to be run once per GEMM variant
(XVEC_GEMM_VARIANT is read once per process) and compared: check_mx_variants.py dump <file> | cmp <a> <b>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402

if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    d = np.abs(a - b).max(axis=1)
    print("chunks that differ: %d of %d; max |diff| %.3g; first differing: %s" % ((d > 0).sum(), len(d), d.max(), np.where(d > 0)[0][:10]))
    sys.exit(0)
P = H.pkg()
net, line = H.synth_model("v2_xvector", 7)
model = P.Model(raw=net.to_bytes(True), nnet_config=line)
lens = (400, 400, 314, 400, 1000, 400, 400, 333)
utts = [H.features(i, T) for i, T in enumerate(lens)]
big = [utts[0]] * 3 + [utts[i % len(utts)] for i in range(261)]
if len(sys.argv) > 3 and sys.argv[3] == "small":
    big = utts
fb, ob = H.pack(big)
ctx = P.Context(model, device=0, precision=P.PRECISIONS[os.environ.get("CHECK_PREC", "fp16mx")])
out = ctx.forward_batch(fb, ob)
out2 = ctx.forward_batch(fb, ob)
print("same call twice identical:", bool(np.array_equal(out, out2)), " chunk0 == chunk1 == chunk2:", bool(np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2])))
np.save(sys.argv[2], out)
