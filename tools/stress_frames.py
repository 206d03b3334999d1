"""GPU box: the soak test of tools/stress_streamk.py for frame-level outputs (BASELINE config 5: senone log-posteriors of the
multitask network, fp16 and fp16x3; bottleneck features): two contexts from two threads plus a noise stream, host-buffer
entry point, every result bit-identical to the first of its shape.  usage: stress_frames.py [iterations]"""
import os
import sys
import threading

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402

P = H.pkg()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfgs, _ = H.TOPOLOGIES["v3_multitask"]
net = H.nm.synthesize([H.config_text(c) for c in cfgs], seed=123, head_stddev=1.0)
bad = []
stop = []


def noise():
    st = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
    with torch.cuda.stream(st):
        while not stop:
            for _ in range(4):
                (a @ a)
            st.synchronize()


def worker(tag, node, prec):
    model = P.Model(raw=net.to_bytes(True), nnet_config="output-node name=output input=%s" % node)
    ctx = P.Context(model, device=0, precision=P.PRECISIONS[prec])
    shapes = [np.random.default_rng(5).integers(200, 601, 24), np.full(16, 400), np.random.default_rng(6).integers(30, 200, 40)]
    data = [H.pack([H.features(700 + 50 * k + i, int(T)) for i, T in enumerate(lens)]) for k, lens in enumerate(shapes)]
    refs = [None] * len(shapes)
    for it in range(N):
        k = it % len(shapes)
        r = ctx.forward_batch(*data[k])
        if refs[k] is None:
            refs[k] = r
            assert np.isfinite(r).all()
        elif not np.array_equal(refs[k], r):
            bad.append((tag, node, prec, it, k, float(np.abs(refs[k] - r).max())))


tn = threading.Thread(target=noise)
tn.start()
try:
    for node, prec in (("output_am.log-softmax", "fp16"), ("output_am.log-softmax", "fp16x3"), ("tdnn5_am.batchnorm", "fp16x3"),
                       ("tdnn5_am.batchnorm", "fp16")):
        ts = [threading.Thread(target=worker, args=("t%d" % i, node, prec)) for i in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        print("%s %s: %d iterations x 2 contexts, mismatches so far: %d %s" % (node, prec, N, len(bad), bad[:3]), flush=True)
finally:
    stop.append(1)
    tn.join()
os._exit(1 if bad else 0)
