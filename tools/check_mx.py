"""GPU check of the fp16mx mode: embedding error vs the fp64 oracle for fp16x2 / fp16mx / fp16x3 on a few utterances and
models, solo == batched, and bit-identity of the stream-K and per-tile kernels.  usage: check_mx.py [topology]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402

P = H.pkg()
topo = sys.argv[1] if len(sys.argv) > 1 else "v2_xvector"
for seed in (123, 7):
    net, line = H.synth_model(topo, seed)
    model = P.Model(raw=net.to_bytes(True), nnet_config=line)
    n2 = H.nm.Nnet3.from_bytes(net.to_bytes(True))
    n2.apply_nnet_config(line)
    ev = H.xo.GraphEvaluator(n2, np.float64)
    utts = [H.features(i, T) for i, T in enumerate((400, 400, 314, 400, 1000, 400, 400, 333))]
    ref = np.stack([ev.compute(u)[0] for u in utts])
    feats, offs = H.pack(utts)
    outs = {}
    for name in ("fp16x2", "fp16mx", "fp16x3"):
        ctx = P.Context(model, device=0, precision=P.PRECISIONS[name])
        out = ctx.forward_batch(feats, offs)
        outs[name] = out
        errs = [H.rel_err(out[i:i + 1], ref[i:i + 1]) for i in range(len(utts))]
        print("seed %d %-7s rel err: max %.2e mean %.2e" % (seed, name, max(errs), float(np.mean(errs))))
        if name == "fp16mx":
            solo = ctx.forward_batch(utts[2], np.array([0, len(utts[2])], np.int32))
            print("   solo == batched:", bool(np.array_equal(solo[0], out[2])))
            # a batch large enough for the persistent grid: 264 chunks of 400 frames
            big = [utts[0]] * 3 + [utts[i % len(utts)] for i in range(261)]
            fb, ob = H.pack(big)
            outb = ctx.forward_batch(fb, ob)
            print("   stream-K (big batch) == per-tile kernel (small batch):", bool(np.array_equal(outb[0], out[0])),
                  " max |diff| %.3g" % float(np.abs(outb[0] - out[0]).max()))
        del ctx
