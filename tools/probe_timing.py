"""GPU, experiment build only (XVEC_LIB=build/libxvec_TIMING.so): segment durations of the stream-K K loop of tdnn2
(s_memtime stamps of waves 0 and 4 of workgroup 0)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
P = H.pkg()
net, line = H.synth_model("v2_xvector")
model = P.Model(raw=net.to_bytes(True), nnet_config=line)
ctx = P.Context(model, device=0, precision=P.PRECISIONS[sys.argv[1] if len(sys.argv) > 1 else "fp16mx"])
utts = [H.features(i % 7, 400) for i in range(256)]
fb, ob = H.pack(utts)
for _ in range(3):
    ctx.forward_batch(fb, ob)
buf = (ctypes.c_ulonglong * 4096)()
n = P.lib().xv_probe_dump(buf, 4096)
ev = [(buf[i], buf[i + 1] >> 56, buf[i + 1] & ((1 << 56) - 1)) for i in range(0, n, 2)]
names = {1: "loop top", 2: "reads issued", 3: "dma issued", 4: "lgkm wait done", 5: "vm wait + barrier done", 6: "compute issued", 7: "barrier done"}
for w in (0, 1):
    e = [(t, c) for (ww, t, c) in ev if ww == w]
    print("wave group", w, "stamps", len(e))
    # durations between consecutive stamps, averaged per (from, to) tag pair over the recorded steps
    agg = {}
    for (t0, c0), (t1, c1) in zip(e[:-1], e[1:]):
        agg.setdefault((t0, t1), []).append(c1 - c0)
    for (t0, t1), v in sorted(agg.items()):
        v = np.array(v[2:] if len(v) > 4 else v, dtype=np.float64)
        print("   %-22s -> %-22s  n=%3d  mean %7.0f  min %6.0f  max %6.0f" % (names.get(t0, t0), names.get(t1, t1), len(v), v.mean(), v.min(), v.max()))
