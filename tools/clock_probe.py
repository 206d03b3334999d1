"""GPU box: the shader clock MI355X holds while the dominant kernel runs, measured from INSIDE the kernel (VERDICT r04 item 4:
"measure the clock for real").  Needs the measurement build of the library:

    make -C speaker-embedding-with-phonetic-information_amd/csrc OUT=../../.ab/clk OBJ=build_clk BIN=../../.ab/clk/bin \
        CXXFLAGS="-O3 -std=c++17 -fPIC -DXVEC_CLOCK_PROBE" ../../.ab/clk/libxvec_hip.so
    XVEC_LIB=.ab/clk/libxvec_hip.so python3 tools/clock_probe.py [seconds] [precision] [--zeros]

Every workgroup of tdnn_gemm_kernel_p8 stamps s_memtime (shader cycles) and s_memrealtime (constant 100 MHz) at its first and
last instruction; clock = cycles / realtime ticks x 100 MHz.  The bench workload (v2 x-vector, 256 x 400) runs back to back for
`seconds` first, so the chip is in its steady thermal / power state; the stamps read are those of the last step."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import helpers as H  # noqa: E402

P = H.pkg()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
seconds = float(args[0]) if args else 3.0
prec = args[1] if len(args) > 1 else "auto"      # auto = fp16mx on every chunk of this workload
zeros = "--zeros" in sys.argv
net, line = H.synth_model("v2_xvector")
model = P.Model(raw=net.to_bytes(True), nnet_config=line)
os.environ["XVEC_LANES"] = "1"
ctx = P.Context(model, device=0, precision=P.PRECISIONS[prec])
B, T = 256, 400
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(20180101)
feats = torch.randn(B * T, 23, generator=g, device=dev) * (8.0 * 0.9 ** torch.arange(23, device=dev))
if zeros:
    feats.zero_()
offs = (np.arange(B + 1) * T).astype(np.int32)
out = torch.empty(B, 512, device=dev)
L = P.lib()
if not hasattr(L, "xvec_clock_probe_read"):
    raise SystemExit("this libxvec_hip.so was not built with -DXVEC_CLOCK_PROBE (set XVEC_LIB to the measurement build)")
buf = (ctypes.c_ulonglong * (3 * 512 * 2))()
t0 = time.time()
n = 0
while time.time() - t0 < seconds:
    for _ in range(50):
        ctx.forward_batch_device(feats.data_ptr(), offs, out.data_ptr(), 512, None)
    ctx.synchronize()
    n += 50
dt = time.time() - t0
assert L.xvec_clock_probe_read(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(3, 512, 2).astype(np.float64)
print("%s%s: %d steps in %.2f s = %.0f utt/s" % (prec, " (all-zero features)" if zeros else "", n, dt, n * B / dt))
for slot, name in enumerate(("act epilogue, <= 8 K tiles (tdnn4)", "act epilogue, 24 K tiles (tdnn2 / tdnn3)", "statistics epilogue (tdnn5)")):
    cyc, rt = a[slot, :, 0], a[slot, :, 1]
    ok = rt > 0
    if not ok.any():
        continue
    ghz = cyc[ok] / rt[ok] * 0.1
    print("  %-42s %3d workgroups: clock %.3f GHz (min %.3f, max %.3f); workgroup lifetime %.1f us = %.0f k cycles"
          % (name, int(ok.sum()), ghz.mean(), ghz.min(), ghz.max(), rt[ok].mean() / 100.0, cyc[ok].mean() / 1e3))

if hasattr(L, "xvec_exchange_probe_read"):
    xb = (ctypes.c_ulonglong * (3 * 512 * 4))()
    assert L.xvec_exchange_probe_read(xb) == 0
    x = np.frombuffer(xb, dtype=np.uint64).reshape(3, 512, 4).astype(np.float64)
    print("stream-K exchange and epilogue, wave 0 of each workgroup, shader cycles (us at the clock above):")
    for slot, name in enumerate(("tdnn4", "tdnn2 / tdnn3", "tdnn5 (statistics)")):
        ghz = (a[slot, :, 0] / np.maximum(a[slot, :, 1], 1) * 0.1)
        ghz = float(ghz[a[slot, :, 1] > 0].mean()) if (a[slot, :, 1] > 0).any() else 2.0
        parts = []
        for j, what in enumerate(("tail part: store of the partial tile until it is out", "head part: flag wait + loads issued",
                                  "head part: drain until the partial tile is in", "epilogue of a whole tile")):
            v = x[slot, :, j]
            v = v[v > 0]
            if v.size:
                parts.append("%s %.0f k = %.1f us (%d wgs)" % (what, v.mean() / 1e3, v.mean() / ghz / 1e3, v.size))
        print("  %-20s %s" % (name, "; ".join(parts)))

if hasattr(L, "xvec_part_probe_read"):
    pb = (ctypes.c_ulonglong * (3 * 512 * 2))()
    assert L.xvec_part_probe_read(pb) == 0
    pp = np.frombuffer(pb, dtype=np.uint64).reshape(3, 512, 2).astype(np.float64)
    print("wait + barrier that opens a whole-tile part behind an epilogue (DMA of its first tiles landing + the epilogue's stores draining), wave 0:")
    for slot, name in enumerate(("tdnn4", "tdnn2 / tdnn3", "tdnn5 (statistics)")):
        ghz = (a[slot, :, 0] / np.maximum(a[slot, :, 1], 1) * 0.1)
        ghz = float(ghz[a[slot, :, 1] > 0].mean()) if (a[slot, :, 1] > 0).any() else 2.0
        ok = pp[slot, :, 1] > 0
        if ok.any():
            per = pp[slot, ok, 0] / pp[slot, ok, 1]
            print("  %-20s %.0f cycles = %.2f us per part (%.1f such parts per workgroup; %.1f us per workgroup and launch)"
                  % (name, per.mean(), per.mean() / ghz / 1e3, pp[slot, ok, 1].mean(), pp[slot, ok, 0].mean() / ghz / 1e3))
