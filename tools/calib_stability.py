"""GPU box: is a fresh context's calibration bit-stable under co-tenancy?  Every repetition builds a NEW context (the state a
run.pl job starts in), calibrates on the 64-chunk sample the command line would draw from the co-tenancy test's table, and
records the raw bytes of xv_calibration plus a hash of a 100-chunk batch in each of fp16x3 / fp16mx / fp16mx2 / the chosen
arithmetic.  With --procs N the same loop runs in N processes at once (optionally beside tools/noise_gpu.py); every record of
every process must be identical.

usage: calib_stability.py [--topology v5_cvector] [--reps 20] [--procs 4] [--noise SECONDS]
       (worker mode: --worker OUTFILE)"""
import argparse
import ctypes
import hashlib
import importlib
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402


def worker(a):
    pkg = importlib.import_module(H.PKG_NAME)
    net, line = H.synth_model(a.topology)
    model = pkg.Model(raw=net.to_bytes(True), nnet_config=line)
    pool = [H.features(3000 + i, 400) for i in range(32)]
    n_utts, want = 6000, 64
    sample = [pool[((((2 * i + 1) * n_utts) // (2 * want)) * 7) % 32] for i in range(want)]   # SampleTable's centres
    sf, so = H.pack(sample)
    batch = [pool[(i * 7) % 32] for i in range(100)]
    bf, bo = H.pack(batch)
    L = pkg.lib()
    with open(a.worker, "w") as out:
        for r in range(a.reps):
            ctx = pkg.Context(model, device=0)
            c = pkg.Calibration()
            st = L.xv_ctx_calibrate(ctx._h, sf.ctypes.data, so.ctypes.data, len(so) - 1, ctypes.c_float(7.5e-5), ctypes.byref(c))
            assert st == 0, L.xv_last_error()
            rec = [bytes(c).hex()]
            rec.append(hashlib.sha1(ctx.forward_batch(bf, bo).tobytes()).hexdigest()[:12])
            mask = ctx.lite_mask
            for m in ("fp16x3", "fp16mx", "fp16mx2"):
                ctx.set_fast_mode(m)
                rec.append(hashlib.sha1(ctx.forward_batch(bf, bo).tobytes()).hexdigest()[:12])
            out.write(" ".join(rec) + " mask=%x tail=%r hold=%r\n" % (mask, c.tail, c.err_holdout))
            out.flush()
            ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--topology", default="v5_cvector")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--procs", type=int, default=4)
    ap.add_argument("--noise", type=float, default=0.0)
    ap.add_argument("--worker", default=None)
    a = ap.parse_args()
    if a.worker:
        return worker(a)
    import tempfile
    d = tempfile.mkdtemp(prefix="xvcal")
    noise = None
    if a.noise > 0:
        noise = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", os.environ.get("NOISE_TOOL", "noise_gpu.py")), str(a.noise)])
        time.sleep(8.0)
    t0 = time.perf_counter()
    outs = [os.path.join(d, "w%d.txt" % j) for j in range(a.procs)]
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--topology", a.topology, "--reps", str(a.reps),
                               "--worker", o]) for o in outs]
    rcs = [p.wait() for p in procs]
    if noise:
        noise.terminate()
        noise.wait()
    recs = {}
    for j, o in enumerate(outs):
        for r, ln in enumerate(open(o).read().splitlines()):
            recs.setdefault(ln, []).append((j, r))
    print("%d processes x %d fresh contexts in %.1f s (exit codes %s): %d distinct record(s)"
          % (a.procs, a.reps, time.perf_counter() - t0, rcs, len(recs)))
    for ln, who in sorted(recs.items(), key=lambda kv: -len(kv[1])):
        print("  x%-4d %s   first at (proc, rep) %s" % (len(who), ln, who[:4]))
    return 0 if len(recs) == 1 and not any(rcs) else 1


if __name__ == "__main__":
    sys.exit(main())
