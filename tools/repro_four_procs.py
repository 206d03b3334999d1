"""GPU box: the co-tenancy case of tests/test_gpu_cli.py::test_four_concurrent_processes_share_one_gpu, looped, with the
evidence kept: for every process whose archive differs from the solo run - its calibration log line, how many utterances
differ and where the first / last ones sit (all of them = another arithmetic was chosen; a few = an extraction race).

usage: repro_four_procs.py [--topology v5_cvector] [--precision default] [--utts 6000] [--iters 5] [--procs 4]
                           [--noise SECONDS] [--env K=V ...] [--solo-repeats N]
The reference's launch mode: extract_xvectors_new.sh:83-93 (run.pl JOB=1:nj, nj / #GPUs processes per GPU)."""
import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from oracle import kaldi_io as kio  # noqa: E402

BIN = os.path.join(ROOT, H.PKG_NAME, "bin")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--topology", default="v5_cvector")
    ap.add_argument("--precision", default="default")
    ap.add_argument("--utts", type=int, default=6000)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--procs", type=int, default=4)
    ap.add_argument("--noise", type=float, default=0.0)
    ap.add_argument("--solo-repeats", type=int, default=1)
    ap.add_argument("--env", nargs="*", default=[])
    ap.add_argument("--mode", default="fixed", choices=["fixed", "shared", "self"],
                    help="fixed: the arithmetic --precision names (default = plain fp16mx2); shared: --calibration=<file>, removed before "
                         "every iteration (all processes measure at once, one publishes, all adopt; the solo reference is re-run after "
                         "the first iteration from the published file); self: --calibrate=true (round 5's default: every process "
                         "measures for itself)")
    a = ap.parse_args()
    env = dict(os.environ)
    for kv in a.env:
        k, v = kv.split("=", 1)
        env[k] = v
    net, line = H.synth_model(a.topology)
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    d = tempfile.mkdtemp(prefix="xvrepro", dir=shm)
    bad = 0
    noise = None
    try:
        open(os.path.join(d, "final.raw"), "wb").write(net.to_bytes(True))
        open(os.path.join(d, "extract.config"), "w").write(line + "\n")
        pool = [H.features(3000 + i, 400) for i in range(32)]
        with open(os.path.join(d, "feats.ark"), "wb") as f:
            for i in range(a.utts):
                f.write(("utt%06d " % i).encode() + b"\0B")
                kio.write_matrix(f, pool[(i * 7) % 32])

        calib_path = os.path.join(d, "xvec.calib")
        extra = {"fixed": [], "shared": ["--calibration=" + calib_path], "self": ["--calibrate=true"]}[a.mode]

        def cmd(job):
            return [os.path.join(BIN, "nnet3-xvector-compute"), "--use-gpu=no", "--min-chunk-size=25", "--chunk-size=10000",
                    "--precision=" + a.precision, "--batch-frames=40000"] + extra + [
                    "%s --nnet-config=%s/extract.config %s/final.raw - |" % (os.path.join(BIN, "nnet3-copy"), d, d),
                    "ark:%s/feats.ark" % d, "ark:%s/xvector.%s.ark" % (d, job)]

        def calib(err):
            return [ln for ln in err.splitlines() if "calibration" in ln or "arithmetic " in ln or "ERROR" in ln]

        def read(job):
            return open(os.path.join(d, "xvector.%s.ark" % job), "rb").read()

        def diff_report(tag, got, ref, err):
            g = np.frombuffer(got, np.uint8)
            r = np.frombuffer(ref, np.uint8)
            if g.size != r.size:
                print("  %s: archive sizes differ (%d vs %d)" % (tag, g.size, r.size))
                return
            rec = g.size // a.utts
            gd = g.reshape(a.utts, rec)
            rd = r.reshape(a.utts, rec)
            rows = np.nonzero((gd != rd).any(axis=1))[0]
            hdr = 20                              # "utt%06d " + "\0B" + "FV " + "\4" + int32 dim
            assert rec == hdr + 4 * int.from_bytes(got[16:20], "little"), rec
            gv = gd[:, hdr:].copy().view(np.float32)
            rv = rd[:, hdr:].copy().view(np.float32)
            rel = np.abs(gv - rv).max(axis=1) / np.maximum(np.abs(rv).max(axis=1), 1e-30)
            print("  %s: %d of %d utterances differ (first %s, last %s); distinct batches touched %d; max rel diff %.3g, median over differing %.3g"
                  % (tag, rows.size, a.utts, rows[:5].tolist(), rows[-3:].tolist(), len(set((rows // 100).tolist())),
                     rel.max(), float(np.median(rel[rows])) if rows.size else 0.0))
            for ln in calib(err):
                print("     " + ln[-600:])

        t0 = time.perf_counter()
        r = subprocess.run(cmd("solo"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        solo = read("solo")
        solo_err = r.stderr.decode()
        print("solo: %.1f s" % (time.perf_counter() - t0))
        for ln in calib(solo_err):
            print("   " + ln[-600:])
        for k in range(1, a.solo_repeats):
            r = subprocess.run(cmd("solo%d" % k), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            if read("solo%d" % k) != solo:
                bad += 1
                diff_report("solo repeat %d" % k, read("solo%d" % k), solo, r.stderr.decode())
        if a.noise > 0:
            noise = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", os.environ.get("NOISE_TOOL", "noise_gpu.py")), str(a.noise)])
            time.sleep(8.0)
        shared_ref = {}
        for it in range(a.iters):
            t0 = time.perf_counter()
            if a.mode == "shared" and os.path.exists(calib_path):
                shared_ref[open(calib_path).read().split("note")[0]] = None
                os.remove(calib_path)
            procs = [subprocess.Popen(cmd(str(j)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env) for j in range(a.procs)]
            errs = [p.communicate()[1].decode() for p in procs]
            nb = 0
            for j, (p, e) in enumerate(zip(procs, errs)):
                if p.returncode != 0:
                    print("  iter %d proc %d: exit %d: %s" % (it, j, p.returncode, e[-500:]))
                    nb += 1
                    continue
                got = read(str(j))
                if a.mode == "shared":
                    # whoever published: all processes of an iteration agree; and the same choice gives the same bytes every time
                    key = open(calib_path).read().split("note")[0]
                    if shared_ref.get(key) is None:
                        shared_ref[key] = got
                    if got != shared_ref[key]:
                        nb += 1
                        diff_report("iter %d proc %d" % (it, j), got, shared_ref[key], e)
                    continue
                if got != solo:
                    nb += 1
                    diff_report("iter %d proc %d" % (it, j), got, solo, e)
            bad += nb
            print("iter %d: %d of %d processes differ from solo (%.1f s)" % (it, nb, a.procs, time.perf_counter() - t0), flush=True)
    finally:
        if noise:
            noise.terminate()
            noise.wait()
        for fn in os.listdir(d):
            os.remove(os.path.join(d, fn))
        os.rmdir(d)
    print("TOTAL differing process runs: %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
