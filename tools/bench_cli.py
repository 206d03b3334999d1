#!/usr/bin/env python3
"""End-to-end rate of the drop-in command line (file in, file out; PCIe and Kaldi I/O included): writes a synthetic
binary feature archive, runs bin/nnet3-xvector-compute on it exactly as extract_xvectors_new.sh:92-93 would, and
reports utterances/s from the wall clock and from the tool's own "Time taken" line."""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from oracle import kaldi_io as kio  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    targ = sys.argv[2] if len(sys.argv) > 2 else "400"   # frames per utterance: "400", or "200-600" = uniform over the range
    if "-" in targ:
        lo, hi = (int(v) for v in targ.split("-"))
        lens = np.random.default_rng(5).integers(lo, hi + 1, 64)
    else:
        lens = np.full(64, int(targ))
    T = float(lens.mean())
    topo = ([a.split("=", 1)[1] for a in sys.argv[3:] if a.startswith("--topology=")] or ["v2_xvector"])[0]   # helpers.TOPOLOGIES
    extra = [a for a in sys.argv[3:] if a.startswith("--") and a not in ("--pipe", "--compressed", "--recipe") and not a.startswith("--topology=")]
    compressed = "--compressed" in sys.argv[3:]   # Kaldi "CM" objects (what make_mfcc.sh stores); with --cmn-window=300 the job runs the device front-end
    pipe = "--pipe" in sys.argv[3:]   # the recipes' form: the features arrive through a pipe (extract_xvectors_new.sh:79)
    recipe = "--recipe" in sys.argv[3:]   # the scripts' own feature rspecifier (extract_xvectors_new.sh:79): recognised, run on the device
    wspec = ([a for a in sys.argv[3:] if not a.startswith("--")] or [None])[0]
    d = tempfile.mkdtemp(prefix="xvcli", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    net, line = H.synth_model(topo)
    open(os.path.join(d, "final.raw"), "wb").write(net.to_bytes(True))
    pool = [H.features(1000 + i, int(lens[i])) for i in range(64)]
    import io
    blobs = []
    for m in pool:
        b = io.BytesIO()
        if compressed:
            kio.write_compressed_matrix(b, m, "CM")
        else:
            kio.write_matrix(b, m)
        blobs.append(b.getvalue())
    scp_lines = []
    with open(os.path.join(d, "feats.ark"), "wb") as f:
        for i in range(n):
            f.write(("utt%07d " % i).encode())
            scp_lines.append("utt%07d %s/feats.ark:%d\n" % (i, d, f.tell()))
            f.write(b"\0B" + blobs[i % 64])
    feat_spec = ("ark:cat %s/feats.ark |" if pipe else "ark:%s/feats.ark") % d
    if recipe:
        open(os.path.join(d, "feats.scp"), "w").writelines(scp_lines)
        vpool = []
        for i in range(64):
            b = io.BytesIO()
            v = np.ones(int(lens[i]), np.float32)
            v[::7] = 0.0                          # a seventh of the frames is not voiced
            kio.write_vector(b, v)
            vpool.append(b.getvalue())
        with open(os.path.join(d, "vad.ark"), "wb") as f, open(os.path.join(d, "vad.scp"), "w") as g:
            for i in range(n):
                f.write(("utt%07d " % i).encode())
                g.write("utt%07d %s/vad.ark:%d\n" % (i, d, f.tell()))
                f.write(b"\0B" + vpool[i % 64])
        feat_spec = ("ark:apply-cmvn-sliding --norm-vars=false --center=true --cmn-window=300 scp:%s/feats.scp ark:- | "
                     "select-voiced-frames ark:- scp,s,cs:%s/vad.scp ark:- |" % (d, d))
    binp = os.path.join(ROOT, H.PKG_NAME, "bin", "nnet3-xvector-compute")
    cmd = [binp, "--use-gpu=yes", "--min-chunk-size=25", "--chunk-size=10000", "--output-node=" + line.split("input=")[1]] + extra + [
        os.path.join(d, "final.raw"), feat_spec, wspec or "ark,scp:%s/x.ark,%s/x.scp" % (d, d)]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE)
    wall = time.perf_counter() - t0
    err = r.stderr.decode()
    m = re.search(r"Time taken ([0-9.e+-]+)s", err)
    loop = float(m.group(1)) if m else None
    print(json.dumps({"input": "pipe" if pipe else "file", "utts": n, "frames": T, "rc": r.returncode, "wall_s": wall, "wall_utt_per_s": n / wall,
                      "loop_s": loop, "loop_utt_per_s": n / loop if loop else None,
                      "feature_GB": n * T * 23 * 4 / 1e9, "frames_per_s": n * T / loop if loop else None, "topology": topo, "tail": [l for l in err.strip().splitlines() if "stages" in l or "calibration" in l or "Done" in l or "WaitHost" in l or "host cost" in l or "CPU seconds" in l or "front-end:" in l or "recognised" in l]}))
    for fn in os.listdir(d):
        os.remove(os.path.join(d, fn))
    os.rmdir(d)


if __name__ == "__main__":
    main()
