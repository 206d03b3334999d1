#!/usr/bin/env python3
"""BASELINE.json config 4 end to end: the 1 M-utterance synthetic set (a pool of 4096 distinct 400-frame matrices cycled under
unique keys) through dist_extract.py on N ranks of one node - file in, file out, PCIe and Kaldi I/O
included, the weights broadcast once over RCCL (utterance sharding: utils/split_scp.pl:208-217; per-rank outputs concatenated as
egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:99 does).  Reports the whole job's rate and, per rank, what its table loop
waited for (XVEC_TIMING: reader wait / pack / submit / finish), so that the host-ingest ceiling at N ranks is a measured number.

  python tools/bench_dist.py [utterances=1000000] [ranks=1] [--backend nccl|gloo] [--force-device D] [--precision P]

The pool lives in one archive under /dev/shm (151 MB); the script file holds one `key path:offset` line per utterance."""
import json
import os
import re
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from oracle import kaldi_io as kio  # noqa: E402

POOL = 4096


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(pos[0]) if pos else 1000000
    ranks = int(pos[1]) if len(pos) > 1 else 1
    opts = {}
    av = sys.argv[1:]
    for i, a in enumerate(av):
        if a.startswith("--") and i + 1 < len(av):
            opts[a[2:]] = av[i + 1]
    d = tempfile.mkdtemp(prefix="xvdist", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    net, line = H.synth_model("v2_xvector")
    open(os.path.join(d, "final.raw"), "wb").write(net.to_bytes(True))
    ark = os.path.join(d, "pool.ark")
    offs = []
    with open(ark, "wb") as f:
        for i in range(min(POOL, n)):
            f.write(("p%04d " % i).encode())
            offs.append(f.tell())
            f.write(b"\0B")
            kio.write_matrix(f, H.features(20180101 + i, 400))
    with open(os.path.join(d, "feats.scp"), "w") as f:
        for i in range(n):
            f.write("utt%07d %s:%d\n" % (i, ark, offs[i % len(offs)]))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, H.PKG_NAME, "dist_extract.py"), "--nnet", os.path.join(d, "final.raw"),
           "--output-node", "tdnn6.affine", "--feats-scp", os.path.join(d, "feats.scp"), "--out-dir", os.path.join(d, "out"),
           "--name", "x", "--min-chunk-size", "25", "--chunk-size", "10000", "--backend", opts.get("backend", "nccl"),
           "--precision", opts.get("precision", "default")]
    if "force-device" in opts:
        cmd += ["--force-device", opts["force-device"]]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       env=dict(os.environ, XVEC_TIMING="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1"))
    wall = time.perf_counter() - t0
    stages = [m.groupdict() for m in re.finditer(
        r"consumer stages: wait for reader (?P<wait>[0-9.e+-]+) s, pack (?P<pack>[0-9.e+-]+) s, plan\+submit (?P<submit>[0-9.e+-]+) s, "
        r"finish\+write (?P<finish>[0-9.e+-]+) s", r.stdout)]
    stages = [{k: float(v) for k, v in s_.items()} for s_ in stages]
    loops = [sum(s_.values()) for s_ in stages]
    cpu = [{"rank": int(m.group(1)), "user_s": float(m.group(2)), "system_s": float(m.group(3)), "cpu_us_per_utt": float(m.group(5))}
           for m in re.finditer(r"rank (\d+) host cpu: ([0-9.]+) s user \+ ([0-9.]+) s system for (\d+) utterances = ([0-9.]+) us", r.stdout)]
    done = re.search(r"Done (\d+) utterances, failed for (\d+)", r.stdout)
    res = {"utterances": n, "ranks": ranks, "backend": opts.get("backend", "nccl"), "rc": r.returncode, "wall_s": wall,
           "wall_utt_per_s": n / wall, "done": int(done.group(1)) if done else None,
           "per_rank_loop_s": loops, "loop_utt_per_s": (n / max(loops)) if loops else None,
           "per_rank_stages_s": stages,
           # CPU seconds each rank's process spent (import of torch and model load included): cores needed at R ranks =
           # sum over ranks of cpu_us_per_utt x that rank's utt/s
           "per_rank_host_cpu": cpu,
           "reader_wait_frac": [s_["wait"] / max(1e-9, sum(s_.values())) for s_ in stages],
           "calibration": [l for l in r.stdout.splitlines() if "calibration" in l][:3]}
    print(json.dumps(res))
    if r.returncode:
        print(r.stdout[-3000:], file=sys.stderr)
    subprocess.run(["rm", "-rf", d])


if __name__ == "__main__":
    main()
