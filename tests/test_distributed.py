"""N > 1 path on CPU: world_size-2 gloo run of the multi-GPU launcher in --dry-run mode (sharding + the single
weight broadcast; no device), plus the sharding rule against utils/split_scp.pl:208-217 semantics.  The GPU
flavour (nccl = RCCL) runs the same code with --backend nccl."""
import importlib
import os
import re
import socket
import subprocess
import sys

import pytest

import helpers as H

D = importlib.import_module(H.PKG_NAME + ".dist_extract")


def test_shard_bounds_matches_split_scp_rule():
    # split_scp.pl: first (n % nj) jobs get floor(n/nj)+1 lines, the rest floor(n/nj); contiguous, in order
    for n in (0, 1, 7, 8, 9, 100, 1000003):
        for w in (1, 2, 3, 8, 32):
            b = D.shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert sizes == [n // w + (1 if r < n % w else 0) for r in range(w)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_world_size_2_gloo_dry_run(tmp_path):
    net, line = H.synth_model("v2_xvector")
    (tmp_path / "final.raw").write_bytes(net.to_bytes(True))
    keys = ["utt%04d" % i for i in range(11)]
    (tmp_path / "feats.scp").write_text("".join("%s /data/feats.ark:%d\n" % (k, 100 * i) for i, k in enumerate(keys)))
    out = tmp_path / "out"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(H.ROOT, H.PKG_NAME, "dist_extract.py"),
           "--nnet", str(tmp_path / "final.raw"), "--output-node", "tdnn6.affine",
           "--feats-scp", str(tmp_path / "feats.scp"), "--out-dir", str(out), "--name", "t",
           "--backend", "gloo", "--dry-run"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout
    ranks = dict((int(m.group(1)), m.groups()) for m in
                 re.finditer(r"rank (\d)/2: blob (\d+) bytes sha1 (\w+), utterances \[(\d+), (\d+)\)", r.stdout))
    assert set(ranks) == {0, 1}, r.stdout
    # both ranks hold the identical packed weights after the ONE broadcast, and they are the image rank 0 packed
    P = H.pkg()
    import hashlib
    want = hashlib.sha1(P.Model(raw=net.to_bytes(True), nnet_config=line).pack()).hexdigest()
    assert ranks[0][2] == ranks[1][2] == want
    # utterance shards: contiguous, disjoint, complete, 6 + 5
    assert (ranks[0][3], ranks[0][4], ranks[1][3], ranks[1][4]) == ("0", "6", "6", "11")
    merged = [l.split()[0] for l in open(out / "xvector_t.scp")]
    assert merged == keys
    assert [l.split()[0] for l in open(out / "feats_t.1.scp")] == keys[:6]
    assert [l.split()[0] for l in open(out / "feats_t.2.scp")] == keys[6:]


def test_rank0_model_failure_reaches_every_rank(tmp_path):
    """A model rank 0 cannot load must end the whole job quickly and non-zero (no rank left waiting in the broadcast)."""
    (tmp_path / "final.raw").write_bytes(b"this is not an nnet3 model")
    (tmp_path / "feats.scp").write_text("utt0 /data/feats.ark:10\n")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(H.ROOT, H.PKG_NAME, "dist_extract.py"),
           "--nnet", str(tmp_path / "final.raw"), "--feats-scp", str(tmp_path / "feats.scp"), "--out-dir", str(tmp_path / "o"),
           "--backend", "gloo", "--dry-run"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120,
                       env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode != 0
    assert "rank 0 could not load the model" in r.stdout


def test_broadcast_watchdog_turns_a_stalled_collective_into_an_exit_status():
    """the start-up broadcast of dist_extract.py / bench.py is bracketed by a watchdog - a rank that waits
    longer than XVEC_BCAST_TIMEOUT for the others exits with status 3 and says what it waited for (a plain exit of the process,
    nothing is re-executed); a collective that completes in time cancels it."""
    code = ("import importlib, sys, time; sys.path.insert(0, %r); P = importlib.import_module(%r)\n"
            "with P.Watchdog('a quick collective'):\n    pass\n"
            "time.sleep(0.5)\n"
            "print('survived the cancelled one', flush=True)\n"
            "with P.Watchdog('the broadcast of the packed weights (2 ranks, backend gloo)'):\n    time.sleep(30)\n"
            "print('not reached')\n") % (H.ROOT, H.PKG_NAME)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60,
                       env=dict(os.environ, XVEC_BCAST_TIMEOUT="0.3", RANK="1"))
    assert r.returncode == 3, (r.returncode, r.stdout, r.stderr)
    assert "survived the cancelled one" in r.stdout and "not reached" not in r.stdout
    assert "rank 1: the broadcast of the packed weights (2 ranks, backend gloo) did not complete within" in r.stderr, r.stderr


def _split_cases():
    import json
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "split_scp", "cases.json")))


def test_sharding_rules_against_the_reference_split_scp_goldens():
    """tests/golden/split_scp/cases.json holds what the reference's utils/split_scp.pl did (run in the build container by
    tests/golden/make_split_scp_goldens.py) with seeded lists: plain (:193-221 -> shard_bounds) and --utt2spk (:84-191 ->
    shard_by_speaker, the split utils/data/split_data.sh makes by default and extract_xvectors_new.sh:72 uses), including
    speakers that are not contiguous in the list, as many jobs as speakers, and more jobs than speakers (an error in both)."""
    n_cases = n_err = 0
    for c in _split_cases():
        if "counts" in c:
            utts = ["spk%03d-utt%03d" % (s, k) for s, n in enumerate(c["counts"]) for k in range(n)]
            spks = ["spk%03d" % s for s, n in enumerate(c["counts"]) for k in range(n)]
        else:
            utts, spks = c["utts"], c["spks"]
        lines = ["%s /feats/%s.ark:%d\n" % (u, u, 17 * i) for i, u in enumerate(utts)]
        n_cases += 1
        if c["by_speaker"]:
            if c.get("error"):
                n_err += 1
                with pytest.raises(ValueError):
                    D.shard_by_speaker(lines, dict(zip(utts, spks)), c["nj"])
                continue
            got = D.shard_by_speaker(lines, dict(zip(utts, spks)), c["nj"])
            assert [[lines.index(l) for l in sh] for sh in got] == c["shards"], (c["nj"], c.get("counts"))
            # no speaker in two jobs
            owner = {}
            for j, sh in enumerate(got):
                for l in sh:
                    assert owner.setdefault(spks[lines.index(l)], j) == j
        elif not c.get("error"):
            b = D.shard_bounds(len(lines), c["nj"])
            assert [list(range(lo, hi)) for lo, hi in b] == c["shards"], (c["nj"], len(lines))
    assert n_cases >= 19 and n_err >= 1


def test_shard_by_speaker_rejects_an_utterance_without_a_speaker():
    with pytest.raises(ValueError, match="No such utterance"):
        D.shard_by_speaker(["a x\n", "b y\n"], {"a": "s1"}, 1)


def test_world_size_2_gloo_dry_run_sharded_by_speaker(tmp_path):
    """--utt2spk: the two ranks take the lists split_scp.pl --utt2spk would give jobs 1 and 2 (no speaker cut in two), and the
    merged output list is the input list again (speakers contiguous, as in the reference's sorted data directories)."""
    net, line = H.synth_model("v2_xvector")
    (tmp_path / "final.raw").write_bytes(net.to_bytes(True))
    counts = [4, 1, 3, 2, 5]                      # 15 utterances: plain halves would be 8 + 7 and cut speaker 2
    keys = ["spk%d-utt%d" % (s, k) for s, n in enumerate(counts) for k in range(n)]
    (tmp_path / "feats.scp").write_text("".join("%s /data/feats.ark:%d\n" % (k, 100 * i) for i, k in enumerate(keys)))
    (tmp_path / "utt2spk").write_text("".join("%s %s\n" % (k, k.split("-")[0]) for k in keys))
    out = tmp_path / "out"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(H.ROOT, H.PKG_NAME, "dist_extract.py"),
           "--nnet", str(tmp_path / "final.raw"), "--output-node", "tdnn6.affine", "--utt2spk", str(tmp_path / "utt2spk"),
           "--feats-scp", str(tmp_path / "feats.scp"), "--out-dir", str(out), "--name", "t",
           "--backend", "gloo", "--dry-run"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout
    lines = ["%s x\n" % k for k in keys]
    want = D.shard_by_speaker(lines, {k: k.split("-")[0] for k in keys}, 2)
    one = [l.split()[0] for l in open(out / "feats_t.1.scp")]
    two = [l.split()[0] for l in open(out / "feats_t.2.scp")]
    assert one == [l.split()[0] for l in want[0]] and two == [l.split()[0] for l in want[1]]
    assert {k.split("-")[0] for k in one}.isdisjoint({k.split("-")[0] for k in two})
    assert (len(one), len(two)) == (8, 7) or abs(len(one) - len(two)) <= 3
    assert [l.split()[0] for l in open(out / "xvector_t.scp")] == keys
