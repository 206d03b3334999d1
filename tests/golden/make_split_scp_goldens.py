#!/usr/bin/env python3
"""Generates tests/golden/split_scp/cases.json: what the reference's own `utils/split_scp.pl` does with a few seeded utterance
lists - plain (`:193-221`, the rule of dist_extract.shard_bounds) and with `--utt2spk` (`:84-191`, what `utils/data/split_data.sh`
calls by default and extract_xvectors_new.sh:72 takes its per-job lists from; dist_extract.shard_by_speaker).

Runs the reference's Perl script where it lies under /root/reference (this container only; nothing of it is copied): the
fixture holds inputs and outputs only - utterance ids, speaker ids, the job each line went to.

    python3 tests/golden/make_split_scp_goldens.py"""
import json
import os
import subprocess
import tempfile

import numpy as np

REF = "/root/reference/egs/sre/v2/utils/split_scp.pl"
HERE = os.path.dirname(os.path.abspath(__file__))


def run(utts, spks, nj, by_speaker):
    d = tempfile.mkdtemp()
    scp, u2s = os.path.join(d, "in.scp"), os.path.join(d, "utt2spk")
    with open(scp, "w") as f:
        f.writelines("%s /feats/%s.ark:%d\n" % (u, u, 17 * i) for i, u in enumerate(utts))
    with open(u2s, "w") as f:
        f.writelines("%s %s\n" % (u, s) for u, s in zip(utts, spks))
    outs = [os.path.join(d, "out.%d.scp" % j) for j in range(1, nj + 1)]
    cmd = ["perl", REF] + (["--utt2spk=" + u2s] if by_speaker else []) + [scp] + outs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        return {"error": True}
    return {"shards": [[l.split()[0] for l in open(o)] for o in outs]}


def main():
    rng = np.random.default_rng(20180105)
    cases = []
    for n_spk, nj, lo, hi in [(7, 3, 1, 6), (40, 8, 1, 30), (13, 13, 1, 4), (100, 8, 5, 6), (9, 4, 1, 40), (64, 32, 1, 12), (5, 8, 1, 3),
                              (31, 2, 1, 9), (200, 7, 1, 3)]:
        counts = rng.integers(lo, hi + 1, n_spk)
        utts, spks = [], []
        for s, c in enumerate(counts):
            for k in range(int(c)):
                utts.append("spk%03d-utt%03d" % (s, k))
                spks.append("spk%03d" % s)
        pos = {u: i for i, u in enumerate(utts)}
        for by_speaker in (True, False):
            res = run(utts, spks, nj, by_speaker)
            if "shards" in res:
                res["shards"] = [[pos[u] for u in sh] for sh in res["shards"]]   # positions in the input list
            # utterance k of speaker s is "spk%03d-utt%03d" % (s, k): the list is rebuilt from the per-speaker counts
            cases.append({"nj": nj, "by_speaker": by_speaker, "counts": [int(c) for c in counts], **res})
    # a list whose speakers are NOT contiguous (the reference groups a speaker's lines under its first appearance)
    utts = ["u%02d" % i for i in range(24)]
    spks = ["s%d" % (i * 7 % 5) for i in range(24)]
    res = run(utts, spks, 3, True)
    res["shards"] = [[utts.index(u) for u in sh] for sh in res["shards"]]
    cases.append({"nj": 3, "by_speaker": True, "utts": utts, "spks": spks, **res})
    os.makedirs(os.path.join(HERE, "split_scp"), exist_ok=True)
    with open(os.path.join(HERE, "split_scp", "cases.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    print("wrote %d cases (%d with an error from the reference)" % (len(cases), sum(1 for c in cases if c.get("error"))))


if __name__ == "__main__":
    main()
