#!/bin/bash
# GPU box: wall time of a 64-utterance job of bin/nnet3-xvector-compute per topology, with and without calibration (XVEC_TIMING stages)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export XVEC_TIMING=1
for topo in v2_xvector v5_cvector; do
  for cal in true false; do
    python3 $R/tools/bench_cli.py 64 400 --topology=$topo --calibrate=$cal | python3 -c '
import json,sys
d=json.loads(sys.stdin.read()); print(d["topology"], sys.argv[1], "wall %.3f s" % d["wall_s"], [l.split(") ",1)[-1][:230] for l in d["tail"] if "calibration" in l])' $cal
  done
done
