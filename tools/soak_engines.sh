#!/bin/bash
# GPU box: the several-engine table loop (consumer thread per engine, ordered writer; XVEC_DEBUG=engines_on_one_device test knob) against
# the one-engine job, many times over batch sizes and engine counts: archives must be byte-identical.  usage: tools/soak_engines.sh [rounds]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-30}
D=$(mktemp -d /dev/shm/xvsoak.XXXX)
python3 - "$D" <<PY
import sys, os
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tests")
import numpy as np, helpers as H
from oracle import kaldi_io as kio
d = sys.argv[1]
net, line = H.synth_model("v2_xvector")
open(d + "/final.raw", "wb").write(net.to_bytes(True))
rng = np.random.default_rng(1)
lens = rng.integers(0, 700, 3000)
utts = [("u%05d" % i, H.features(i % 97, int(t)) if t else np.zeros((0, 23), np.float32)) for i, t in enumerate(lens)]
kio.write_ark_matrices(d + "/feats.ark", utts)
PY
B=$R/speaker-embedding-with-phonetic-information_amd/bin/nnet3-xvector-compute
A="--use-gpu=yes --min-chunk-size=25 --chunk-size=10000 --output-node=tdnn6.affine"
bad=0
for i in $(seq 1 $N); do
  bf=$(( (RANDOM % 60 + 2) * 1024 ))
  ne=$(( RANDOM % 4 + 2 ))
  $B $A --batch-frames=$bf $D/final.raw ark:$D/feats.ark ark:$D/one.ark 2> $D/one.log || { echo "one-engine job failed"; bad=$((bad+1)); }
  XVEC_DEBUG=engines_on_one_device=$ne timeout 120 $B $A --batch-frames=$bf $D/final.raw ark:$D/feats.ark ark:$D/many.ark 2> $D/many.log || { echo "round $i: $ne engines, batch-frames $bf: exit $?"; bad=$((bad+1)); }
  cmp -s $D/one.ark $D/many.ark || { echo "round $i: $ne engines, batch-frames $bf: archives differ"; bad=$((bad+1)); }
done
echo "soak: $N rounds, $bad problems"
rm -rf $D
exit $bad
