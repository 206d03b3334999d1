#!/bin/bash
# Same-box A/B of two builds of libxvec_hip.so (GPU box): tools/ab_libs.sh <tag> <rounds> -- runs bench.py alternately
# with .ab/libxvec_base.so and .ab/libxvec_new.so copied over the in-tree library; JSON lines under gpurun_out/<tag>/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-ab}; rounds=${2:-2}
L=$R/speaker-embedding-with-phonetic-information_amd/libxvec_hip.so
mkdir -p $R/gpurun_out/$tag
cp $L /tmp/libxvec_keep.so
for i in $(seq 1 $rounds); do
  for v in base new; do
    cp $R/.ab/libxvec_$v.so $L
    python3 $R/bench.py --no-cpu-baseline --no-extra-modes > $R/gpurun_out/$tag/${v}_$i.json 2>/dev/null
  done
done
cp /tmp/libxvec_keep.so $L
