#!/bin/bash
# GPU box: tools/stress_streamk.py over arithmetic modes, kernel families and topologies (two contexts concurrently; bit-identity
# of every result with the first of its shape).  usage: tools/stress_matrix.sh [iterations]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-600}
[ -n "$STRESS_SKIP_INPROC" ] || for topo in v2_xvector v5_cvector; do
  for prec in auto fp16mx2 fp16x3 fp16x2 fp16 bf16 default; do
    for env in "" "XVEC_DEBUG=gemm_variant=2" "XVEC_DEBUG=p8=0"; do
      r=$(env $env timeout 400 python3 $R/tools/stress_streamk.py $N $prec $topo 2>&1 | grep -E "^solo:|^two contexts" | tr '\n' ' ' | cut -c1-200)
      echo "$topo $prec [$env] $r"
    done
  done
done
# the multi-PROCESS leg: what the in-process tools above do not model - four (and eight) independent
# nnet3-xvector-compute processes on one GPU, each a fresh context with its own first batches, under a noise process; the
# arithmetic fixed by the model, shared through a calibration file that all of them race to publish, and measured per process
M=${2:-10}
for topo in v2_xvector v5_cvector; do
  for mode in fixed shared self; do
    for procs in 4 8; do
      r=$(timeout 900 python3 $R/tools/repro_four_procs.py --topology $topo --mode $mode --procs $procs --iters $M --noise 600 2>&1 | grep -E "^TOTAL|differ \(|exit [0-9]" | tr '\n' ' ' | cut -c1-300)
      echo "$topo processes=$procs mode=$mode: $r"
    done
  done
done
