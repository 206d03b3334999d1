#!/bin/bash
# GPU box: tools/stress_streamk.py over arithmetic modes, kernel families and topologies (two contexts concurrently; bit-identity
# of every result with the first of its shape).  usage: tools/stress_matrix.sh [iterations]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-600}
for topo in v2_xvector v5_cvector; do
  for prec in auto fp16mx2 fp16x3 fp16x2 fp16 bf16 default; do
    for env in "" "XVEC_GEMM_VARIANT=2" "XVEC_P8=0"; do
      r=$(env $env timeout 400 python3 $R/tools/stress_streamk.py $N $prec $topo 2>&1 | grep -E "^solo:|^two contexts" | tr '\n' ' ' | cut -c1-200)
      echo "$topo $prec [$env] $r"
    done
  done
done
