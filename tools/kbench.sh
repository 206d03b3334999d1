#!/bin/bash
# GPU box: per-kernel times of bench.py for several library builds / precisions: tools/kbench.sh <outdir> <label:lib:prec> ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/$1; shift; mkdir -p $out
for spec in "$@"; do
  IFS=: read label lib prec <<< "$spec"
  if [ "$lib" != "-" ]; then export XVEC_LIB=$R/$lib; else unset XVEC_LIB; fi
  python3 $R/bench.py --precision $prec --no-cpu-baseline --no-extra-modes > $out/$label.json 2> $out/$label.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1])
    k=d["kernels_ms_per_step"]
    print("%-10s %7.0f utt/s  frac %.3f  err %.2e | "%("$label", d["value"], d["roofline"]["frac"], d["parity_rel_err_vs_oracle_fp32"]) + "  ".join("%s %.4f"%(n.split(":")[-1].split()[0].replace(".batchnorm","").replace(".affine",""), v) for n,v in k.items()))
except Exception as e:
    print("$label failed", e, open("$out/$label.err").read()[-300:])
PY
done
