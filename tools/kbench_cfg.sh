#!/bin/bash
# GPU box: per-kernel times of the other BASELINE configurations (3: v5 c-vector, 5: v3 senone head fp16 ragged)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/${1:-cfg}; mkdir -p $out
run() { label=$1; shift; python3 $R/bench.py --no-cpu-baseline --no-extra-modes "$@" > $out/$label.json 2> $out/$label.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1])
    k=d["kernels_ms_per_step"]
    print("%-8s %7.0f utt/s  %.3f ms/step  %s  err %.2e"%("$label", d["value"], d["ms_per_step"], d["config"].get("arithmetic"), d["parity_rel_err_vs_oracle_fp32"]))
    for n,v in k.items(): print("     %-90s %.4f"%(n[:90], v))
except Exception as e:
    print("$label failed", e, open("$out/$label.err").read()[-400:])
PY
}
run v5 --topology v5_cvector
run v3head --topology v3_multitask --precision fp16 --output-node output_am.log-softmax --ragged 200-600
