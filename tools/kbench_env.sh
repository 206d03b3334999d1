#!/bin/bash
# GPU box: per-kernel times of bench.py under different environments: tools/kbench_env.sh <outdir> <label:prec:VAR=val,VAR=val> ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/$1; shift; mkdir -p $out
for spec in "$@"; do
  IFS=: read label prec envs <<< "$spec"
  ( IFS=,; for kv in $envs; do export "$kv"; done
    python3 $R/bench.py --precision $prec --no-cpu-baseline --no-extra-modes > $out/$label.json 2> $out/$label.err )
  python3 - <<PY
import json
try:
    d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1])
    k=d["kernels_ms_per_step"]
    print("%-10s %7.0f utt/s  frac %.3f  err %.2e worst-vs-x3 %s | "%("$label", d["value"], d["roofline"]["frac"], d["parity_rel_err_vs_oracle_fp32"], ("%.2e" % d["parity_worst_vs_fp16x3"]["value"]) if "parity_worst_vs_fp16x3" in d else "-") + "  ".join("%s %.4f"%(n.split(":")[-1].split()[0].replace(".batchnorm","").replace(".affine",""), v) for n,v in k.items()))
    print("           ", " ".join(sorted(set(n.split()[-1] for n in k if " " in n))))
except Exception as e:
    print("$label failed", e, open("$out/$label.err").read()[-600:])
PY
done
