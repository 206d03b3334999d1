#!/bin/bash
# GPU box: same-box A/B of one environment switch on the bench workload: kernel times (bench.py's own HIP events, alternating
# runs) and L2-miss traffic of the GEMM kernels (rocprofv3 --pmc, separate passes).
#   tools/ab_env.sh <out dir under gpurun_out> <VAR> <value A> <value B> [bench.py arguments ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/$1; var=$2; va=$3; vb=$4; shift 4
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-extra-modes --no-parity-sweep $*"
for rep in 1 2 3; do
  for v in $va $vb; do
    env $var=$v python3 $B --steps 40 --warmup 5 > "$out/bench_${var}_${v}_$rep.json" 2> "$out/bench_${var}_${v}_$rep.err"
  done
done
for v in $va $vb; do
  export $var=$v
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$out/pmc_${v}_$c" -o bench -- python3 $B --lanes 1 --steps 3 --warmup 1 > "$out/pmc_${v}_$c.log" 2>&1
  done
done
python3 - "$out" $var $va $vb <<'PY'
import csv, glob, json, os, sys
out, var, va, vb = sys.argv[1:5]
for v in (va, vb):
    rows = []
    for f in sorted(glob.glob("%s/bench_%s_%s_*.json" % (out, var, v))):
        try:
            j = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception as e:
            print(f, "unreadable", e); continue
        rows.append(j)
    if not rows: continue
    ks = rows[0]["kernels_ms_per_step"].keys()
    print("%s=%s: value %s utt/s, ms/step %s" % (var, v, [round(r["value"]) for r in rows], [round(r["ms_per_step"], 4) for r in rows]))
    for k in ks:
        print("    %-90s %s" % (k[-90:], ["%.4f" % r["kernels_ms_per_step"][k] for r in rows]))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot = {}
        for f in glob.glob("%s/pmc_%s_%s/**/*counter_collection.csv" % (out, v, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") != c: continue
                k = r["Kernel_Name"][:70]
                t = tot.setdefault(k, [0, 0.0]); t[0] += 1; t[1] += float(r["Counter_Value"])
        for k, (n, s) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:8]:
            print("    %s %-70s launches %4d  mean %.1f MB%s" % (c, k, n, s / n * 1024 / 1e6 * (2 if c == "FETCH_SIZE" else 1), " (x2 applied)" if c == "FETCH_SIZE" else ""))
PY
find "$out" -name "*.csv" -size +4M -delete
