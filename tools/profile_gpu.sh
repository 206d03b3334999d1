#!/bin/bash
# Collects the rocprofv3 evidence of one bench configuration on the GPU box (run through gpurun):
#   tools/profile_gpu.sh <name> [bench.py arguments of the configuration ...]
# Output under gpurun_out/prof_<name>/; summarise with  python tools/parse_rocprof.py gpurun_out/prof_<name> <tag>.
# Counters are collected in their own passes, never together with tracing (MI355X_MICROARCH.md, HBM section).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
name=${1:-main}; shift
P=$R/gpurun_out/prof_$name
mkdir -p "$P"
cd /tmp && export TMPDIR=/tmp
# one batch in flight (--lanes 1): kernel durations then mean what they mean in bench.py's roofline, whose per-kernel HIP events
# also come from a one-lane context (two lanes overlap one batch's tail kernels with the next batch's GEMMs)
B="$R/bench.py --no-cpu-baseline --no-extra-modes --no-parity-sweep --lanes 1 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$P/kt" -o bench -- python3 $B --steps 20 --warmup 4 > "$P/kt_bench.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$P/pmc_$c" -o bench -- python3 $B --steps 3 --warmup 1 > "$P/pmc_$c.log" 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d "$P/pmc_sq" -o bench -- python3 $B --steps 3 --warmup 1 > "$P/pmc_sq.log" 2>&1
if [ "$name" = main ]; then python3 $R/bench.py "$@" > "$P/bench_default.json" 2> "$P/bench_default.err"; fi
# the raw traces are large: keep the summaries only
find "$P" -name "*kernel_trace.csv" -size +8M -delete
ls -R "$P" | head -30
