#!/bin/bash
# GPU box: L2 (TCC) and L1 / texture-addresser (TCP, TA) counters of one bench step, two separate --pmc passes, no tracing.
# Output under gpurun_out/prof2/; summarised into profiles/<tag>_pmc_cache.md.
R=$GRAFT_REPO_ROOT; P=$R/gpurun_out/prof2; rm -rf $P; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-extra-modes --steps 3 --warmup 1"
timeout 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE --output-format csv -d $P/tcc -o b -- python3 $B > $P/tcc.log 2>&1
echo "tcc rc $?"
timeout 150 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $P/tcp -o b -- python3 $B > $P/tcp.log 2>&1
echo "tcp rc $?"
ls $P/*/ 2>/dev/null | head
