#!/bin/bash
# GPU box: SQ / LDS counters of the probe_p8 kernels on one shape: tools/probe_p8_pmc.sh <outdir> <shape prefix>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=$R/gpurun_out/$1; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d $P/pmc_sq -o p8 -- $R/build/probe_p8 3 0 abl "$2" > $P/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_F16 \
  --output-format csv -d $P/pmc_lds -o p8 -- $R/build/probe_p8 3 0 abl "$2" > $P/pmc_lds.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("pmc_sq", "pmc_lds"):
    for f in glob.glob("$P/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            agg.setdefault(k, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, c in agg.items():
            print(d, k, "  ".join("%s=%.4g" % (n, sum(v) / len(v)) for n, v in c.items()))
PY
