#!/bin/bash
# Register / scratch usage of every kernel in csrc/kernels.hip as hipcc reports it (no GPU needed).
# usage: tools/kernel_resources.sh [filter-regex]
cd "$(dirname "$0")/../speaker-embedding-with-phonetic-information_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c kernels.hip -o /tmp/kernel_resources.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/remark: Function Name:/ {name=$5}
       /remark: +TotalSGPRs:/ {sg=$4} /remark: +VGPRs:/ {v=$4} /remark: +AGPRs:/ {ag=$4}
       /remark: +ScratchSize/ {sc=$5} /remark: +Occupancy/ {oc=$5}
       /remark: +VGPRs Spill:/ {print name, "vgpr="v, "agpr="ag, "sgpr="sg, "scratch="sc, "occ="oc, "vspill="$5}' |
  sort -u | while read -r n rest; do echo "$(echo "$n" | c++filt | sed 's/void xv:://; s/(xv::GemmArgs)//') $rest"; done | grep -E "${1:-.}"
