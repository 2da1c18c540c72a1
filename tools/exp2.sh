#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp2; mkdir -p $O; cd $R
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17"
for it in 8 16; do
  $H -DRS_ABLATE=0 -DSB_ITEMS=$it tools/sortbench.hip -o /tmp/sb_$it || exit 1
  for n in 1048576 5000 300000 16789561 67108864; do timeout 60 /tmp/sb_$it $n >> $O/sortbench.txt 2>&1; echo "rc=$?" >> $O/sortbench.txt; done
done
cat $O/sortbench.txt
