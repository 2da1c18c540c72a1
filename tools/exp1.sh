#!/bin/bash
# experiment batch: sort ablations + memory pattern microbench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp1; mkdir -p $O; cd $R
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17"
$H tools/microbench.hip -o /tmp/microbench && /tmp/microbench > $O/microbench.txt 2>&1
for ab in 0 1 2 3 4; do for it in 8 16; do
  $H -DRS_ABLATE=$ab -DSB_ITEMS=$it tools/sortbench.hip -o /tmp/sb_${ab}_${it} && /tmp/sb_${ab}_${it} >> $O/sortbench.txt 2>&1
done; done
cat $O/microbench.txt $O/sortbench.txt
