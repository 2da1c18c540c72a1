#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp14; mkdir -p $O; cd $R
for pr in 1 0 1 0; do
echo "== aux_priority $pr" >> $O/out.txt
TCGPU_AUX_PRIORITY=$pr timeout 300 python tools/stage_bench.py 40 1048576 1 2>&1 | grep -v amdgpu.ids | grep "bits" >> $O/out.txt
done
cat $O/out.txt
