#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp4; mkdir -p $O; cd $R
for d in 3 4 6; do for pr in 0 1; do
echo "== depth $d aux_priority $pr" >> $O/out.txt
TCGPU_PIPE_DEPTH=$d TCGPU_AUX_PRIORITY=$pr timeout 300 python tools/stage_bench.py 30 1048576 1 2>&1 | grep -v amdgpu.ids >> $O/out.txt
done; done
cat $O/out.txt
