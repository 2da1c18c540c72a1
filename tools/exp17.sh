#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp17; mkdir -p $O; cd $R
export GPU_MAX_HW_QUEUES=8
for cfg in "2 3" "3 4" "4 5" "4 6" "3 6"; do set -- $cfg
echo "== aux $1 depth $2 (8 HW queues)" >> $O/out.txt
TCGPU_AUX_STREAMS=$1 TCGPU_PIPE_DEPTH=$2 timeout 300 python tools/stage_bench.py 40 1048576 1 2>&1 | grep -v amdgpu.ids | grep bits >> $O/out.txt
done
cat $O/out.txt
