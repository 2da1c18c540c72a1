#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp18; mkdir -p $O; cd $R
for b in 262144 4194304 8388608; do
echo "== batch $b" >> $O/out.txt
timeout 300 python tools/stage_bench.py 12 $b 1 2>&1 | grep -v amdgpu.ids | grep bits >> $O/out.txt
timeout 300 python tools/stage_bench.py 12 $b 0 2>&1 | grep -v amdgpu.ids | grep bits >> $O/out.txt
done
cat $O/out.txt
