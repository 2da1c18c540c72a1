#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp6; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for cfg in "1 2" "1 3" "2 3" "2 4" "3 4"; do set -- $cfg
echo "== aux $1 depth $2" >> $O/out.txt
TCGPU_AUX_STREAMS=$1 TCGPU_PIPE_DEPTH=$2 timeout 300 python tools/stage_bench.py 30 1048576 1 2>&1 | grep -v amdgpu.ids | grep bits >> $O/out.txt
done
echo "== aux 3 depth 4 GPU_MAX_HW_QUEUES=8" >> $O/out.txt
GPU_MAX_HW_QUEUES=8 TCGPU_AUX_STREAMS=3 TCGPU_PIPE_DEPTH=4 timeout 300 python tools/stage_bench.py 30 1048576 1 2>&1 | grep bits >> $O/out.txt
echo "== bench default" >> $O/out.txt
timeout 600 python bench.py --no-cpu --no-also >> $O/out.txt 2>&1
cat $O/out.txt
