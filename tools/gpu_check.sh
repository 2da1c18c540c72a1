#!/bin/bash
# quick GPU visit: full parity suite + per-stage timings (pipelined and in order)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/check; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 300 python tools/stage_bench.py 30 1048576 1 2>&1 | grep -v amdgpu.ids > $O/stages.txt
timeout 300 python tools/stage_bench.py 30 1048576 0 2>&1 | grep -v amdgpu.ids >> $O/stages.txt
cat $O/stages.txt
