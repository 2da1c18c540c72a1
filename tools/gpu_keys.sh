#!/bin/bash
# string-key mode: parity tests + stage timings + kernel trace
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/keys; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python tools/stage_keys.py 16 2>&1 | grep -v amdgpu.ids | tail -2
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python $R/tools/stage_keys.py 8 > $O/stats.log 2>&1; echo "stats rc=$?"
