#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp10; mkdir -p $O; cd $R
timeout 600 python tools/stage_keys.py 16 2>&1 | grep -v amdgpu.ids > $O/keys.txt; cat $O/keys.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python $R/tools/stage_keys.py 8 > $O/stats.log 2>&1; echo "stats rc=$?"
