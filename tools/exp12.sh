#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp12; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout 300 python tools/stage_general.py 2>&1 | grep -v amdgpu.ids > $O/general.txt; cat $O/general.txt
