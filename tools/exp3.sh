#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp3; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 300 python tools/stage_bench.py 20 1048576 1 > $O/stage_piped.txt 2>&1; cat $O/stage_piped.txt
timeout 300 python tools/stage_bench.py 20 1048576 0 > $O/stage_serial.txt 2>&1; cat $O/stage_serial.txt
