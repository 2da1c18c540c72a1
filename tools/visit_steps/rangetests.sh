#!/bin/bash
# the grouping paths' tests: range path, hot form, timed configuration, prefill, fuzz
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_range.py tests/test_gpu_timed_config.py tests/test_gpu_prefill.py tests/test_gpu_fuzz.py tests/test_gpu_bucket.py -m gpu -x -q --timeout 400 > $O/pytest_range.log 2>&1; echo "range tests rc=$?"; tail -15 $O/pytest_range.log
