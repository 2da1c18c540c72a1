#!/bin/bash
# soak: wide fuzz sweep + the duplicate-heavy / hot-key / pipelined tests repeated
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/soak; mkdir -p $O; cd $R
TC_FUZZ_SEEDS=${1:-150} timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 $O/fuzz.log
for i in 1 2 3 4 5 6; do
  timeout 600 python -m pytest tests/test_gpu_slots.py tests/test_gpu_keys.py -m gpu -x -q --durations=3 -k "heavy_duplicates or hot_keys or pipelined or whole_batch or full_size or config5 or async" > $O/rep$i.log 2>&1; echo "rep $i rc=$?"; tail -1 $O/rep$i.log
done
