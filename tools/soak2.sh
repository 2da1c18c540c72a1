#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/soak2; mkdir -p $O; cd $R
for i in $(seq 1 12); do
  s=$(date +%s.%N)
  timeout 600 python -m pytest tests/test_gpu_slots.py tests/test_gpu_keys.py -m gpu -x -q --durations=3 -k "heavy_duplicates or hot_keys or pipelined or whole_batch or full_size or config5" > $O/rep$i.log 2>&1
  e=$(date +%s.%N); echo "rep $i rc=$? $(echo "$e - $s" | bc) s"; grep -E "^[0-9.]+s (call|setup)" $O/rep$i.log | head -3
done
