#!/bin/bash
# experiment: sorted positions per lane in the uniform evaluation (TCGPU_EVAL_ITEMS)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/items; mkdir -p $O; cd $R
for it in ${ITEMS_TEST:-8}; do
  TCGPU_EVAL_ITEMS=$it timeout 600 python -m pytest tests/test_gpu_slots.py tests/test_gpu_fuzz.py tests/test_gpu_metrics.py -m gpu -x -q > $O/pytest_$it.log 2>&1; echo "items=$it pytest rc=$?"; tail -3 $O/pytest_$it.log
done
export TC_STAGE_KINDS=uniform,zipf TC_STAGE_MODES=bits,dec
for it in ${ITEMS_BENCH:-2 4 8}; do
  echo "== items $it"
  TCGPU_EVAL_ITEMS=$it timeout 300 python tools/stage_bench.py 100 1048576 1 2>&1 | grep -v amdgpu.ids
  TCGPU_EVAL_ITEMS=$it timeout 300 python tools/stage_bench.py 100 1048576 0 2>&1 | grep -v amdgpu.ids
done | tee $O/stages.txt
