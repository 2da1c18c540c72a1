#!/bin/bash
# Zipf stream, rank form: evaluation items per lane A/B (one process per configuration)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
for CFG in "" "TCGPU_EVAL_ITEMS=2" "TCGPU_EVAL_ITEMS=1" "TCGPU_HOT_RANK=0" "TCGPU_HOT=0" "TCGPU_EVAL_ITEMS=2 TCGPU_AUX_STREAMS=2" ; do
  echo "== $CFG"; env $CFG timeout 120 python tools/host_bound.py 200 fixed zipf 2>&1 | grep "piped=True"
done | tee $O/zipfab.txt
