#!/bin/bash
# general batches: the variant without the allowed-runs rule on / off, driver form and 200 steps (one process each)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
for WL in general_zipf general; do for CFG in "TCGPU_GENERAL_LEAN=1" "TCGPU_GENERAL_LEAN=0"; do for ST in "20 5" "200 10"; do
  set -- $ST
  echo -n "$WL $CFG steps $1: "; env $CFG timeout 300 python bench.py --steps $1 --warmup $2 --no-also --no-cpu --no-verify --workload $WL 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), 'us/step', round(d['value']/1e9,2), 'G/s  kernel', d['roofline']['avg_ms'])"
done; done; done | tee $O/genlean.txt
