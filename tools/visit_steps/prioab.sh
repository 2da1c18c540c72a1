#!/bin/bash
# grouping streams at the lowest (default) against the highest priority: driver form and 200 steps, both streams
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
for WL in uniform zipf; do for CFG in "TCGPU_AUX_PRIORITY=0" "TCGPU_AUX_PRIORITY=1"; do for ST in "20 5" "200 10"; do
  set -- $ST
  echo -n "$WL $CFG steps $1: "; env $CFG timeout 300 python bench.py --steps $1 --warmup $2 --no-also --no-cpu --no-verify --workload $WL 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), 'us/step', round(d['value']/1e9,2), 'G/s  median5', d.get('ms_per_step_median5'), 'kernel', d['roofline']['avg_ms'])"
done; done; done | tee $O/prioab.txt
