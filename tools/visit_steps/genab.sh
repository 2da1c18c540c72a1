#!/bin/bash
# driver-form A/B (20 steps, 5 warm-up, fresh process each): Zipf and general Zipf with the hot form on / off
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
for WL in zipf general_zipf; do for CFG in "TCGPU_HOT=1" "TCGPU_HOT=0" "TCGPU_HOT_RANK=0"; do
  echo -n "$WL $CFG: "; env $CFG timeout 300 python bench.py --steps 20 --warmup 5 --no-also --no-cpu --no-verify --workload $WL 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), 'us/step', round(d['value']/1e9,2), 'G/s  median5', d.get('ms_per_step_median5'), 'kernel', d['roofline']['avg_ms'], d.get('engine_info', {}).get('grouping_path'))"
done; done | tee $O/genab.txt
