#!/bin/bash
# the headline in driver form, three times (separate processes), without the secondary legs
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-also --no-cpu --no-verify 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver form', round(d['ms_per_step']*1e3,1), 'us/step', round(d['value']/1e9,2), 'G/s kernel', d['roofline']['avg_ms'])"; done | tee $O/bench3.txt
