#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_slots.py tests/test_gpu_fixed.py tests/test_gpu_fuzz.py tests/test_gpu_robustness.py tests/test_gpu_timed_config.py -m gpu -x -q > $O/pytest_general.log 2>&1; echo "general tests rc=$?"; tail -4 $O/pytest_general.log
for WL in general general_zipf; do for ST in 20 200; do timeout 300 python bench.py --workload $WL --no-also --no-cpu --no-verify --steps $ST --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['stream'], d['steps'], 'steps', round(d['ms_per_step']*1e3,1), 'us/step', round(d['value']/1e9,2), 'G/s', 'kernel', d.get('roofline',{}).get('avg_ms'))"; done; done | tee $O/general_bench.txt
