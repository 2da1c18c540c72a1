#!/bin/bash
# the multi-GPU code paths on ONE GPU (TC_BENCH_FORCE_DIST=1: process group, router, RCCL all-gather with one rank): both modes, both streams
R=${GRAFT_REPO_ROOT:-$(pwd)}
for route in replicate exchange; do
  for w in uniform zipf; do
    TC_BENCH_FORCE_DIST=1 python $R/bench.py --route $route --workload $w --steps ${1:-100} --warmup 10 --no-cpu > /tmp/d1.out 2> /tmp/d1.err
    grep "host us" /tmp/d1.err | tail -1
    grep '^{' /tmp/d1.out | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$route', '$w', round(d['value']/1e9,2), 'G decisions/s', round(1e3*d['ms_per_step'],1), 'us per step')" || tail -3 /tmp/d1.err
  done
done
