#!/bin/bash
# the multi-GPU code paths on ONE GPU: the tree against an older build under ab_old/<name> (its own bench.py and library), alternating on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; OLD=${2:-r04}
run() { # dir route workload
  (cd $1 && RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29600 + RANDOM % 300)) TC_BENCH_FORCE_DIST=1 python bench.py --route $2 --workload $3 --steps 100 --warmup 10 --no-cpu 2>/tmp/d1.err | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,2), 'G', round(1e3*d['ms_per_step'],1), 'us/step')"; grep "host us" /tmp/d1.err | tail -1 | cut -c1-160)
}
for rep in 1 2; do for route in exchange replicate; do for w in uniform zipf; do
  echo "HEAD  $route $w: $(run $R $route $w | tr '\n' ' ')"
  echo "$OLD   $route $w: $(run $R/ab_old/$OLD $route $w | tr '\n' ' ')"
done; done; done | tee $O/distab.txt
