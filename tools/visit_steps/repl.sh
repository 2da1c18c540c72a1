#!/bin/bash
# replicate on one rank with and without the router (TC_BENCH_SKIP_ROUTER=1): what does the router cost a step?
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
for SK in 0 1 0 1; do
  TC_BENCH_SKIP_ROUTER=$SK TC_BENCH_FORCE_DIST=1 timeout 90 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29537 bench.py --gpus 1 --steps 100 --warmup 10 --route replicate > $O/repl_$SK.txt 2> $O/repl_$SK.err
  tail -1 $O/repl_$SK.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('skip_router=$SK', round(d['ms_per_step']*1e3,1), 'us/step', round(d['value']/1e9,2), 'G/s eval', (d.get('roofline') or {}).get('avg_ms'))"; grep "host us" $O/repl_$SK.err
done | tee $O/repl.txt
