#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_route.py tests/test_gpu_sharding.py -m gpu -x -q 2>&1 | tail -6
for WL in uniform zipf; do
TC_BENCH_FORCE_DIST=1 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 100 --warmup 10 --route exchange --workload $WL > $O/x1_$WL.txt 2> $O/x1_$WL.err; tail -1 $O/x1_$WL.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['route'], round(d['ms_per_step']*1e3,1), 'us/step', round(d['value']/1e9,2), 'G/s')"; grep "host us\|stages" $O/x1_$WL.err
done
cd /tmp; export TMPDIR=/tmp
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29535 TC_BENCH_FORCE_DIST=1 timeout 120 rocprofv3 --kernel-trace --stats -d $O/x_stats -o s -- python $R/bench.py --gpus 1 --steps 40 --warmup 10 --route exchange > $O/x_stats.log 2>&1; echo "rc=$?"
cd $R; python tools/summarize_prof.py ${1}_exchange_1rank $O/x_stats > /dev/null 2>&1; head -16 profiles/${1}_exchange_1rank.txt | cut -c1-170; mkdir -p $O/profiles; cp profiles/${1}_exchange_1rank* $O/profiles/; rm -rf $O/x_stats
