#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp13; mkdir -p $O; cd $R
TC_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu --no-also > $O/dist1.json 2> $O/dist1.err; echo "rc=$?"; tail -c 1500 $O/dist1.json; tail -5 $O/dist1.err
