#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29535 TC_BENCH_FORCE_DIST=1 timeout 120 rocprofv3 --hip-trace --kernel-trace -d $O/x_api -o s -- python $R/bench.py --gpus 1 --steps 40 --warmup 10 --route exchange > $O/x_api.log 2>&1; echo "rc=$?"
cd $R; python tools/hip_api_counts.py $O/x_api 2>&1 | head -60; rm -rf $O/x_api
