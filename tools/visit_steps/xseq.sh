#!/bin/bash
# kernel sequence of the exchange on one rank (TC_BENCH_FORCE_DIST=1): what a step is made of on the device
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for ROUTE in exchange replicate; do
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29536 TC_BENCH_FORCE_DIST=1 timeout 160 rocprofv3 --kernel-trace -d $O/xs -o s -- python $R/bench.py --gpus 1 --steps 40 --warmup 10 --route $ROUTE --no-cpu > $O/xseq_$ROUTE.log 2>&1; echo "rc=$?"
python $R/tools/trace_seq.py $O/xs -160 150 > $O/xseq_$ROUTE.txt 2>&1; rm -rf $O/xs
done
cd $R
