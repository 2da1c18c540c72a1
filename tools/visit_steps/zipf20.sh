#!/bin/bash
# the driver's form of the Zipf stream (20 timed batches behind 5) under the kernel trace: the per-batch timeline + every kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace -d $O/tl20 -o t -- python $R/bench.py --profile-run --steps 20 --warmup 5 --layout fixed --workload ${WL:-zipf} > $O/tl20.log 2>&1
python $R/tools/timeline.py $O/tl20 0 40 > $O/timeline_${WL:-zipf}20.txt 2>&1; cat $O/timeline_${WL:-zipf}20.txt | grep -v "kernels columns"
python $R/tools/kernel_list.py $O/tl20 k_probe_occupy > $O/kernels_${WL:-zipf}20.txt 2>&1; rm -rf $O/tl20
