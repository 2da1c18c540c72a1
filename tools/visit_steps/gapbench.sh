#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
GPU_MAX_HW_QUEUES=8 timeout 120 $R/tools/bin/gapbench > $O/gapbench.txt 2>&1; echo "gapbench rc=$?"; cat $O/gapbench.txt
