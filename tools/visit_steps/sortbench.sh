#!/bin/bash
# the grouping kernels alone: LSD passes vs the range path (tools/sortbench.hip, prebuilt into tools/bin by the caller)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
timeout 120 $R/tools/bin/sortbench > $O/sortbench.txt 2>&1; echo "sortbench rc=$?"; cat $O/sortbench.txt
