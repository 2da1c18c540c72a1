#!/bin/bash
# round 6: the range path's new first half and the hot form against round 4's kernels and the LSD passes (tools/partbench.hip, prebuilt into tools/bin)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
timeout 180 $R/tools/bin/partbench > $O/partbench.txt 2>&1; echo "partbench rc=$?"; cat $O/partbench.txt
