#!/bin/bash
# the evaluation's bare memory pattern: 8-byte column, streamed through LDS, 4-byte column (tools/streambench.hip, prebuilt into tools/bin)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
for i in 1 2 3; do timeout 60 $R/tools/bin/streambench; done > $O/streambench.txt 2>&1; echo "streambench rc=$?"; cat $O/streambench.txt
