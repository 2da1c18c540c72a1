#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp7; mkdir -p $O; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench.hip -o /tmp/microbench && /tmp/microbench > $O/microbench.txt 2>&1
cat $O/microbench.txt
