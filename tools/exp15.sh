#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp15; mkdir -p $O; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench.hip -o /tmp/microbench || exit 1
for k in 1000000 2000000 5000000 10000000 20000000 40000000; do /tmp/microbench $k 2>&1 | grep -E "keys|16B \+ store 16B \+ scattered|gather 16B \+ coalesced|32B \+ store 16B \+ scattered" >> $O/mb.txt; done
cat $O/mb.txt
