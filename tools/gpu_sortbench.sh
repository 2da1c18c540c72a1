#!/bin/bash
# times rs::k_hist / rs::k_onesweep alone and checks the result (tools/sortbench.hip), then the pipeline stages
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSB_ITEMS=8 tools/sortbench.hip -o /tmp/sb || exit 1
for n in 1048576 5000 300000 16789561; do timeout 60 /tmp/sb $n; done
timeout 300 python tools/stage_bench.py 40 1048576 1 2>&1 | grep -v amdgpu.ids | grep bits
timeout 300 python tools/stage_bench.py 40 1048576 0 2>&1 | grep -v amdgpu.ids | grep bits
