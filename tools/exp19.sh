#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/exp19; mkdir -p $O; cd $R
for sl in 1 4 16 48; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DRS_SPIN_SLEEP=$sl -shared throttlecrab_amd/csrc/tcgpu.hip -o throttlecrab_amd/libtcgpu.so || exit 1
  echo "== spin sleep $sl" >> $O/out.txt
  timeout 300 python tools/stage_bench.py 40 1048576 1 2>&1 | grep -v amdgpu.ids | grep bits >> $O/out.txt
  timeout 300 python tools/stage_bench.py 20 1048576 0 2>&1 | grep -v amdgpu.ids | grep "uniform  bits" >> $O/out.txt
done
cat $O/out.txt
