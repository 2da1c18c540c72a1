#!/bin/bash
# the host's enqueue loop against the device, both streams (tools/host_bound.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
for S in uniform zipf; do timeout 120 python tools/host_bound.py 200 fixed $S 2>&1 | grep piped; done | tee $O/hostbound.txt
echo "TCGPU_HOT=0"; TCGPU_HOT=0 timeout 120 python tools/host_bound.py 200 fixed zipf 2>&1 | grep piped | tee -a $O/hostbound.txt
