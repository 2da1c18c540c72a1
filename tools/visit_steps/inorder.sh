#!/bin/bash
# in-order batches (the ABI's default, no TC_B_INPUTS_READY): range path vs round 3's bucket path + gated sort
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
for RG in 2 1 0; do echo "TCGPU_RANGE=$RG"; TCGPU_RANGE=$RG timeout 120 python tools/host_bound.py 200 fixed 2>&1 | grep piped; done | tee $O/inorder.txt
