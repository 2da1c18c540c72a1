#!/bin/bash
# pipelined 1 Mi batches when the process has used torch's stream BEFORE the engine created its own (BS_PRE=1): knobs
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
run() { echo "== $*"; env "$@" BS_SIZES=1048576 BS_MODES=piped timeout 120 python tools/batch_sizes.py 2>&1 | grep -v "amdgpu.ids\|batch |" ; }
{
run BS_X=0
run BS_PRE=1
run BS_PRE=1 TCGPU_PIPE_PROBE=0
run BS_X=0 TCGPU_PIPE_PROBE=0
run BS_PRE=1 GPU_MAX_HW_QUEUES=4
run BS_X=0 GPU_MAX_HW_QUEUES=4
run BS_PRE=1 GPU_MAX_HW_QUEUES=4 TCGPU_PIPE_PROBE=0
run BS_PRE=1 GPU_MAX_HW_QUEUES=16
} 2>&1 | tee $O/premap.txt
