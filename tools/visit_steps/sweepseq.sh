#!/bin/bash
# kernel sequence of the string-key leg around its sweeps (tools/profile_keys.py short 12: sweeps behind batches 3, 7, 11), the sweep
# beside the newest evaluation (TCGPU_SWEEP_ASIDE=1) and behind it (=0)
TAG=$1; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for A in 1 0; do
  TCGPU_SWEEP_ASIDE=$A timeout 420 rocprofv3 --kernel-trace -d $O/ss$A -o t -- python $R/tools/profile_keys.py short 12 > $O/sweepseq$A.log 2>&1
  tail -1 $O/sweepseq$A.log
  python $R/tools/trace_seq.py $O/ss$A 0 100000 > $O/sweep_seq_aside$A.txt 2>&1
  rm -rf $O/ss$A
done
cd $R
