#!/bin/bash
# kernel sequence of the string-key leg around its sweeps (tools/profile_keys.py short 12), interleaved ranges on / off
TAG=$1; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for A in 1 0; do
  TCGPU_RANGE_ILV=$A timeout 420 rocprofv3 --kernel-trace -d $O/ks$A -o t -- python $R/tools/profile_keys.py short 12 > $O/keysseq_ilv$A.log 2>&1
  python $R/tools/trace_seq.py $O/ks$A 0 100000 | grep -v "k_probe_stamp\|k_probe_occupy\|copyBuffer" | tail -150 > $O/keys_seq_ilv$A.txt 2>&1
  rm -rf $O/ks$A
done
cd $R
