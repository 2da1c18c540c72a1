#!/bin/bash
# kernel sequence of the string-key leg (configs[4], both key sets): what runs beside what
TAG=$1; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace -d $O/ks -o t -- python $R/tools/keys_only.py > $O/keysseq.log 2>&1
tail -2 $O/keysseq.log
python $R/tools/trace_seq.py $O/ks 0 4000 > $O/keys_trace_seq.txt 2>&1
rm -rf $O/ks
cd $R
