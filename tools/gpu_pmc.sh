#!/bin/bash
# extra PMC passes over the in-order pipeline (one counter group per pass)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9a-z]+|TCP_[A-Z_0-9a-z]+|GRBM_[A-Z_]+|MemUnitStalled|L2CacheHit|VALUBusy|MemUnitBusy|OccupancyPercent|FetchSize|WriteSize)\b" | sort -u > $O/counters.txt
wc -l $O/counters.txt
CMD="python $R/tools/stage_bench.py 10 1048576 0"
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp -d $O/$tag -o p -- $CMD > $O/$tag.log 2>&1; echo "$grp rc=$?"
done
