#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel stats + PMC passes.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag> [notests]
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
if [ "${2:-}" != "notests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
  tail -3 $O/pytest.log
fi
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 600 $O/bench.json
BENCH="python $R/bench.py --steps 20 --warmup 3 --no-cpu --no-also"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o s -- $BENCH > $O/stats.log 2>&1; echo "stats rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- $BENCH > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w -- $BENCH > $O/pmc_write.log 2>&1; echo "write rc=$?"
cd $R
find $O -name "*.csv" | head -20
