#!/bin/bash
# k_eval_general's look-back several windows per round trip (late round 6): the library in the tree against ab_old/prewalk, general
# batches of both streams, 20 and 200 timed batches, old and new alternating on one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
cp $R/throttlecrab_amd/libtcgpu.so /tmp/new.so
for rep in 1 2; do for WL in general_zipf general; do for LIB in new old; do for ST in 20 200; do
  if [ $LIB = old ]; then cp $R/ab_old/prewalk/libtcgpu.so $R/throttlecrab_amd/libtcgpu.so; else cp /tmp/new.so $R/throttlecrab_amd/libtcgpu.so; fi
  echo -n "$WL lib=$LIB steps $ST: "; timeout 300 python bench.py --steps $ST --warmup 5 --no-also --no-cpu --no-verify --workload $WL 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), 'us/step', round(d['value']/1e9,2), 'G/s  kernel', round(d['roofline']['avg_ms']*1e3,1))"
done; done; done; done | tee $O/walkab.txt
cp /tmp/new.so $R/throttlecrab_amd/libtcgpu.so
