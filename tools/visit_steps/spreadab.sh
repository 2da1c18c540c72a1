#!/bin/bash
# string keys (configs[4]): TCGPU_SPREAD_FREE=1 (a free stack that hands out slots from all over the key space: key batches on the
# range path) against the default, with the per-stage kernel times; round 6's range path and hot form
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
for rep in 1 2; do for A in 1 0; do
  echo "TCGPU_SPREAD_FREE=$A:"; KEYS_DETAIL=1 KEYS_NO_VERIFY=1 TCGPU_SPREAD_FREE=$A timeout 300 python tools/keys_only.py 2>/dev/null | cut -c1-700
done; done | tee $O/spreadab.txt
