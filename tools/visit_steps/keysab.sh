#!/bin/bash
# string keys (configs[4]): the library in the tree against ab_old/prefuse (k_bind still a kernel), hot form on / off -- one box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
cp $R/throttlecrab_amd/libtcgpu.so /tmp/new.so
for rep in 1 2; do for LIB in new old; do for HOT in 1 0; do
  if [ $LIB = old ]; then cp $R/ab_old/prefuse/throttlecrab_amd/libtcgpu.so $R/throttlecrab_amd/libtcgpu.so; else cp /tmp/new.so $R/throttlecrab_amd/libtcgpu.so; fi
  echo -n "lib=$LIB TCGPU_HOT=$HOT: "; KEYS_NO_VERIFY=1 TCGPU_HOT=$HOT timeout 300 python tools/keys_only.py 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: (round(v['ms_per_step']*1e3,1), round(v['value']/1e9,2)) for k,v in d.items() if isinstance(v, dict)})"
done; done; done | tee $O/keysab.txt
cp /tmp/new.so $R/throttlecrab_amd/libtcgpu.so
