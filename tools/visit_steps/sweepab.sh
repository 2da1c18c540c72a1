#!/bin/bash
# string keys (configs[4]): the sweep beside the newest batch's evaluation (TCGPU_SWEEP_ASIDE=1, the default) against every sweep
# behind every evaluation (=0), alternating on one box; verified runs first (the oracle replays both legs)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O
for A in 1 0; do
  echo -n "verified TCGPU_SWEEP_ASIDE=$A: "; TCGPU_SWEEP_ASIDE=$A timeout 600 python tools/keys_only.py 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: (round(v['ms_per_step']*1e3,1), round(v['value']/1e9,2)) if isinstance(v, dict) else v for k,v in d.items()})"
done | tee $O/sweepab.txt
for rep in 1 2 3; do for A in 1 0; do
  echo -n "TCGPU_SWEEP_ASIDE=$A: "; KEYS_NO_VERIFY=1 TCGPU_SWEEP_ASIDE=$A timeout 300 python tools/keys_only.py 2>/dev/null | head -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: (round(v['ms_per_step']*1e3,1), round(v['value']/1e9,2)) for k,v in d.items() if isinstance(v, dict)})"
done; done | tee -a $O/sweepab.txt
