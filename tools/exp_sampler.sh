#!/bin/bash
# build variants (VPN_EXTRA_FLAGS) and report the per-kernel times of the C3 step (GPU box); ablations give wrong results
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  VPN_EXTRA_FLAGS="$v" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
  echo "== $v"; python bench.py --steps 50 --warmup 10 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['hip_event_ms_per_step']['median'], {k:v['avg_us'] for k,v in d['kernel_us'].items()})"
done
VPN_EXTRA_FLAGS="" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
