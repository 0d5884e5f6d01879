#!/bin/bash
# usage: tools/exp_c2.sh  (GPU box): C2 (module path) raster step with 5 / 4 waves per SIMD for raster_total_kernel<false>
cd $GRAFT_REPO_ROOT
for v in "-DR_TOTAL_WAVES=5" "-DR_TOTAL_WAVES=4"; do
  VPN_RASTER_FLAGS="$v" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
  for i in 1 2; do
  python bench.py --workload c2 --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['c2']; print('$v', r['ms_per_step'], {k:v['avg_us'] for k,v in r['kernel_us'].items()})"
  done
done
VPN_RASTER_FLAGS="" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
