#!/bin/bash
# raster.hip built with extra compiler flags (VPN_RASTER_FLAGS), C3 step kernel times + the raster parity tests (GPU box)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  VPN_RASTER_FLAGS="$v" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
  echo "== raster flags: '$v'"
  python bench.py --steps 50 --warmup 10 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['hip_event_ms_per_step']['median'], {k:v['avg_us'] for k,v in d['kernel_us'].items()})"
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "raster or hot_path" 2>&1 | tail -1
done
VPN_RASTER_FLAGS="" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
