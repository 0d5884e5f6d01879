#!/bin/bash
# the memory-model-strict build (-DVPN_STRICT_ORDER: acquire-release arrival adds in the raster finalisation and the EMD
# group barrier) against the default: tests that exercise both, then the step and EMD timings (GPU box)
cd $GRAFT_REPO_ROOT
for v in "-DVPN_STRICT_ORDER" ""; do
  VPN_EXTRA_FLAGS="$v" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
  echo "== build flags: '$v'"
  timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_emd.py -m gpu -x -q -k "hot_path or raster_total or config2 or emd" 2>&1 | tail -1
  python bench.py --steps 100 --warmup 10 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('step ms', d['hip_event_ms_per_step']['median'], {k:v['avg_us'] for k,v in d['kernel_us'].items()})"
  python tools/time_emd.py 2>/dev/null | head -2
done
