#!/bin/bash
# build variants and time them (GPU box)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  VPN_EXTRA_FLAGS="$v" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
  echo "== $v"; python tools/time_raster.py 2>&1 | grep "B="
done
