#!/bin/bash
# build variants of the Chamfer kernels and time them (GPU box)
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  VPN_EXTRA_FLAGS="$v" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
  echo "== $v"; python tools/time_chamfer.py 2>&1 | tail -4
done
