#!/bin/bash
# build variants of the Chamfer kernels (VPN_EXTRA_FLAGS) and time the fp16 filter (GPU box); the last line of each
# block says whether the result still equals brute force
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  VPN_EXTRA_FLAGS="$v" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
  echo "== $v"; python tools/time_chamfer.py mfma16 2>&1 | grep "^mfma16\|== brute\|debug"
done
VPN_EXTRA_FLAGS="" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
