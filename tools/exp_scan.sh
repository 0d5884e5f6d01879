#!/bin/bash
# ablation variants of the fp16 scan kernel (results wrong, timing only), GPU box
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  VPN_EXTRA_FLAGS="$v" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
  echo "== $v"; python tools/time_chamfer.py mfma16 2>&1 | grep "^mfma16"
done
VPN_EXTRA_FLAGS="" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
