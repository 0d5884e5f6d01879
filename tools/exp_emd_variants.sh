#!/bin/bash
# usage: tools/exp_emd_variants.sh   (GPU box): walk / grid variants of the team auction kernel x team-size knobs
cd $GRAFT_REPO_ROOT
for v in "-DEMD_WALK_SPLIT" "-DEMD_WALK_SPLIT -DEMD_EG=8 -DEMD_EGX=32" "-DEMD_EG=16 -DEMD_EGX=16" "-DEMD_EG=8 -DEMD_EGX=32"; do
  VPN_EXTRA_FLAGS="$v" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
  for t in "1024 64" "512 16" "1024 16" "2048 64" "256 8"; do
    set -- $t
    echo "== [$v] tnum=$1 tmax=$2: $(VPN_EMD_TNUM=$1 VPN_EMD_TMAX=$2 python tools/time_emd_one.py uniform 2>&1 | grep ms) | $(VPN_EMD_TNUM=$1 VPN_EMD_TMAX=$2 python tools/time_emd_one.py step 2>&1 | grep ms) | $(VPN_EMD_TNUM=$1 VPN_EMD_TMAX=$2 VPN_EMD_GROUP=4 python tools/time_emd_one.py uniform 2>&1 | grep ms) g4"
  done
done
VPN_EXTRA_FLAGS="" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
