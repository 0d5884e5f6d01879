#!/bin/bash
# usage: tools/exp_emd.sh   (GPU box, repo root): the auction kernels side by side, B = 64 / 8 / 256 at n = 2048
cd $GRAFT_REPO_ROOT
for cfg in "new_auto:" "new_g4:VPN_EMD_GROUP=4" "new_g8:VPN_EMD_GROUP=8" "new_g2:VPN_EMD_GROUP=2" "new_plain:VPN_EMD_PLAIN_LAUNCH=1" "old_grid:VPN_EMD_GRID1=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo "== $name ($envs)"
  env $envs python tools/time_emd.py 2>&1 | grep "B="
done
