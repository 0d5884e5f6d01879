#!/bin/bash
# usage: tools/exp_emd.sh   (GPU box, repo root): the auction kernels side by side, B = 64 / 8 / 256 at n = 2048
cd $GRAFT_REPO_ROOT
for cfg in "${@:-new_auto: new_g4:VPN_EMD_GROUP=4 coop:VPN_EMD_COOP_LAUNCH=1 old_grid:VPN_EMD_GRID1=1}"; do
  for c in $cfg; do
  name=${c%%:*}; envs=${c#*:}
  echo "== $name ($envs)"
  env ${envs//,/ } python tools/time_emd.py 2>&1 | grep -E "B=64|B=8 |step clouds"
  done
done
