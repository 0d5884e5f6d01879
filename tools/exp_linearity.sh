#!/bin/bash
# usage: tools/exp_linearity.sh  (GPU box): the raster backward's W -> 2W check with the per-pixel dump of the differing tile
cd $GRAFT_REPO_ROOT
VPN_RASTER_FLAGS="-DR_DEBUG_LIN" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
python tools/raster_linearity_check.py 2>&1 | grep -v amdgpu.ids
VPN_RASTER_FLAGS="" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
