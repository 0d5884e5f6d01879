#!/bin/bash
# usage: tools/exp_emd_trace.sh  (GPU box): rebuild with -DEMD_TRACE, print the timelines, rebuild clean
cd $GRAFT_REPO_ROOT
VPN_EXTRA_FLAGS="-DEMD_TRACE ${EMD_TRACE_EXTRA}" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
for m in ${EMD_TRACE_MODES:-step uniform}; do echo "=== $m"; python tools/emd_timeline.py $m 2>&1 | grep -v amdgpu.ids | head -58; done

VPN_EXTRA_FLAGS="" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
