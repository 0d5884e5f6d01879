#!/bin/bash
# A/B of emd.hip variants on ONE box (boxes differ by ~5 %): tools/ab_emd_lib.sh <name> <emd.hip variant>  ->  ab/libvpn_<name>.so
# (the other objects are the tree's; select with VPN_HIP_LIB=ab/libvpn_<name>.so)
set -e
cd "$(dirname "$0")/.."
C=volumetric-primitives-net_amd/csrc
mkdir -p ab
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function -ffp-contract=off -fno-slp-vectorize \
    -Iinclude -I$C -c "$2" -o ab/emd_$1.o
objs=""
for s in vpn_api sampler chamfer raster head mesh trainstep; do objs="$objs $C/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libvpn_$1.so $objs ab/emd_$1.o
echo ab/libvpn_$1.so
