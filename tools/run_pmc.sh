#!/bin/bash
# usage: tools/run_pmc.sh <tag> <pmc counters...>   (run on the GPU box from the repo root)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
echo "$tag rc=$?"
