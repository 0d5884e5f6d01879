#!/bin/bash
# usage: tools/run_trace_emd.sh <tag>   (GPU box, repo root): rocprofv3 kernel-trace + stats of tools/time_emd.py
tag=$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/tools/time_emd.py > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
echo "$tag rc=$?"
cat $GRAFT_REPO_ROOT/gpurun_out/$tag.log | grep "B="
