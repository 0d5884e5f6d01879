#!/bin/bash
# usage: tools/run_trace_py.sh <tag> <script.py> [args]   (GPU box, repo root): rocprofv3 kernel-trace + stats of a python tool
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/"$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
echo "$tag rc=$?"
