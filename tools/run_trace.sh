#!/bin/bash
# usage: tools/run_trace.sh <tag> [bench args]   (GPU box, repo root): kernel-trace + stats of bench.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
echo "$tag rc=$?"
grep '^{' $GRAFT_REPO_ROOT/gpurun_out/$tag.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_us'])"
