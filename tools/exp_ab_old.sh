#!/bin/bash
# same-box A/B of two BUILDS of the step: `_old/` (a copy of an earlier commit, made and built in the container with
#   mkdir _old && git archive <commit> | tar -x -C _old && (cd _old/volumetric-primitives-net_amd && python build.py)
# -- it travels to the GPU box with the snapshot and is not tracked) against the working tree, three rounds each
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for d in _old .; do
  echo "== $d"; (cd $d && python bench.py --steps 100 --warmup 20 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['hip_event_ms_per_step']['median'], {k:v['avg_us'] for k,v in d['kernel_us'].items()})")
done; done
