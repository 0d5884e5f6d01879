cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for d in _old .; do
  echo "== $d"; (cd $d && python bench.py --steps 100 --warmup 20 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['hip_event_ms_per_step']['median'], {k:v['avg_us'] for k,v in d['kernel_us'].items()})")
done; done
