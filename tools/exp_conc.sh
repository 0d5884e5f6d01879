cd $GRAFT_REPO_ROOT
for v in "-DR_TOTAL_WAVES=5" "-DR_TOTAL_WAVES=4" "-DR_TOTAL_WAVES=3"; do
  VPN_EXTRA_FLAGS="$v" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
  for c in 0 1; do
    VPN_CONCURRENT=$c python bench.py --no-extras > gpurun_out/r3r_tmp.json 2>/dev/null
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r3r_tmp.json") if l.startswith("{")][-1])
print("$v conc=$c", d["value"], d["hip_event_ms_per_step"]["median"])
PY
  done
done
VPN_EXTRA_FLAGS="" python volumetric-primitives-net_amd/build.py --force > /dev/null 2>&1
