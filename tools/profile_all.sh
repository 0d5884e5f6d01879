#!/bin/bash
# usage: tools/profile_all.sh <tag>    (GPU box, repo root)
# One kernel-trace + stats pass and the PMC passes (separate runs, --kernel-trace only beside --pmc) of bench.py for
# the C3 headline workload, the C2 raster-only workload and the C5 training step (fused node).  Output: gpurun_out/<tag>_{c3,c2}_{stats,pmcN}/ ;
# tools/pmc_collect.py turns them into profiles/<tag>_pmc.json + text summaries.
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() {   # name, bench args..., then `--` and rocprof args
  local name=$1; shift
  local bargs=(); while [ "$1" != "--" ]; do bargs+=("$1"); shift; done; shift
  rocprofv3 "$@" --output-format csv -d $R/gpurun_out/${tag}_$name -- python3 $R/bench.py --steps 10 --warmup 3 --no-extras --no-graph "${bargs[@]}" > $R/gpurun_out/${tag}_$name.log 2>&1
  echo "${tag}_$name rc=$?"
}
for wl in c3 c2 c5; do
  extra=""; if [ $wl = c5 ]; then extra="--c5-form fused --no-cpu-baseline"; fi
  run ${wl}_stats --workload $wl $extra -- --kernel-trace --stats || exit 1
  run ${wl}_pmc1 --workload $wl $extra -- --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE || exit 1
  run ${wl}_pmc2 --workload $wl $extra -- --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS || exit 1
  run ${wl}_pmc3 --workload $wl $extra -- --kernel-trace --pmc FETCH_SIZE || exit 1
  run ${wl}_pmc4 --workload $wl $extra -- --kernel-trace --pmc WRITE_SIZE || exit 1
done
