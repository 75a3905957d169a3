#!/bin/bash
# kernel stats + PMC of every bench workload at the current commit (run on the GPU box; one gpurun call per
# few workloads: each takes ~1-2 minutes). usage: COMMIT=<hash> WLS="headline config2 ..." bash tools/prof_all.sh
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
WLS="${WLS:-headline config2 config3 config4 config5}"
PMCG="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES|SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE|GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum|FETCH_SIZE|WRITE_SIZE"
for w in $WLS; do
  TAG=r03_prof_$w BENCH_ARGS="--workload $w" PMC_GROUPS="$PMCG" bash tools/gpu_prof.sh > /dev/null 2>&1
  echo "== $w"; cat gpurun_out/r03_prof_$w/progress.txt; grep -c . gpurun_out/r03_prof_$w/summary.txt
done
