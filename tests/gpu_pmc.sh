#!/bin/bash
# PMC passes over a bench workload (counters only, one group per pass).
# usage: PMC_GROUPS="A B|C D" BENCH_ARGS="--workload headline" bash tests/gpu_pmc.sh
set -e
R="$GRAFT_REPO_ROOT"
cd "$R"; rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
DEFAULT="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES|SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS|FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum|TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum|GRBM_GUI_ACTIVE GRBM_COUNT"
GROUPS_="${PMC_GROUPS:-$DEFAULT}"
i=0
IFS='|' read -ra GS <<< "$GROUPS_"
for ctrs in "${GS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$R/gpurun_out/pmc/p$i" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > "$R/gpurun_out/pmc/p$i.log" 2>&1 || echo "pass $i failed: $ctrs"
done
cd "$R"
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'eu_render' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])].append(float(r['Counter_Value']))
with open('gpurun_out/pmc/summary.txt','w') as o:
    for k,v in sorted(agg.items()):
        line=f"{k[0]:42s} {k[1]:36s} n={len(v):3d} mean={sum(v)/len(v):.6g}"
        print(line); o.write(line+"\n")
PY
