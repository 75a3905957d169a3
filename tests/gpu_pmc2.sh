#!/bin/bash
# compare PMC of two env settings on the headline kernel: VARIANTS="A=1 B=2"
set -e
R="$GRAFT_REPO_ROOT"; cd "$R"; rm -rf gpurun_out/pmc2; mkdir -p gpurun_out/pmc2
export TMPDIR=/tmp
GROUPS_="${PMC_GROUPS:-SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU|FETCH_SIZE|TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum|GRBM_GUI_ACTIVE SPI_CSN_BUSY SPI_CSN_WAVE}"
IFS='|' read -ra GS <<< "$GROUPS_"
for kv in ${VARIANTS}; do
  i=0
  for ctrs in "${GS[@]}"; do
    i=$((i+1))
    (cd /tmp && env $kv rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$R/gpurun_out/pmc2/$kv/p$i" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > "$R/gpurun_out/pmc2/$kv.p$i.log" 2>&1) || echo "pass failed $kv $ctrs"
  done
done
python3 - <<'PY'
import csv, glob, collections, os
agg=collections.defaultdict(dict)
for d in sorted(glob.glob('gpurun_out/pmc2/*/')):
    v=os.path.basename(d.rstrip('/'))
    tmp=collections.defaultdict(list)
    for f in glob.glob(d+'p*/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'eu_render' in r['Kernel_Name']:
                tmp[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,x in tmp.items(): agg[k][v]=sum(x)/len(x)
vs=sorted({v for d in agg.values() for v in d})
with open('gpurun_out/pmc2/summary.txt','w') as o:
    hdr="%-34s"%"counter"+"".join("%18s"%v for v in vs); print(hdr); o.write(hdr+"\n")
    for k in sorted(agg):
        line="%-34s"%k+"".join("%18.5g"%agg[k].get(v,float('nan')) for v in vs); print(line); o.write(line+"\n")
PY
