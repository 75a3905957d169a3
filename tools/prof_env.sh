#!/bin/bash
# usage (GPU box): TAG=x WL=headline ENVS="EU_HIP_R4=1" [PMC="SQ_WAVES SQ_INSTS_VALU|SQ_WAVE_CYCLES SQ_WAIT_ANY"] bash tools/prof_env.sh
# kernel trace + optional PMC passes (one rocprofv3 run per '|' group) of tools/ab_env.py under one setting
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${TAG:-prof}"; WL="${WL:-headline}"
OUT="$R/gpurun_out/$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
for kv in $ENVS; do export "$kv"; done
export AB_REPS=10
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/tools/ab_env.py" $WL > "$OUT/trace.log" 2>&1 || echo "trace failed"
i=0
IFS='|' read -ra GS <<< "${PMC}"
for ctrs in "${GS[@]}"; do
  i=$((i+1))
  # FETCH_SIZE / WRITE_SIZE are derived from TCC counters that fill a pass on their own
  n=0; for c in $ctrs; do case $c in FETCH_SIZE|WRITE_SIZE|TCC_*) n=$((n+1));; esac; done
  if [ $n -gt 1 ]; then echo "pass $i REFUSED (more than one TCC-derived counter): $ctrs" | tee -a "$OUT/progress.txt"; continue; fi
  echo "pass $i: $ctrs" >> "$OUT/progress.txt"
  AB_REPS=3 timeout -k 5 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/p$i" -- python3 "$R/tools/ab_env.py" $WL > "$OUT/p$i.log" 2>&1 || echo "pass $i failed: $ctrs" | tee -a "$OUT/progress.txt"
done
cd "$R"
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
with open(out + '/summary.txt', 'w') as o:
    def emit(line):
        print(line); o.write(line + "\n")
    for f in glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True):
        emit("# kernel stats: name, calls, total ns, avg ns, %")
        for r in csv.DictReader(open(f)):
            if float(r['Percentage']) < 0.5: continue
            emit(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {r['TotalDurationNs']:>12s} {float(r['AverageNs']):12.1f} {r['Percentage']:>6s}")
    agg = collections.defaultdict(list)
    for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'eu_render' in r['Kernel_Name']:
                agg[(r['Kernel_Name'].split('(')[0][-60:], r['Counter_Name'], r.get('VGPR_Count',''), r.get('LDS_Block_Size',''))].append(float(r['Counter_Value']))
    if agg: emit("# counters: mean per dispatch (kernel, counter, vgpr, lds)")
    for k, v in sorted(agg.items()):
        emit(f"{k[0]:62s} {k[1]:28s} vgpr {k[2]:>4s} lds {k[3]:>6s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
