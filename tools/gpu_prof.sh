#!/bin/bash
# Per-kernel times (rocprofv3 --kernel-trace --stats) and optional PMC passes of one bench
# workload. usage (on the GPU box): TAG=r02_x BENCH_ARGS="--workload headline" [PMC_GROUPS="A B|C"] bash tools/gpu_prof.sh
set -e
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${TAG:-prof}"
OUT="$R/gpurun_out/$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > "$OUT/trace.log" 2>&1 || echo "trace failed"
i=0
IFS='|' read -ra GS <<< "${PMC_GROUPS}"
for ctrs in "${GS[@]}"; do
  i=$((i+1))
  echo "pass $i: $ctrs" >> "$OUT/progress.txt"
  timeout -k 5 180 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/p$i" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > "$OUT/p$i.log" 2>&1 || echo "pass $i failed: $ctrs"
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
            emit(f"{r['Name'][:90]:90s} {r['Calls']:>6s} {r['TotalDurationNs']:>12s} {float(r['AverageNs']):12.1f} {r['Percentage']:>6s}")
    agg = collections.defaultdict(list)
    for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'eu_' in r['Kernel_Name'] or 'filter_' in r['Kernel_Name']:
                agg[(r['Kernel_Name'].split('(')[0][-46:], r['Counter_Name'])].append(float(r['Counter_Value']))
    if agg: emit("# counters: mean per dispatch")
    for k, v in sorted(agg.items()):
        emit(f"{k[0]:48s} {k[1]:34s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
