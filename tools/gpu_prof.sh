#!/bin/bash
# Per-kernel times (rocprofv3 --kernel-trace --stats) and PMC passes of one bench workload.
# usage (on the GPU box):
#   TAG=r03_headline BENCH_ARGS="--workload headline" [PMC_GROUPS="A B|C"] bash tools/gpu_prof.sh
# Every '|'-separated group is one rocprofv3 run. FETCH_SIZE and WRITE_SIZE are derived from the
# L2's memory-side request counters and each fills the TCC block's slots on its own
# (MI355X_MICROARCH.md, "rocprofv3 PMC slots": FETCH_SIZE 3 of 4, WRITE_SIZE 2 of 4): a group that
# names more than one TCC-derived counter is SPLIT into one pass per such counter instead of being
# handed to rocprofv3 (which aborts inside the first HIP call with "error code 38: Request exceeds
# the capabilities of the hardware to collect" - three aborted runs in round 2).
set -e
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${TAG:-prof}"
OUT="$R/gpurun_out/$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > "$OUT/trace.log" 2>&1 || echo "trace failed" | tee -a "$OUT/progress.txt"

# split the groups: at most one TCC-derived counter per pass
PASSES=()
IFS='|' read -ra GS <<< "${PMC_GROUPS}"
for ctrs in "${GS[@]}"; do
  tcc=(); rest=()
  for c in $ctrs; do
    case $c in
      FETCH_SIZE|WRITE_SIZE|TCC_*) tcc+=("$c");;
      *) rest+=("$c");;
    esac
  done
  if [ ${#tcc[@]} -le 1 ]; then
    PASSES+=("$ctrs")
  else
    echo "group '$ctrs' names ${#tcc[@]} TCC-derived counters: split into one pass each" | tee -a "$OUT/progress.txt"
    first=1
    for c in "${tcc[@]}"; do
      if [ $first = 1 ] && [ ${#rest[@]} -gt 0 ]; then PASSES+=("${rest[*]} $c"); else PASSES+=("$c"); fi
      first=0
    done
  fi
done
i=0
for ctrs in "${PASSES[@]}"; do
  i=$((i+1))
  echo "pass $i: $ctrs" >> "$OUT/progress.txt"
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/p$i" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > "$OUT/p$i.log" 2>&1 || echo "pass $i failed: $ctrs" | tee -a "$OUT/progress.txt"
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
            if 'eu_' in r['Name'] or 'filter_' in r['Name'] or float(r['Percentage']) >= 1.0:
                emit(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {r['TotalDurationNs']:>12s} {float(r['AverageNs']):12.1f} {r['Percentage']:>6s}")
    agg = collections.defaultdict(list)
    for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'eu_render' in r['Kernel_Name']:
                agg[(r['Kernel_Name'].split('(')[0][-60:], r['Counter_Name'])].append(float(r['Counter_Value']))
    if agg: emit("# counters: mean per dispatch")
    for k, v in sorted(agg.items()):
        emit(f"{k[0]:62s} {k[1]:34s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
PY
