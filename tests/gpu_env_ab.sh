#!/bin/bash
# A/B of environment switches on bench workloads: VARIANTS="EU_HIP_PLANAR=0 EU_HIP_PLANAR=1" WORKLOADS="headline config3"
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
make -s -C oracle _build/libeu_oracle.so
for kv in ${VARIANTS}; do
  for w in ${WORKLOADS:-headline}; do
    f="gpurun_out/envab_${kv//\//_}_${w}.json"
    env $kv python bench.py --workload $w --steps ${STEPS:-20} --no-cpu-baseline 2>&1 | tail -1 > "$f"
    python - "$kv" "$w" "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[3]).read())
print(sys.argv[1], sys.argv[2], "kernel_ms", d['roofline']['kernel_ms'], "ms/step", d['ms_per_step'], "frac", d['roofline']['frac'])
PY
  done
done
