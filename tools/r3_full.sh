#!/bin/bash
# the whole GPU test suite, then the bench line of every workload
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${TAG:-r3_full}; mkdir -p gpurun_out/$T
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/$T/tests.log 2>&1; echo "tests rc $?" | tee gpurun_out/$T/summary.txt
tail -3 gpurun_out/$T/tests.log | tee -a gpurun_out/$T/summary.txt
for w in headline config2 config3 config4 config5; do
  timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/$T/bench_$w.json
  python - "$w" gpurun_out/$T/bench_$w.json <<'PY' | tee -a gpurun_out/$T/summary.txt
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]
print(sys.argv[1], "ms/step", d["ms_per_step"], "kernel_ms", r["kernel_ms"], "frac", r["frac"], "launches", r.get("launches_per_step"))
PY
done
