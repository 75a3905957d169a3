#!/bin/bash
# bench the other BASELINE configs (kernel-only + step time); CONFIGS="config2 config3 ..."
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for w in ${CONFIGS:-config2 config3 config4}; do
  python bench.py --workload $w --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_$w.json
  python - "$w" <<'PY'
import json,sys
w=sys.argv[1]
try:
    d=json.loads(open(f'gpurun_out/bench_{w}.json').read())
    print(w, 'Mpix/s', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'setup_s', d['config']['setup_s'], d['roofline']['kernel'])
except Exception as e:
    print(w, 'FAILED', open(f'gpurun_out/bench_{w}.json').read()[-400:])
PY
done
