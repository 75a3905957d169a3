#!/bin/bash
# env-variable sweeps of the headline bench: SWEEP="VAR=a VAR=b ..."
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for kv in ${SWEEP}; do
  env $kv python bench.py --steps 10 --no-cpu-baseline ${BENCH_ARGS} 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$kv', d['config']['name'], 'kernel_ms', d['roofline']['kernel_ms'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'])"
done
