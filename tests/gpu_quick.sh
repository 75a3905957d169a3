#!/bin/bash
# quick loop: packed-kernel parity tests + headline bench (+ optional env sweep)
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
make -s -C oracle _build/libeu_oracle.so
python -m pytest tests/test_gpu_math.py -m gpu -q -x 2>&1 | tail -3
for g in ${GRIDS:-default}; do
  if [ "$g" = default ]; then unset EU_HIP_GRID; else export EU_HIP_GRID=$g; fi
  python bench.py --steps 10 --no-cpu-baseline ${BENCH_ARGS} 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('grid','$g', d['config']['name'], 'kernel_ms', d['roofline']['kernel_ms'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'])"
done
