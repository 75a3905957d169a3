#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
make -s -C oracle _build/libeu_oracle.so
python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.log
for w in headline probe_smallsrc probe_bilinear; do
  EU_HIP_DIRECT=1 python bench.py --workload $w --steps 10 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/ab_${w}_direct.json
  python bench.py --workload $w --steps 10 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/ab_${w}_lds.json
done
python bench.py --steps 20 2>&1 | tail -1 > gpurun_out/bench_headline.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_*.json'))+['gpurun_out/bench_headline.json']:
    d=json.loads(open(f).read()); print(f, d['roofline']['kernel_ms'], d['value'], d['roofline']['frac'], d.get('cpu_baseline',{}).get('gpu_rows_bit_identical'))
PY
