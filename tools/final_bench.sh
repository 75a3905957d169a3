#!/bin/bash
# the bench line of every workload (headline with the CPU baseline), one JSON line each -> gpurun_out/final_bench.jsonl
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
: > gpurun_out/final_bench.jsonl
timeout -k 10 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 >> gpurun_out/final_bench.jsonl
for w in config2 config3 config4 config5; do
  timeout -k 10 600 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/final_bench.jsonl
done
python3 - <<'PY'
import json
for l in open("gpurun_out/final_bench.jsonl"):
    if not l.strip(): continue
    d = json.loads(l)
    print(d["config"].get("name"), "ms/step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "value", d["value"], (d.get("cpu_baseline") or {}).get("value"))
PY
