#!/bin/bash
# GPU box: smoke, bench (headline), rocprofv3 kernel trace of the same command
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
make -s -C oracle _build/libeu_oracle.so
python __graft_entry__.py smoke 2>&1 | tee gpurun_out/smoke.log
python bench.py --workload small --steps 5 --warmup 1 2>&1 | tee gpurun_out/bench_small.log
python bench.py --steps 20 --warmup 3 2>&1 | tee gpurun_out/bench_headline.log
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_headline" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/prof_headline.log" 2>&1
cd "$R"
find gpurun_out/prof_headline -name "*stats*" | head
