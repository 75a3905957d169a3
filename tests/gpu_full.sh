#!/bin/bash
# full GPU pass: all gpu tests, smoke, headline bench + rocprofv3 kernel trace of the same command
set -e
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
make -s -C oracle _build/libeu_oracle.so
python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py --steps 20 --warmup 3 2>&1 | tail -1 > gpurun_out/bench_headline.json
cat gpurun_out/bench_headline.json
export TMPDIR=/tmp
rm -rf gpurun_out/prof_headline
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_headline" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/prof_headline.log" 2>&1)
head -4 gpurun_out/prof_headline/*/*kernel_stats.csv | cut -c1-160
