#!/bin/bash
# first-contact script for a GPU box: build oracle, run GPU tests
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
make -s -C oracle _build/libeu_oracle.so
python -m pytest tests -m gpu -q 2>&1 | tee gpurun_out/pytest_gpu.log
