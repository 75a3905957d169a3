#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
: > gpurun_out/r3_fma_experiment.jsonl
for w in ${WLS:-headline config2 config3 config5 config4}; do
  unset EU_HIP_LIB; timeout -k 10 300 python tools/fma_experiment.py render default $w 2>/dev/null | tail -1 >> gpurun_out/r3_fma_experiment.jsonl
  EU_HIP_LIB=$PWD/envutil_amd/build/libeu_hip_fma.so timeout -k 10 300 python tools/fma_experiment.py render fma $w 2>/dev/null | tail -1 >> gpurun_out/r3_fma_experiment.jsonl
  timeout -k 10 300 python tools/fma_experiment.py compare $w 2>/dev/null | tail -1 >> gpurun_out/r3_fma_experiment.jsonl
done
cat gpurun_out/r3_fma_experiment.jsonl
