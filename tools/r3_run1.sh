#!/bin/bash
# first GPU run of round 3: staged-kernel parity tests, then A/B of the kernels at full size
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r3_1
timeout -k 10 500 python -m pytest tests/test_gpu_staged.py -x -q -m gpu > gpurun_out/r3_1/staged.log 2>&1; echo "staged rc $?" | tee -a gpurun_out/r3_1/summary.txt
tail -3 gpurun_out/r3_1/staged.log | tee -a gpurun_out/r3_1/summary.txt
timeout -k 10 300 python tools/ab_env.py headline "EU_HIP_R4=0" "EU_HIP_R4=1" "EU_HIP_R4=1 EU_HIP_R5=0" "EU_HIP_R4=1 EU_HIP_R5_WGS=4" 2>&1 | tee -a gpurun_out/r3_1/summary.txt
timeout -k 10 300 python tools/ab_env.py config3 "EU_HIP_R4=0" "EU_HIP_R4=1" "EU_HIP_R4=1 EU_HIP_R5=0" 2>&1 | tee -a gpurun_out/r3_1/summary.txt
