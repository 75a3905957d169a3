#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${TAG:-r3_3}
mkdir -p gpurun_out/$T
EU_HIP_DEBUG=1 timeout -k 10 300 python tools/ab_env.py headline "EU_HIP_R4=0" "EU_HIP_R4=1" 2>&1 | grep -v amdgpu.ids | sort | uniq -c | tee gpurun_out/$T/summary.txt
timeout -k 10 300 python tools/ab_env.py config3 "EU_HIP_R4=0" "EU_HIP_R4=1" "EU_HIP_R4=1 EU_HIP_R5=0" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/$T/summary.txt
timeout -k 10 500 python -m pytest tests/test_gpu_staged.py -x -q -m gpu 2>&1 | tail -2 | tee -a gpurun_out/$T/summary.txt
TAG=${T}_p WL=headline ENVS="EU_HIP_R4=1" PMC="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES|SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" bash tools/prof_env.sh > /dev/null 2>&1
grep -v "colplan\|copyBuffer\|at::native\|filter_\|verify_" gpurun_out/${T}_p/summary.txt
