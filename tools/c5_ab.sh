#!/bin/bash
# usage (GPU box): tools/c5_ab.sh lib1 lib2 ...  - config 5 step time of each library variant, with and without the early-miss tables
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for l in "$@"; do
  if [ "$l" = default ]; then unset EU_HIP_LIB; else export EU_HIP_LIB=$PWD/envutil_amd/build/libeu_hip_$l.so; fi
  for r in 0 1; do
    echo -n "$l REJ=$r: "
    EU_HIP_REJ=$r timeout -k 10 200 python bench.py --workload config5 --no-cpu-baseline 2>&1 | grep -o "\"kernel_ms\": [0-9.]*" | tr "\n" " "; echo
  done
done
