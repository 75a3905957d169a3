#!/bin/bash
# usage (GPU box): tools/wl_ab.sh "workload ..." lib1 lib2 ...  - bench.py's kernel time of each workload under each library variant, twice
cd "${GRAFT_REPO_ROOT:-/root/repo}"
WLS=$1; shift
for round in 1 2; do
for wl in $WLS; do
for l in "$@"; do
  if [ "$l" = default ]; then unset EU_HIP_LIB; else export EU_HIP_LIB=$PWD/envutil_amd/build/libeu_hip_$l.so; fi
  echo -n "$wl $l: "
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline 2>&1 | grep -o "\"kernel_ms\": [0-9.]*" | tr "\n" " "; echo
done
done
done
