#!/bin/bash
# usage: tools/ab_libs.sh WORKLOAD "ENV SETTINGS" lib1 lib2 ...   (lib = name of envutil_amd/build/libeu_hip_NAME.so, or "default")
cd "${GRAFT_REPO_ROOT:-/root/repo}"
WL=$1; ENVS=$2; shift 2
for round in 1 2; do
for l in "$@"; do
  if [ "$l" = default ]; then unset EU_HIP_LIB; else export EU_HIP_LIB=$PWD/envutil_amd/build/libeu_hip_$l.so; fi
  echo -n "$l: "; timeout -k 10 200 python tools/ab_env.py $WL "$ENVS" 2>&1 | grep kernel_ms
done
done
