#!/bin/bash
# A/B of two builds of the library on one box, alternating: usage  bash tools/ab_lib.sh TAG LIB_A LIB_B [workloads...]
# bench lines -> gpurun_out/ab_<TAG>_<workload>_<A|B>_<i>.json, summary on stdout
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
TAG=$1; LA=$2; LB=$3; shift 3
WL=${@:-headline config2}
for w in $WL; do
  for i in 1 2 3; do
    EU_HIP_LIB=$LA timeout -k 10 200 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_${TAG}_${w}_A_$i.json || exit 1
    EU_HIP_LIB=$LB timeout -k 10 200 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_${TAG}_${w}_B_$i.json || exit 1
  done
done
python3 - "$TAG" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(f"gpurun_out/ab_{sys.argv[1]}_*.json")):
    d = json.loads(open(f).read().strip().split("\n")[-1])
    print(f.split("/")[-1], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"])
PY
