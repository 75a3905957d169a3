cd /root/repo
for l in envutil_amd/lib/libeu_hip.so envutil_amd/build/libeu_hip_ballot.so; do
echo "== $l"
for i in 1 2; do EU_HIP_LIB=$l timeout 600 python bench.py --workload config3 --steps 30 --warmup 3 --cpu-seconds 2 | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["roofline"]["kernel_ms"], r["cpu_baseline"]["gpu_rows_bit_identical"])'; done
done
export EU_HIP_LIB=envutil_amd/build/libeu_hip_ballot.so
TAG=r02_ballot_config3 BENCH_ARGS="--workload config3" PMC_GROUPS="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU TCP_TOTAL_CACHE_ACCESSES_sum" bash tools/gpu_prof.sh | grep "render4s"
