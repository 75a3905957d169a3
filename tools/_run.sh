cd /root/repo
for wl in config2 config3 headline; do
for r4 in 1 0; do
echo "$wl R4=$r4: $(EU_HIP_R4=$r4 timeout 600 python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["frac"])')"
done; done
