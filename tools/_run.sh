cd /root/repo
python bench.py --steps 20 --warmup 3 --cpu-seconds 2 | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(json.dumps(r["config"]["host_boundary"]), r["cpu_baseline"]["gpu_rows_bit_identical"])'
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cpp_dispatch.py -m gpu -x -q 2>&1 | tail -3
