cd /root/repo
TAG=r02_prefilter bash tools/gpu_prof.sh | grep "filter_\|brace\|pole_rows"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "device or coeff or setup or prefilter or fuzz" 2>&1 | tail -3
