"""one-off wider sweep of the randomised differential tests (not collected by
pytest: no test_ prefix): python tests/fuzz_wide.py FIRST LAST"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import test_gpu_fuzz as F

first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, last):
    for fn in (F.test_random_jobs_bit_identical, F.test_random_multi_facet_jobs_bit_identical,
               F.test_random_device_setup_bit_identical):
        try:
            fn(seed)
        except AssertionError as e:
            bad += 1
            print("FAIL", fn.__name__, seed, str(e)[:400], flush=True)
    if seed % 5 == 0:
        print("seed", seed, "done, failures so far", bad, flush=True)
print("failures", bad)
sys.exit(1 if bad else 0)
