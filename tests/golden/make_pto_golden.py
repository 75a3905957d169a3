"""tests/golden/pto_golden.json: the line groups the REFERENCE's pto_parser_type (pto.h, compiled in place into
oracle/_ref/libref_zimt.so) makes of the scripts in tests/pto_cases.py. Run where /root/reference exists:
    python tests/golden/make_pto_golden.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import pto_cases            # noqa: E402
from test_pto_pinned import ref_parse   # noqa: E402

json.dump({k: ref_parse(v) for k, v in sorted(pto_cases.CASES.items())}, open(os.path.join(HERE, "pto_golden.json"), "w"), indent=1)
print("wrote pto_golden.json")
