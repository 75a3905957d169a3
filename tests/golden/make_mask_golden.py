"""tests/golden/mask_golden.npz: what the REFERENCE's alpha_masking_t (masking.h:95-135, compiled in place into
oracle/_ref/libref_zimt.so) yields for the cases of tests/mask_cases.py - inputs (core, degree, coordinates) and
the reference's outputs for paint 0 and 1. Build container only: python tests/golden/make_mask_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import mask_cases  # noqa: E402
import refz  # noqa: E402

out = {}
for name, core, deg, crd in mask_cases.cases():
    r = refz.RefSpline(core, deg, 3, 3)           # REFLECT on both axes, as a facet image is loaded
    r.prefilter(deg)
    out[name + "_core"] = core
    out[name + "_deg"] = np.int32(deg)
    out[name + "_crd"] = crd
    for paint in (0.0, 1.0):
        out[f"{name}_out{int(paint)}"] = refz.alpha_masking(r, paint, crd)
for nch in (1, 2, 3, 4):
    for paint in (0.0, 1.0):
        out[f"masking_{nch}_{int(paint)}"] = refz.masking(nch, paint, 37)
np.savez_compressed(os.path.join(HERE, "mask_golden.npz"), **out)
print("wrote mask_golden.npz:", len(out), "arrays")
