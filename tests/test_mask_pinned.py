"""--mask_for's functors pinned (round 3): the oracle's euo_mask_paint - what oracle/eu_oracle.c's mount_eval puts
in the place of the interpolated pixel - against the REFERENCE's own masking_t and alpha_masking_t
(masking.h:70-135; the header includes nothing but zimt, so it compiles in place into oracle/_ref).
Live where /root/reference exists, and against tests/golden/mask_golden.npz (generated from the same library)
everywhere. The HIP path is compared with the oracle in tests/test_mask_for.py."""
import ctypes as C
import os

import numpy as np
import pytest

import euo
import mask_cases
import refz

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mask_golden.npz")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def oracle_mask(px, mask_paint):
    """euo_mask_paint on every row of px (in place on a copy)"""
    px = np.ascontiguousarray(px, np.float32).copy()
    f = euo.lib().euo_mask_paint
    f.restype = None
    f.argtypes = [C.c_int, C.c_int, C.c_void_p]
    for i in range(px.shape[0]):
        f(px.shape[1], mask_paint, C.c_void_p(px[i].ctypes.data))
    return px


def oracle_alpha_masking(core, deg, crd, paint):
    o = euo.BSpline(core, deg, 3, 3)
    o.prefilter(deg)
    return oracle_mask(o.eval(crd), 2 if paint == 1.0 else 1)


@pytest.mark.parametrize("paint", [0.0, 1.0])
def test_fixture(paint):
    g = np.load(GOLDEN)
    for name, core, deg, crd in mask_cases.cases():
        assert (bits(g[name + "_core"]) == bits(core)).all() and int(g[name + "_deg"]) == deg
        got = oracle_alpha_masking(core, deg, crd, paint)
        assert (bits(got) == bits(g[f"{name}_out{int(paint)}"])).all(), name
    for nch in (1, 2, 3, 4):
        ref = g[f"masking_{nch}_{int(paint)}"]
        if nch in (1, 3):           # facets without alpha: masking_t (the oracle's 1- and 3-channel branch)
            got = oracle_mask(np.random.default_rng(nch).random(ref.shape, dtype=np.float32), 2 if paint else 1)
            assert (bits(got) == bits(ref)).all(), nch
        assert (ref == paint).all()


@pytest.mark.skipif(not refz.available(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("paint", [0.0, 1.0])
def test_live(paint):
    for name, core, deg, crd in mask_cases.cases():
        r = refz.RefSpline(core, deg, 3, 3)
        r.prefilter(deg)
        assert (bits(oracle_alpha_masking(core, deg, crd, paint)) == bits(refz.alpha_masking(r, paint, crd))).all(), name
    rng = np.random.default_rng(7)
    for nch in (1, 3):
        px = rng.random((53, nch), dtype=np.float32)
        assert (bits(oracle_mask(px, 2 if paint else 1)) == bits(refz.masking(nch, paint, 53))).all()
