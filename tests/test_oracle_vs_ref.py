"""CPU oracle vs the reference's own zimt headers compiled in place
(oracle/_ref/libref_zimt.so): a wider sweep than the committed fixtures. Runs
only where /root/reference exists (the build container); skipped elsewhere."""
import numpy as np
import pytest

import euo
import refz

pytestmark = [pytest.mark.ref,
              pytest.mark.skipif(not refz.available(),
                                 reason="oracle/_ref not built (no /root/reference here)")]


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_lanes_and_segment():
    assert refz.lib().ref_lanes() == 16
    assert refz.lib().ref_segment() == 512


@pytest.mark.parametrize("shape", [(24, 12, 3), (40, 17, 4), (7, 5, 1), (64, 32, 2)])
def test_prefilter_brace_eval_sweep(shape):
    w, h, n = shape
    rng = np.random.default_rng(w * 1000 + h)
    core = rng.random((h, w, n), dtype=np.float32)
    for deg in range(0, 8):
        for pdeg in sorted({deg, 0, 3}):
            for b0, b1 in [(1, 2), (2, 2), (0, 0), (3, 3), (1, 1), (2, 3)]:
                r = refz.RefSpline(core, deg, b0, b1)
                o = euo.BSpline(core, deg, b0, b1)
                r.prefilter(pdeg)
                o.prefilter(pdeg)
                assert (bits(o.container) == bits(r.container())).all(), (deg, pdeg, b0, b1)
                crd = np.stack([rng.uniform(-2.5 * w, 3.5 * w, 256),
                                rng.uniform(-2.5 * h, 3.5 * h, 256)], 1).astype(np.float32)
                crd[:64] = np.stack([rng.uniform(-1, w, 64), rng.uniform(-1, h, 64)], 1)
                assert (bits(o.eval(crd)) == bits(r.eval(crd))).all(), (deg, pdeg, b0, b1)


@pytest.mark.parametrize("shape", [(32, 16, 3), (64, 32, 4), (16, 8, 1), (128, 64, 3)])
def test_spherical_prefilter_sweep(shape):
    w, h, n = shape
    rng = np.random.default_rng(w)
    core = rng.random((h, w, n), dtype=np.float32)
    for deg in range(0, 6):
        for pdeg in sorted({deg, 3, 1}):
            r = refz.RefSpline(core, deg, 1, 2)
            o = euo.BSpline(core, deg, euo.PERIODIC, euo.REFLECT)
            r.spherical(pdeg)
            o.spherical_prefilter(pdeg)
            assert (bits(o.container) == bits(r.container())).all(), (deg, pdeg)


def test_weights_and_poles():
    for d in range(10):
        for delta in np.linspace(-0.5, 1.0, 31, dtype=np.float32):
            assert (bits(euo.basis_weights(d, float(delta)))
                    == bits(refz.basis_weights(d, float(delta)))).all()
        if d >= 2:
            assert (euo.poles(d).astype(np.float32)
                    == refz.poles(d).astype(np.float32)).all()


def test_lens_polynomial_sweep():
    """lcp<float>::eval of the oracle against the reference's lens_correction.h, compiled in
    place: random coefficient triples, radii over [0, 4] and all floats of a few binades"""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.random(20000, dtype=np.float32) * np.float32(4.0),
                        np.arange(0x3f000000, 0x3f000000 + 50000, dtype=np.uint32).view(np.float32),
                        np.arange(0x3f800000 - 25000, 0x3f800000 + 25000, dtype=np.uint32).view(np.float32)])
    for _ in range(40):
        a, b, c = (rng.uniform(-0.4, 0.4) for _ in range(3))
        assert (bits(euo.lens_factor(a, b, c, x)) == bits(refz.lcp_factor(a, b, c, x))).all(), (a, b, c)


def test_inverse_lens_polynomial_sweep():
    """inverse_lcp of the oracle against the reference's class for random mild coefficient triples"""
    rng = np.random.default_rng(9)
    x = np.concatenate([rng.random(5000, dtype=np.float32) * np.float32(2.5), np.float32([0.0, 1e-9, 1.0, 7.0])])
    for _ in range(25):
        a, b, c = (rng.uniform(-0.02, 0.02) for _ in range(3))
        r_max = rng.uniform(1.1, 2.0)
        for sz in (32, 100):
            o, k = euo.inverse_lcp(a, b, c, r_max, sz, x)
            ro, rk = refz.inverse_lcp(a, b, c, r_max, sz, x)
            assert (bits(k) == bits(rk)).all() and (bits(o) == bits(ro)).all(), (a, b, c, r_max, sz)
