"""CPU oracle vs fixtures generated from the reference's own zimt headers
(tests/golden/make_golden.py). Bit-exact: float32 results compared as uint32."""
import os

import numpy as np
import pytest

import euo

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "zimt_golden.npz"))
EPS = float(np.finfo(np.float32).eps)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_same(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    assert a.shape == b.shape
    bad = np.argwhere(bits(a) != bits(b))
    assert bad.size == 0, f"{len(bad)} floats differ, first at {bad[0]}: {a[tuple(bad[0])]!r} vs {b[tuple(bad[0])]!r}"


@pytest.mark.parametrize("degree", range(10))
def test_basis_weights(degree):
    for i, d in enumerate(G["weights_deltas"]):
        assert_same(euo.basis_weights(degree, float(d)), G[f"weights_d{degree}"][i])


@pytest.mark.parametrize("degree", range(2, 10))
def test_poles(degree):
    got = euo.poles(degree).astype(np.float64)
    ref = G[f"poles_d{degree}"]
    assert got.shape == ref.shape
    assert np.all(got.astype(np.float32) == ref.astype(np.float32))
    assert np.max(np.abs(got - ref) / np.abs(ref)) < 1e-15


@pytest.mark.parametrize("k", range(len(G["spl_cases"])))
def test_prefilter_brace_eval(k):
    w, h, n, deg, pdeg, b0, b1 = (int(v) for v in G["spl_cases"][k])
    s = euo.BSpline(G[f"spl{k}_core"], deg, b0, b1)
    g = G[f"spl{k}_geometry"]
    assert list(s.s.shape) == [g[0], g[1]]
    assert list(s.s.left) == [g[4], g[5]] and list(s.s.right) == [g[6], g[7]]
    s.prefilter(pdeg)
    assert_same(s.container, G[f"spl{k}_container"])
    assert_same(s.eval(G[f"spl{k}_crd"]), G[f"spl{k}_val"])


@pytest.mark.parametrize("k", range(len(G["sph_cases"])))
def test_spherical_prefilter(k):
    w, h, n, deg, pdeg = (int(v) for v in G["sph_cases"][k])
    s = euo.BSpline(G[f"sph{k}_core"], deg, euo.PERIODIC, euo.REFLECT)
    s.spherical_prefilter(pdeg)
    assert_same(s.container, G[f"sph{k}_container"])
    assert_same(s.eval(G[f"sph{k}_crd"]), G[f"sph{k}_val"])


@pytest.mark.parametrize("k", range(len(G["nat_cases"])))
def test_natural_section_filter(k):
    w, n, deg = (int(v) for v in G["nat_cases"][k])
    b = G[f"nat{k}_in"].copy()
    for c in range(n):
        euo.lib().euo_filter_lines(b.ctypes.data + 4 * c, w, w * n, w, n,
                                   euo.NATURAL, deg, EPS)
    for c in range(n):
        euo.lib().euo_filter_lines(b.ctypes.data + 4 * c, w, n, w, w * n,
                                   euo.NATURAL, deg, EPS)
    assert_same(b, G[f"nat{k}_out"])


@pytest.mark.parametrize("name,w,h", [("drv_out", 600, 3), ("drv_short", 11, 2)])
def test_driver_raster(name, w, h):
    """zimt::process (segments of 512, vectors of 16, leftover lanes) equals
    per-pixel evaluation: the driver adds no arithmetic of its own."""
    s = euo.BSpline(G["drv_core"], 3, euo.PERIODIC, euo.REFLECT)
    s.prefilter(3)
    a = G["drv_aff"]
    x = np.arange(w, dtype=np.float32) * a[1] + a[0]
    y = np.arange(h, dtype=np.float32) * a[3] + a[2]
    crd = np.stack(np.broadcast_arrays(x[None, :], y[:, None]), -1).reshape(-1, 2)
    assert_same(s.eval(crd).reshape(h, w, 3), G[name])


def test_screen_lut_knots_and_eval():
    """to_screen_t: the oracle's sRGB LUT and its clamp + linear evaluation
    against zimt's bspline<float,1> / make_safe_evaluator (fixture)"""
    import ctypes as C
    L = euo.lib()
    lut = np.zeros(257, np.float32)
    L.euo_screen_lut(lut.ctypes.data_as(C.c_void_p))
    assert_same(lut[:256], G["lut_knots"])
    L.euo_lut_eval.restype = C.c_float
    L.euo_lut_eval.argtypes = [C.c_void_p, C.c_float]
    got = np.array([L.euo_lut_eval(lut.ctypes.data, float(v)) for v in G["lut_in"]], np.float32)
    assert_same(got, G["lut_out"])


def test_to_screen_packing():
    """channel order of the packed sRGBA8 word (envutil_payload.cc:357-410)"""
    import ctypes as C
    L = euo.lib()
    lut = np.zeros(257, np.float32)
    L.euo_screen_lut(lut.ctypes.data_as(C.c_void_p))
    L.euo_to_screen.restype = C.c_uint
    L.euo_to_screen.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    px = np.array([1.0, 0.0, 0.2140411, 0.5], np.float32)
    def ts(n):
        return L.euo_to_screen(lut.ctypes.data, n, px.ctypes.data)
    assert ts(1) == 0xFFFFFFFF
    assert ts(2) == 0x00FFFFFF
    w3 = ts(3)
    assert (w3 >> 24) == 0xFF and (w3 & 0xFF) == 255 and ((w3 >> 8) & 0xFF) == 0
    assert ((w3 >> 16) & 0xFF) in (126, 127, 128)      # sRGB(0.214) ~ 0.5
    w4 = ts(4)
    assert (w4 >> 24) in (186, 187, 188) and (w4 & 0xFFFFFF) == (w3 & 0xFFFFFF)


def test_lens_polynomial_against_reference_fixture():
    """oracle's lcp<float>::eval (planar_lens' radial factor, config 5) against outputs of the
    reference's own lens_correction.h (compiled in place, tests/golden/make_golden.py)"""
    G2 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lcp_golden.npz"))
    for abc, ref in zip(G2["lcp_abc"], G2["lcp_out"]):
        got = euo.lens_factor(*abc, G2["lcp_x"])
        assert (got.view(np.uint32) == ref.view(np.uint32)).all(), abc


def test_inverse_lens_polynomial_against_reference_fixture():
    """oracle's inverse_lcp (the spline model of the inverse radial factor, --single jobs) against the
    reference's own class (lens_correction.h:236-301, compiled in place by tests/golden/make_golden.py):
    the model's prefiltered knots and its factors, bit for bit"""
    G3 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inverse_lcp_golden.npz"))
    for s4, ref, knots in zip(G3["sets"], G3["out"], G3["knots"]):
        got, k = euo.inverse_lcp(*s4[:4], int(s4[4]), G3["x"])
        assert (k.view(np.uint32) == knots[:len(k)].view(np.uint32)).all(), s4
        assert (got.view(np.uint32) == ref.view(np.uint32)).all(), s4
