"""Generates tests/golden/zimt_*.npz from the reference's own zimt headers,
compiled in place as oracle/_ref/libref_zimt.so (oracle/Makefile target 'ref').
Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

The files hold inputs and the reference's outputs (data only); they pin the
CPU oracle's zimt stages on machines where /root/reference is absent.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import refz  # noqa: E402

EPS = float(np.finfo(np.float32).eps)


def coords(rng, w, h, n):
    c = np.stack([rng.uniform(-2.5 * w, 3.5 * w, n),
                  rng.uniform(-2.5 * h, 3.5 * h, n)], 1).astype(np.float32)
    m = n // 4
    c[:m] = np.stack([rng.uniform(-1, w, m), rng.uniform(-1, h, m)], 1)
    c[m:m + 6] = [[-0.5, -0.5], [w - 0.5, h - 0.5], [w - 1, h - 1], [0, 0],
                  [-0.0, 0.5], [w / 2, -0.5]]
    return c


def main():
    rng = np.random.default_rng(20251226)
    out = {}
    # basis weights, degrees 0..9
    deltas = np.array([0.0, 0.25, 0.5, 0.75, 0.999, -0.3, -0.5, 0.123456], np.float32)
    for d in range(10):
        out[f"weights_d{d}"] = np.stack([refz.basis_weights(d, float(x)) for x in deltas])
        if d >= 2:
            out[f"poles_d{d}"] = refz.poles(d).astype(np.float64)
    out["weights_deltas"] = deltas
    # ordinary splines: container after prefilter + brace, evaluation
    cases = []
    for (w, h, n) in [(24, 12, 3), (7, 5, 1), (20, 9, 4)]:
        core = rng.random((h, w, n), dtype=np.float32)
        for deg, pdeg, b0, b1 in [(1, 1, 1, 2), (3, 3, 1, 2), (3, 3, 2, 2),
                                  (2, 2, 2, 2), (0, 0, 0, 0), (5, 5, 3, 3),
                                  (4, 3, 1, 1), (1, 3, 2, 2)]:
            r = refz.RefSpline(core, deg, b0, b1)
            r.prefilter(pdeg)
            crd = coords(rng, w, h, 96)
            k = len(cases)
            cases.append((w, h, n, deg, pdeg, b0, b1))
            out[f"spl{k}_core"] = core
            out[f"spl{k}_container"] = r.container()
            out[f"spl{k}_geometry"] = np.array(r.geometry(), np.int64)
            out[f"spl{k}_crd"] = crd
            out[f"spl{k}_val"] = r.eval(crd)
    out["spl_cases"] = np.array(cases, np.int64)
    # full-sphere lat/lon sources
    cases = []
    for (w, h, n) in [(32, 16, 3), (16, 8, 4)]:
        core = rng.random((h, w, n), dtype=np.float32)
        for deg, pdeg in [(1, 1), (3, 3), (2, 2), (5, 5), (1, 3)]:
            r = refz.RefSpline(core, deg, 1, 2)
            r.spherical(pdeg)
            crd = coords(rng, w, h, 64)
            k = len(cases)
            cases.append((w, h, n, deg, pdeg))
            out[f"sph{k}_core"] = core
            out[f"sph{k}_container"] = r.container()
            out[f"sph{k}_crd"] = crd
            out[f"sph{k}_val"] = r.eval(crd)
    out["sph_cases"] = np.array(cases, np.int64)
    # frameless NATURAL x NATURAL filter (cubemap IR sections)
    cases = []
    for (w, n, deg) in [(32, 3, 3), (20, 4, 2), (9, 1, 5)]:
        img = rng.random((w, w, n), dtype=np.float32)
        k = len(cases)
        cases.append((w, n, deg))
        out[f"nat{k}_in"] = img
        out[f"nat{k}_out"] = refz.filter_2d(img, deg, 3, 3)
    out["nat_cases"] = np.array(cases, np.int64)
    # the strip-mining driver: 600 x 3 raster (one full 512 segment, a short
    # one with a leftover), affine coordinates into a cubic spline
    core = rng.random((12, 24, 3), dtype=np.float32)
    r = refz.RefSpline(core, 3, 1, 2)
    r.prefilter(3)
    aff = np.array([-3.25, 0.0517, -1.5, 4.875], np.float32)
    out["drv_core"] = core
    out["drv_aff"] = aff
    out["drv_out"] = r.process_affine(600, 3, aff)
    out["drv_short"] = r.process_affine(11, 2, aff)
    # to_screen_t's LUT evaluation (envutil_payload.cc:251-287): 1-D degree-1
    # NATURAL spline over 255 * sRGB(i / 255), clamp gate, at in * 255.0f
    x = np.arange(256) / 255.0
    y = np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 0.41666666666666667) - 0.055) * 255.0
    knots = y.astype(np.float32)
    one = np.float32(1.0)
    vin = np.concatenate([
        np.linspace(-0.05, 1.05, 6000).astype(np.float32),
        (np.arange(256) / 255.0).astype(np.float32),
        rng.random(6000, dtype=np.float32),
        np.array([0.0, -0.0, 1.0, 1e-8, 0.0031308, np.nextafter(one, np.float32(0)),
                  np.nextafter(one, np.float32(2)), 0.5, 254.5 / 255.0, 2.0, -3.0, 1e30], np.float32)])
    out["lut_knots"] = knots
    out["lut_in"] = vin
    out["lut_out"] = refz.lut_eval(knots, vin)
    # the PTO lens polynomial (lens_correction.h compiles in place: it includes only
    # zimt/eval.h): radial factor for r / s in [0, 2.5], several coefficient sets incl. the
    # one of BASELINE config 5; its own file
    lx = np.concatenate([np.linspace(0.0, 2.5, 2000).astype(np.float32),
                         rng.random(2000, dtype=np.float32) * np.float32(1.5),
                         np.array([0.0, 1.0, 0.5, 1e-20, 3.0], np.float32)])
    sets = np.array([[0.01, -0.03, 0.02], [0.0, 0.0, 0.05], [-0.12, 0.3, -0.21],
                     [1e-3, 0.0, 0.0], [0.25, 0.25, 0.25]], np.float64)
    lcp = {"lcp_x": lx, "lcp_abc": sets,
           "lcp_out": np.stack([refz.lcp_factor(*abc, lx) for abc in sets])}
    np.savez_compressed(os.path.join(HERE, "lcp_golden.npz"), **lcp)
    print("wrote", os.path.join(HERE, "lcp_golden.npz"))
    # the inverse of that polynomial as pto_planar<T, L, true> builds it (sz = 100; --single jobs):
    # factors and the model's prefiltered knots; its own file
    ix = np.concatenate([np.linspace(0.0, 2.6, 1500).astype(np.float32), rng.random(1500, dtype=np.float32) * np.float32(2.0),
                         np.array([0.0, 1e-8, 1.0, 5.0], np.float32)])
    isets = np.array([[0.01, -0.03, 0.02, 1.8027756377319946, 100], [0.0, 0.0, 0.05, 1.2, 100], [-0.02, 0.01, 0.0, 2.3, 100],
                      [0.001, 0.002, -0.004, 1.5, 32], [0.02, 0.0, 0.0, 1.8, 100]], np.float64)
    res = [refz.inverse_lcp(*s4[:4], int(s4[4]), ix) for s4 in isets]
    np.savez_compressed(os.path.join(HERE, "inverse_lcp_golden.npz"), x=ix, sets=isets,
                        out=np.stack([r[0] for r in res]), knots=np.stack([np.pad(r[1], (0, 104 - len(r[1]))) for r in res]))
    print("wrote", os.path.join(HERE, "inverse_lcp_golden.npz"))
    np.savez_compressed(os.path.join(HERE, "zimt_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "zimt_golden.npz"),
          os.path.getsize(os.path.join(HERE, "zimt_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
