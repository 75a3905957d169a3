"""PTO exclude masks and lens crops (environment.h:700-890): the library's host function
eu_hip_facet_alpha against the oracle's loop-for-loop restatement, the oracle's binomial against
the reference's own zimt::convolve (live where /root/reference exists, and on the committed
fixture tests/golden/alpha_golden.npz), and properties written from the definitions.
No GPU needed: this is load-time host code."""
import os

import numpy as np
import pytest

import envutil_amd as ea
import euo
import refz

GOLD = os.path.join(os.path.dirname(__file__), "golden", "alpha_golden.npz")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def random_polygons(rng, w, h, n):
    polys = []
    for _ in range(n):
        k = int(rng.integers(3, 9))
        # vertices also outside the image, self-intersecting orders, integer and fractional coordinates
        x = rng.uniform(-0.3 * w, 1.3 * w, k).astype(np.float32)
        y = rng.uniform(-0.3 * h, 1.3 * h, k).astype(np.float32)
        if rng.random() < 0.3:
            x, y = np.round(x), np.round(y)
        polys.append((x, y))
    return polys


def test_binomial_matches_the_reference_fixture():
    g = np.load(GOLD)
    n = len(g.files) // 2
    assert n >= 6
    for k in range(n):
        got = euo.binomial_plane(g[f"in_{k}"])
        assert (bits(got) == bits(g[f"out_{k}"])).all(), k


@pytest.mark.ref
@pytest.mark.skipif(not refz.available(), reason="oracle/_ref not built (no /root/reference here)")
def test_binomial_matches_zimt_live():
    rng = np.random.default_rng(7)
    for w, h in [(1, 1), (2, 3), (4, 4), (5, 5), (6, 11), (33, 17), (200, 3), (257, 129)]:
        for kind in range(2):
            p = rng.random((h, w), dtype=np.float32) if kind else (rng.random((h, w)) > 0.5).astype(np.float32)
            assert (bits(euo.binomial_plane(p)) == bits(refz.binomial_alpha(p))).all(), (w, h, kind)


@pytest.mark.parametrize("seed", range(12))
def test_library_alpha_is_the_oracles(seed):
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(8, 90)), int(rng.integers(8, 70))
    polys = random_polygons(rng, w, h, int(rng.integers(0, 4)))
    kind = int(rng.integers(0, 3))
    x0, x1 = sorted(rng.integers(-5, w + 5, 2).tolist())
    y0, y1 = sorted(rng.integers(-5, h + 5, 2).tolist())
    if kind == 2 and (x1 == x0 or y1 == y0):
        x1, y1 = x0 + 7, y0 + 5          # a degenerate ellipse divides by zero in the reference too
    crop = (x0, x1, y0, y1) if kind else None
    nch = (2, 4)[seed % 2]
    px = rng.random((h, w, nch), dtype=np.float32)
    px0 = px.copy()
    alpha = ea.facet_alpha(px, polys, crop, kind)
    ref = euo.facet_alpha(w, h, polys, crop, kind)
    assert (bits(alpha) == bits(ref)).all()
    assert (bits(px) == bits(px0 * ref[:, :, None])).all()


def test_polygon_fill_is_the_winding_rule():
    """a convex polygon with integer vertices: a pixel (x, y) is cleared iff the scan line y crosses the
    polygon and x lies in [int(left crossing), int(right crossing)) - checked against a float64 model"""
    w, h = 60, 40
    xs = np.array([10, 50, 45, 12], np.float32)
    ys = np.array([5, 8, 33, 30], np.float32)
    a = euo.facet_alpha(w, h, [(xs, ys)], stage=0)
    n = len(xs)
    for y in range(h):
        cr = []
        for i in range(n):
            j = (i - 1) % n
            if (ys[i] < y <= ys[j]) or (ys[j] < y <= ys[i]):
                cr.append(int(float(xs[i]) + (y - float(ys[i])) / (float(ys[j]) - float(ys[i])) * (float(xs[j]) - float(xs[i]))))
        row = np.ones(w, np.float32)
        if len(cr) == 2:
            row[min(cr):max(cr)] = 0
        assert (a[y] == row).all(), y
    # a polygon traversed twice (winding number 2) is filled like the simple one; reversed order too
    a2 = euo.facet_alpha(w, h, [(np.concatenate([xs, xs]), np.concatenate([ys, ys]))], stage=0)
    a3 = euo.facet_alpha(w, h, [(xs[::-1].copy(), ys[::-1].copy())], stage=0)
    assert (a2 == a).all() and (a3 == a).all()


def test_crops():
    w, h = 50, 30
    r = euo.facet_alpha(w, h, crop=(5, 40, 3, 25), crop_kind=1, stage=0)
    want = np.zeros((h, w), np.float32)
    want[3:25, 5:40] = 1
    assert (r == want).all()
    e = euo.facet_alpha(w, h, crop=(5, 45, 2, 28), crop_kind=2, stage=0)
    yy, xx = np.mgrid[0:h, 0:w]
    inside = ((xx - 25.0) / 20.0) ** 2 + ((yy - 15.0) / 13.0) ** 2
    # away from the rim the float32 decisions agree with the float64 ellipse
    assert (e[inside < 0.98] == 1).all() and (e[inside > 1.02] == 0).all()


def test_binomial_properties():
    """a constant plane stays constant (the weights sum to 1, REFLECT adds nothing new); the filter is
    separable and symmetric: the transposed plane gives the transposed result within rounding"""
    c = np.full((9, 14), 0.625, np.float32)
    assert (euo.binomial_plane(c) == c).all()
    rng = np.random.default_rng(3)
    p = rng.random((20, 31), dtype=np.float32)
    f = euo.binomial_plane(p)
    k = np.array([1, 4, 6, 4, 1], np.float64) / 16
    pad = np.pad(p.astype(np.float64), 2, mode="symmetric")       # numpy's symmetric = zimt's REFLECT
    tmp = sum(k[i] * pad[2:-2, i:i + 31] for i in range(5))
    pad2 = np.pad(tmp, ((2, 2), (0, 0)), mode="symmetric")
    want = sum(k[i] * pad2[i:i + 20, :] for i in range(5))
    assert np.abs(f - want).max() < 4e-7


def test_argument_errors():
    px = np.zeros((4, 4, 3), np.float32)
    with pytest.raises(ea.EuError):
        ea.facet_alpha(px)
