"""Host set-up arithmetic of the library (extents, steps, twining taps, cubemap metrics)
against formulations written HERE from the definitions, in float64 with numpy - not against
the oracle, whose set-up code is a line-for-line twin of the library's (the other tests'
`*_match_oracle` comparisons cannot catch a shared misreading; these can). The rotation
(make_r3) is checked the same way on the HIP path's rays in test_gpu_geometry.py."""
import math

import numpy as np
import pytest

import envutil_amd as ea


def half_extent_x(prj, hfov):
    """planar x of a ray hfov / 2 to the right of FORWARD, by the projection's definition"""
    t = hfov / 2.0
    return {ea.SPHERICAL: t, ea.CYLINDRICAL: t, ea.FISHEYE: t,            # angle itself
            ea.RECTILINEAR: math.tan(t), ea.CUBEMAP: math.tan(t), ea.BIATAN6: math.tan(t),
            ea.STEREOGRAPHIC: 2.0 * math.tan(t / 2.0)}[prj]


@pytest.mark.parametrize("prj", range(7))
def test_extent_is_symmetric_with_square_pixels(prj):
    for w, h, hfov in [(1024, 512, 360.0), (640, 480, 90.0), (300, 200, 65.5), (801, 333, 123.4)]:
        if prj in (ea.CUBEMAP, ea.BIATAN6):
            h = 6 * w
            hfov = 90.0
        if (prj == ea.RECTILINEAR and hfov >= 180.0) or (prj == ea.STEREOGRAPHIC and hfov >= 340.0):
            continue
        hf = math.radians(hfov)
        x0, x1, y0, y1 = ea.get_extent(prj, w, h, hf)
        want = half_extent_x(prj, hf)
        assert abs(x1 - want) <= 1e-12 * max(1.0, want) and x0 == -x1 and y0 == -y1
        # pixels are square in model space: the vertical extent follows from the aspect ratio
        assert abs(y1 - want * h / w) <= 1e-12 * max(1.0, want * h / w)
        # get_step (envutil_basic.cc:111): "the width of one pixel in the image center" as an
        # angle - the angle between FORWARD and the ray one pixel to its right, by definition
        dx = 2.0 * want / w
        angle = {ea.SPHERICAL: dx, ea.CYLINDRICAL: dx, ea.FISHEYE: dx, ea.RECTILINEAR: math.atan(dx),
                 ea.CUBEMAP: math.atan(dx), ea.BIATAN6: dx * math.pi / 4.0,
                 ea.STEREOGRAPHIC: 2.0 * math.atan(dx / 2.0)}[prj]
        assert abs(ea.get_step(prj, w, h, hf) - angle) <= 1e-3 * angle


def test_box_spread_is_the_grid_of_cell_centres():
    """w x h equal cells over [-0.5, 0.5]^2 scaled by d, equal weights summing to one, rows first"""
    for w, h, d in [(2, 2, 1.0), (3, 3, 1.0), (5, 4, 1.5), (7, 2, 0.5)]:
        t = ea.make_spread(w, h, d)
        xs = ((np.arange(w) + 0.5) / w - 0.5) * d
        ys = ((np.arange(h) + 0.5) / h - 0.5) * d
        gx, gy = np.meshgrid(xs, ys)
        assert t.shape == (w * h, 3)
        assert np.abs(t[:, 0] - gx.ravel()).max() < 1e-7 and np.abs(t[:, 1] - gy.ravel()).max() < 1e-7
        assert np.abs(t[:, 2] - 1.0 / (w * h)).max() < 1e-8


def test_weighted_spread_decays_with_distance_and_sums_to_one():
    """sigma > 0: weights fall off as exp(-r / (sigma * half_width)), normalised; a threshold
    drops the faint taps and renormalises the rest"""
    w = 7
    t = ea.make_spread(w, w, 1.0, 1.5)
    half = (w - 1.0) / (2.0 * w)
    r = np.hypot(t[:, 0].astype(np.float64), t[:, 1].astype(np.float64))
    want = np.exp(-r / (1.5 * half))
    want /= want.sum()
    assert abs(t[:, 2].sum() - 1.0) < 1e-6 and np.abs(t[:, 2] - want).max() < 1e-7
    cut = ea.make_spread(w, w, 1.0, 1.5, 0.02)
    keep = want >= 0.02
    assert len(cut) == keep.sum() and abs(cut[:, 2].sum() - 1.0) < 1e-6
    assert np.abs(cut[:, 2] - want[keep] / want[keep].sum()).max() < 1e-6


def test_cubemap_metrics_from_their_definition():
    """the IR section holds the face plus a frame that is at least support_min pixels wide, its
    size a multiple of the tile; the face centre maps to the section centre"""
    for face, smin, tile in [(2048, 8, 64), (64, 8, 64), (512, 4, 64), (100, 8, 32), (333, 16, 64)]:
        m = ea.cubemap_metrics(face, math.pi / 2, smin, tile)
        sec, lf = m["section_px"], m["left_frame_px"]
        assert sec % tile == 0 and sec >= face + 2 * smin and sec - tile < face + 2 * smin + tile
        assert lf >= smin and sec - face - lf >= smin and abs((sec - face - lf) - lf) <= 1
        # model space: the 90 degree face spans [-1, 1]; pixels per unit = face / 2
        assert abs(m["model_to_px"] - face / 2.0) < 1e-9
        # the left edge of the section in model units
        assert abs(m["refc_md"] - (1.0 + lf / (face / 2.0))) < 1e-9
