"""The checks the reference's own test program makes (geometry.cc:283-420,
:560-990), re-expressed on the oracle: they pin the geometry conventions (axes,
cube-face numbering and orientation, stepper <-> functor equivalence), not any
float32 result."""
import math

import numpy as np
import pytest

import envutil_amd as ea
import euo
import jobs


@pytest.mark.parametrize("prj", [euo.SPHERICAL, euo.CYLINDRICAL, euo.RECTILINEAR,
                                 euo.STEREOGRAPHIC, euo.FISHEYE])
def test_ray_to_2d_to_ray_round_trip(prj):
    """geometry.cc test_r2r: ray -> planar -> ray returns the direction, 1e-13"""
    rng = np.random.default_rng(prj)
    L = euo.lib()
    for _ in range(2000):
        v = rng.normal(size=3)
        if prj == euo.RECTILINEAR:
            v[2] = abs(v[2]) + 0.05
        v /= np.linalg.norm(v)
        p2 = np.zeros(2)
        back = np.zeros(3)
        L.euo_ray_to_prj_d(prj, euo.ptr(v), euo.ptr(p2))
        L.euo_prj_to_ray_d(prj, euo.ptr(p2), euo.ptr(back))
        back /= np.linalg.norm(back)
        assert np.abs(back - v).max() < 1e-13, (prj, v, back)


def test_rotation_conventions():
    """README.md:967-980 / SURVEY A.1: yaw looks right, pitch looks up (y is
    DOWN), roll turns clockwise"""
    fwd = np.array([0.0, 0.0, 1.0])
    up = np.array([0.0, -1.0, 0.0])
    assert np.allclose(fwd @ euo.make_r3(0, 0, 0.3), [math.sin(0.3), 0, math.cos(0.3)], atol=1e-7)
    assert np.allclose(fwd @ euo.make_r3(0, 0.3, 0), [0, -math.sin(0.3), math.cos(0.3)], atol=1e-7)
    assert np.allclose(up @ euo.make_r3(0.3, 0, 0), [math.sin(0.3), -math.cos(0.3), 0], atol=1e-7)
    assert np.allclose(fwd @ euo.make_r3(0.1, 0.2, 0.3), [0.289629, -0.198669, 0.936293], atol=1e-6)
    m, mi = euo.make_r3(0.4, -0.2, 1.1), euo.make_r3(0.4, -0.2, 1.1, True)
    assert np.allclose(m @ mi, np.eye(3), atol=1e-7) and np.allclose(m @ m.T, np.eye(3), atol=1e-7)


@pytest.mark.parametrize("prj,w,h,hfov", [(ea.SPHERICAL, 96, 48, 360.0), (ea.CYLINDRICAL, 90, 40, 200.0),
                                          (ea.RECTILINEAR, 80, 60, 90.0), (ea.CUBEMAP, 32, 192, 90.0),
                                          (ea.FISHEYE, 64, 64, 170.0), (ea.STEREOGRAPHIC, 64, 64, 120.0)])
def test_rotated_stepper_equals_stepper_then_rotation(prj, w, h, hfov):
    """geometry.cc:560-990: a stepper built with the rotated basis yields the
    unrotated stepper's rays pushed through the rotation"""
    img = jobs.synth_image(64, 32, 3)
    o = jobs.OracleSource(euo.SPHERICAL, 64, 32, 360.0, img, 1)
    a0 = ea.arguments(prj, w, h, hfov, spline_degree=1)
    a1 = ea.arguments(prj, w, h, hfov, yaw=33.0, pitch=-12.0, roll=48.0, spline_degree=1)
    r0 = jobs.oracle_render(a0, o, stage=1).astype(np.float64)
    r1 = jobs.oracle_render(a1, o, stage=1).astype(np.float64)
    m = euo.make_r3(math.radians(48.0), math.radians(-12.0), math.radians(33.0))
    assert np.abs(r0 @ m - r1).max() < 2e-5


def test_cubemap_stepper_and_cubeface_agree():
    """face numbering / orientation: a cubemap rendered onto a cubemap of the
    same size reproduces its faces (bilinear at pixel centres is the identity)"""
    faces = jobs.synth_cubefaces(64, 3)
    o = jobs.OracleSource(euo.CUBEMAP, 64, 384, 90.0, faces, 1)
    a = ea.arguments(ea.CUBEMAP, 64, 384, 90.0, spline_degree=1)
    out = jobs.oracle_render(a, o)
    assert np.abs(out - faces).max() < 2e-4
    # and the pickup lands in the right IR section: face index from stage 2
    dbg = jobs.oracle_render(a, o, stage=2)
    assert (dbg[:, :, 2].reshape(6, 64, 64) == np.arange(6)[:, None, None]).all()


def test_latlon_identity_reprojection():
    img = jobs.synth_image(256, 128, 3)
    o = jobs.OracleSource(euo.SPHERICAL, 256, 128, 360.0, img, 3)
    a = ea.arguments(ea.SPHERICAL, 256, 128, 360.0, spline_degree=3)
    assert np.abs(jobs.oracle_render(a, o) - img).max() < 2e-4


def test_cubemap_of_latlon_round_trip():
    """lat/lon -> cubemap -> lat/lon comes back to the smooth field (the two
    directions of the geometry are inverse to each other)"""
    x = np.arange(512)[None, :]
    y = np.arange(256)[:, None]
    img = np.stack([0.5 + 0.3 * np.sin(2 * np.pi * x / 512) * np.cos(np.pi * (y + 0.5) / 256 - np.pi / 2)] * 3,
                   -1).astype(np.float32)
    o = jobs.OracleSource(euo.SPHERICAL, 512, 256, 360.0, img, 3)
    cube = jobs.oracle_render(ea.arguments(ea.CUBEMAP, 128, 768, 90.0, spline_degree=3), o)
    oc = jobs.OracleSource(euo.CUBEMAP, 128, 768, 90.0, cube, 3)
    back = jobs.oracle_render(ea.arguments(ea.SPHERICAL, 512, 256, 360.0, spline_degree=3), oc)
    assert np.abs(back - img).max() < 5e-3


def test_cropped_render_is_a_window_of_the_frame():
    """store_cropped (envutil_payload.cc:440-474): a crop that starts on a
    512-pixel segment boundary reproduces the frame's pixels bit for bit (same
    segment starts, same number of delta additions); any other crop origin
    restarts the segments and agrees to rounding only"""
    img = jobs.synth_image(128, 64, 3)
    o = jobs.OracleSource(euo.SPHERICAL, 128, 64, 360.0, img, 3)
    kw = dict(yaw=25, pitch=10, roll=-4, spline_degree=3)
    full = jobs.oracle_render(ea.arguments(ea.RECTILINEAR, 1100, 40, 100.0, **kw), o)
    a = ea.arguments(ea.RECTILINEAR, 1100, 40, 100.0, crop=(512, 1100, 7, 33), **kw)
    part = jobs.oracle_render(a, o)
    assert part.shape == (26, 588, 3)
    assert (jobs.bits(part) == jobs.bits(full[7:33, 512:1100])).all()
    a = ea.arguments(ea.RECTILINEAR, 1100, 40, 100.0, crop=(37, 700, 0, 40), **kw)
    part = jobs.oracle_render(a, o)
    assert np.abs(part - full[:, 37:700]).max() < 1e-4
    assert (jobs.bits(part) != jobs.bits(full[:, 37:700])).any()


def test_tethered_words_follow_the_float_pixels():
    """to_screen_t: each byte is the truncated LUT value of the float pixel -
    within 1 of 255 * sRGB(v), alpha 255 for RGB"""
    img = jobs.synth_image(128, 64, 3)
    o = jobs.OracleSource(euo.SPHERICAL, 128, 64, 360.0, img, 1)
    fl = jobs.oracle_render(ea.arguments(ea.SPHERICAL, 90, 45, 360.0, spline_degree=1), o)
    w = jobs.oracle_render(ea.arguments(ea.SPHERICAL, 90, 45, 360.0, spline_degree=1, tethered=True), o)
    assert w.dtype == np.uint32 and (w >> 24 == 255).all()
    v = np.clip(fl.astype(np.float64), 0.0, 1.0)
    srgb = np.where(v <= 0.0031308, 12.92 * v, 1.055 * v ** (1 / 2.4) - 0.055) * 255.0
    for c in range(3):
        byte = ((w >> (8 * c)) & 0xFF).astype(np.float64)
        assert np.all(byte <= srgb[:, :, c] + 0.51) and np.all(byte >= srgb[:, :, c] - 1.01)
