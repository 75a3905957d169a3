"""Geometry of the HIP path checked two ways that do not go through the oracle's copy of the
host set-up code:

* edge-case rays (exact ties and signed zeros of ray_to_cubeface, geometry.h:1189-1191 and
  :1293-1357; axis-aligned, denormal, huge and zero rays for every mount) pushed through the
  kernels' ray -> source-coordinate stage (eu_diag.hip) and compared bit for bit with the oracle;
* the properties the reference's own test program asserts (geometry.cc:411-448 round trips for
  SPHERICAL ... BIATAN6, :560-990 rotated stepper == stepper + rotation), evaluated on the HIP
  path's stage-1 (rays) and stage-2 (source coordinates) OUTPUT against float64 formulations
  written here from the projection definitions and README.md:967-980 (yaw right, pitch up,
  roll clockwise; axes RIGHT, DOWN, FORWARD)."""
import ctypes as C
import math

import numpy as np
import pytest

import envutil_amd as ea
import euo
import jobs
from test_gpu_parity import make_pair

pytestmark = pytest.mark.gpu


def hip_coords(gsrc, rays, variant):
    rays = np.ascontiguousarray(rays, np.float32)
    out = np.zeros_like(rays)
    L = ea.lib()
    L.eu_hip_diag_source_coordinates.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
    rc = L.eu_hip_diag_source_coordinates(gsrc.handle, rays.ctypes.data, len(rays), variant, out.ctypes.data)
    assert rc == 0, L.eu_hip_last_error()
    return out


def edge_rays():
    f = np.float32
    vals = [0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, 1e-40, -1e-40, 1e-30, 3e38, -3e38, 0.7071067811865476]
    rays = [(x, y, z) for x in vals for y in vals for z in vals]
    # ties of the dominance tests with every sign pattern, at awkward magnitudes
    for m in (1.0, 0.3333333432674408, 123.456, 1e-20):
        for sx in (1, -1):
            for sy in (1, -1):
                for sz in (1, -1):
                    rays += [(sx * m, sy * m, sz * m), (sx * m, sy * m, sz * m * 0.5), (sx * m, sy * m * 0.5, sz * m),
                             (sx * m * 0.5, sy * m, sz * m), (sx * m, sy * np.nextafter(f(m), f(0)), sz * m)]
    rng = np.random.default_rng(99)
    rnd = rng.normal(size=(4000, 3))
    rnd[:1000, 0] = 0.0
    rnd[1000:2000, 1] = 0.0
    rnd[2000:3000, 2] = 0.0
    return np.concatenate([np.array(rays, np.float64), rnd]).astype(np.float32)


def same_or_both_nan(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


SOURCES = [(euo.SPHERICAL, 64, 32, 360.0), (euo.SPHERICAL, 64, 32, 200.0), (euo.CYLINDRICAL, 64, 40, 300.0),
           (euo.RECTILINEAR, 64, 48, 100.0), (euo.STEREOGRAPHIC, 64, 64, 250.0), (euo.FISHEYE, 64, 64, 190.0),
           (euo.CUBEMAP, 32, 192, 90.0), (euo.BIATAN6, 32, 192, 90.0)]


@pytest.mark.parametrize("prj,w,h,hfov", SOURCES)
def test_edge_rays_source_coordinate_bit_exact(prj, w, h, hfov):
    img = jobs.synth_cubefaces(w, 3) if prj in (euo.CUBEMAP, euo.BIATAN6) else jobs.synth_image(w, h, 3)
    o, g = make_pair(prj, w, h, hfov, img, 1)
    rays = edge_rays()
    ref = euo.source_coordinates(o.s, rays)
    got = hip_coords(g, rays, 0)
    ok = same_or_both_nan(got, ref).all(axis=1)
    assert ok.all(), f"{(~ok).sum()} rays differ, first {rays[~ok][0]!r}: gpu {got[~ok][0]!r} oracle {ref[~ok][0]!r}"
    if prj in (euo.SPHERICAL, euo.CUBEMAP, euo.BIATAN6):
        # The packed kernels' form: the face is folded into y, the third value is the hit flag.
        # The null ray (all components +-0: no stepper produces it) is outside their contract:
        # its in-face coordinates are NaN, which the table-driven atanf of the biatan6 pickup
        # does not propagate (eu_math2.h: the range select treats NaN as 'large').
        nonnull = (rays != 0).any(axis=1)
        rays, ref = rays[nonnull], ref[nonnull]
        hitflag = np.where(ref[:, 2] < 0, -1.0, 0.0).astype(np.float32)
        got1 = hip_coords(g, rays, 1)
        ok = same_or_both_nan(got1[:, :2], ref[:, :2]).all(axis=1) & (got1[:, 2] == hitflag)
        assert ok.all(), f"packed: {(~ok).sum()} rays differ, first {rays[~ok][0]!r}: {got1[~ok][0]!r} vs {ref[~ok][0]!r}"
        # the staged kernel's form has no fallbacks: a lane either agrees or says so (-2)
        got2 = hip_coords(g, rays, 2)
        flagged = got2[:, 2] == -2.0
        ok = flagged | (same_or_both_nan(got2[:, :2], ref[:, :2]).all(axis=1) & (got2[:, 2] == hitflag))
        assert ok.all(), f"fast path: {(~ok).sum()} unflagged rays differ, first {rays[~ok][0]!r}"
        # ... and ordinary rays are never flagged
        mag = np.abs(rays)
        ordinary = ((mag > 1e-3) & (mag < 1e3)).all(axis=1)
        assert not flagged[ordinary].any()


# ---------------------------------------------------------------------------
# float64 formulations, written from the definitions
# ---------------------------------------------------------------------------

def rot(roll, pitch, yaw):
    """camera orientation: yaw to the right about DOWN, pitch upward about RIGHT, roll clockwise
    about FORWARD, applied roll first (column vectors): R = Ry(yaw) Rx(pitch) Rz(roll)"""
    cy, sy, cp, sp, cr, sr = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch), math.cos(roll), math.sin(roll)
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    return ry @ rx @ rz


def planar_to_ray(prj, px, py):
    """pixel-centre planar coordinates -> ray (RIGHT, DOWN, FORWARD), float64"""
    if prj == ea.SPHERICAL:
        return np.stack([np.cos(py) * np.sin(px), np.sin(py), np.cos(py) * np.cos(px)], -1)
    if prj == ea.CYLINDRICAL:
        return np.stack([np.sin(px), py, np.cos(px)], -1)
    if prj == ea.RECTILINEAR:
        return np.stack([px, py, np.ones_like(px)], -1)
    r = np.hypot(px, py)
    theta = r if prj == ea.FISHEYE else 2.0 * np.arctan(r / 2.0)       # stereographic: r = 2 tan(theta / 2)
    phi = np.arctan2(px, py)                                           # from the DOWN axis towards RIGHT
    return np.stack([np.sin(theta) * np.sin(phi), np.sin(theta) * np.cos(phi), np.cos(theta)], -1)


def cube_ray(face, in0, in1):
    """in-face coordinates in [-1, 1] -> ray; faces LEFT0 RIGHT1 TOP2 BOTTOM3 FRONT4 BACK5"""
    one = np.ones_like(in0)
    table = [(-one, in1, in0), (one, in1, -in0), (-in0, -one, -in1), (-in0, one, in1), (in0, in1, one), (-in0, in1, -one)]
    out = np.zeros(in0.shape + (3,))
    for f, (x, y, z) in enumerate(table):
        m = face == f
        out[m] = np.stack([x, y, z], -1)[m]
    return out


def unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def extent(prj, w, h, hfov):
    """half extents from the definitions: the planar x of a ray hfov / 2 to the right, square pixels"""
    t = hfov / 2.0
    x1 = {ea.SPHERICAL: t, ea.CYLINDRICAL: t, ea.FISHEYE: t, ea.RECTILINEAR: math.tan(t),
          ea.STEREOGRAPHIC: 2.0 * math.tan(t / 2.0), ea.CUBEMAP: math.tan(t), ea.BIATAN6: math.tan(t)}[prj]
    return x1, x1 * h / w


TARGETS = [(ea.SPHERICAL, 120, 60, 360.0), (ea.CYLINDRICAL, 100, 50, 200.0), (ea.RECTILINEAR, 90, 70, 95.0),
           (ea.STEREOGRAPHIC, 80, 80, 200.0), (ea.FISHEYE, 80, 60, 180.0), (ea.CUBEMAP, 24, 144, 90.0),
           (ea.BIATAN6, 24, 144, 90.0)]


def target_rays_f64(prj, w, h, hfov_deg):
    x1, y1 = extent(prj, w, h, math.radians(hfov_deg))
    xs = -x1 + (np.arange(w) + 0.5) * (2 * x1 / w)
    ys = -y1 + (np.arange(h) + 0.5) * (2 * y1 / h)
    px, py = np.meshgrid(xs, ys)
    if prj in (ea.CUBEMAP, ea.BIATAN6):
        face = (np.arange(h) // w)[:, None] + np.zeros((1, w), int)
        in1 = py + (5 - 2 * face) * x1          # the face's own vertical coordinate in [-x1, x1]
        in0 = px
        if prj == ea.BIATAN6:
            in0, in1 = np.tan(in0 * math.pi / 4.0), np.tan(in1 * math.pi / 4.0)
        return cube_ray(face, in0, in1)
    return planar_to_ray(prj, px, py)


@pytest.mark.parametrize("tprj,tw,th,thfov", TARGETS)
def test_hip_rays_match_the_projection_definitions(tprj, tw, th, thfov):
    """stage 1 of the HIP path against the float64 definition of every target projection, and
    (geometry.cc:560-990) the rotated stepper against the unrotated one followed by the rotation"""
    img = jobs.synth_image(64, 32, 3)
    _, g = make_pair(euo.SPHERICAL, 64, 32, 360.0, img, 1)
    r0 = ea.render(ea.arguments(tprj, tw, th, thfov, spline_degree=1), g, stage=1).astype(np.float64)
    want = target_rays_f64(tprj, tw, th, thfov)
    assert np.abs(unit(r0) - unit(want)).max() < 3e-6
    for roll, pitch, yaw in ((48.0, -12.0, 33.0), (0.0, 90.0, 0.0), (-170.0, 5.0, 250.0)):
        r1 = ea.render(ea.arguments(tprj, tw, th, thfov, yaw=yaw, pitch=pitch, roll=roll, spline_degree=1),
                       g, stage=1).astype(np.float64)
        m = rot(math.radians(roll), math.radians(pitch), math.radians(yaw))
        assert np.abs(unit(r1) - unit(r0 @ m.T)).max() < 3e-6, (roll, pitch, yaw)


@pytest.mark.parametrize("sprj,sw,sh,shfov", [(euo.SPHERICAL, 128, 64, 360.0), (euo.CYLINDRICAL, 128, 50, 300.0),
                                              (euo.RECTILINEAR, 96, 72, 110.0), (euo.STEREOGRAPHIC, 96, 96, 220.0),
                                              (euo.FISHEYE, 96, 96, 200.0), (euo.CUBEMAP, 48, 288, 90.0),
                                              (euo.BIATAN6, 48, 288, 90.0)])
def test_hip_source_coordinates_round_trip(sprj, sw, sh, shfov):
    """geometry.cc:411-448 on the HIP path: the source pixel coordinate of stage 2, taken back to a
    ray with the float64 definition of the source projection, is the ray of stage 1 - also for
    CUBEMAP and BIATAN6 sources and with the facet itself oriented"""
    img = jobs.synth_cubefaces(sw, 3) if sprj in (euo.CUBEMAP, euo.BIATAN6) else jobs.synth_image(sw, sh, 3)
    cube = sprj in (euo.CUBEMAP, euo.BIATAN6)
    for fypr in ((0.0, 0.0, 0.0), (20.0, -35.0, 10.0)):
        _, g = make_pair(sprj, sw, sh, shfov, img, 1, yaw=fypr[0], pitch=fypr[1], roll=fypr[2])
        a = ea.arguments(ea.RECTILINEAR, 60, 40, 70.0, yaw=15.0, pitch=-8.0, roll=4.0, spline_degree=1)
        rays = ea.render(a, g, stage=1).astype(np.float64)      # in the facet's frame
        crd = ea.render(a, g, stage=2).astype(np.float64)
        hit = crd[:, :, 2] >= 0
        assert hit.mean() > 0.5
        if cube:
            m = ea.cubemap_metrics(sw)
            face = crd[:, :, 2].astype(int)
            p0 = crd[:, :, 0]
            p1 = crd[:, :, 1] - face * m["section_px"]
            in0 = (p0 + 0.5) / m["model_to_px"] - m["refc_md"]
            in1 = (p1 + 0.5) / m["model_to_px"] - m["refc_md"]
            if sprj == euo.BIATAN6:
                in0, in1 = np.tan(in0 * math.pi / 4.0), np.tan(in1 * math.pi / 4.0)
            back = cube_ray(face, in0, in1)
        else:
            x1, y1 = extent(sprj, sw, sh, math.radians(shfov))
            px = (crd[:, :, 0] + 0.5) / sw * (2 * x1) - x1
            py = (crd[:, :, 1] + 0.5) / sh * (2 * y1) - y1
            back = planar_to_ray(sprj, px, py)
        err = np.abs(unit(back) - unit(rays))[hit].max()
        assert err < 2e-5, (sprj, fypr, err)


def test_facet_orientation_is_the_inverse_camera_rotation():
    """a facet and a camera with the same yaw / pitch / roll see each other head on: the source
    coordinates are those of the unrotated pair"""
    img = jobs.synth_image(128, 64, 3)
    _, g0 = make_pair(euo.RECTILINEAR, 128, 64, 80.0, img, 1)
    _, g1 = make_pair(euo.RECTILINEAR, 128, 64, 80.0, img, 1, yaw=37.0, pitch=21.0, roll=-13.0)
    c0 = ea.render(ea.arguments(ea.RECTILINEAR, 64, 32, 60.0, spline_degree=1), g0, stage=2)
    c1 = ea.render(ea.arguments(ea.RECTILINEAR, 64, 32, 60.0, yaw=37.0, pitch=21.0, roll=-13.0, spline_degree=1),
                   g1, stage=2)
    assert (c0[:, :, 2] >= 0).all() and (c1[:, :, 2] >= 0).all()
    assert np.abs(c0[:, :, :2] - c1[:, :, :2]).max() < 2e-3
