"""Parity tests proper: the HIP path, called through the C ABI, against the
CPU oracle on identical inputs. BIT-EXACT (float32 compared as uint32) at every
stage: rays, source coordinates, pixels, and the device-built coefficients."""
import math
import os

import numpy as np
import pytest

import envutil_amd as ea
import euo
import jobs

pytestmark = pytest.mark.gpu


def assert_bits(a, b, what=""):
    assert a.shape == b.shape, (a.shape, b.shape)
    bad = np.argwhere(jobs.bits(a) != jobs.bits(b))
    if bad.size:
        i = tuple(bad[0])
        ul = jobs.ulp_diff(a, b).max()
        raise AssertionError(f"{what}: {len(bad)} of {a.size} floats differ (max {ul} ULP); "
                             f"first at {i}: gpu {a[i]!r} oracle {b[i]!r}")


SRC_W, SRC_H = 256, 128


@pytest.fixture(scope="module")
def latlon():
    return {n: jobs.synth_image(SRC_W, SRC_H, n) for n in (1, 3, 4)}


def make_pair(prj, w, h, hfov, img, degree, pdeg=None, **kw):
    o = jobs.OracleSource(prj, w, h, hfov, img, degree, pdeg, **kw)
    fct = ea.facet_spec(prj, w, h, hfov, nchannels=img.shape[2],
                        yaw=kw.get("yaw", 0.0), pitch=kw.get("pitch", 0.0),
                        roll=kw.get("roll", 0.0), brighten=kw.get("brighten", 1.0),
                        lens=kw.get("lens"))
    # the GPU gets the oracle's coefficients: stage-wise parity of the render
    # path alone ("given identical coefficients")
    g = ea.Source.adopt(fct, o.container, degree, o.bc[0], o.bc[1])
    return o, g


TARGETS = [
    (ea.CUBEMAP, 40, 240, 90.0),
    (ea.SPHERICAL, 200, 100, 360.0),
    (ea.RECTILINEAR, 129, 97, 90.0),
    (ea.CYLINDRICAL, 150, 70, 220.0),
    (ea.BIATAN6, 24, 144, 90.0),
    (ea.SPHERICAL, 1100, 12, 360.0),      # segments of 512 + leftover lanes
    (ea.FISHEYE, 120, 90, 200.0),         # per-pixel sinf/cosf/atan2f on the device
    (ea.STEREOGRAPHIC, 110, 84, 240.0),   # ... plus the double atan of stepper.h:1146
]


@pytest.mark.parametrize("tprj,tw,th,thfov", TARGETS)
@pytest.mark.parametrize("ypr", [(0, 0, 0), (30, 15, 7.5)])
def test_rays_bit_exact(latlon, tprj, tw, th, thfov, ypr):
    o, g = make_pair(euo.SPHERICAL, SRC_W, SRC_H, 360.0, latlon[3], 1)
    for twine in (0, 2):
        a = ea.arguments(tprj, tw, th, thfov, yaw=ypr[0], pitch=ypr[1], roll=ypr[2],
                         spline_degree=1, twine=twine)
        assert_bits(ea.render(a, g, stage=1), jobs.oracle_render(a, o, stage=1), "rays")


@pytest.mark.parametrize("tprj,tw,th,thfov", TARGETS[:4])
def test_source_coordinates_bit_exact(latlon, tprj, tw, th, thfov):
    a = ea.arguments(tprj, tw, th, thfov, yaw=-40, pitch=20, roll=-5, spline_degree=1)
    o, g = make_pair(euo.SPHERICAL, SRC_W, SRC_H, 360.0, latlon[3], 1)
    assert_bits(ea.render(a, g, stage=2), jobs.oracle_render(a, o, stage=2), "lat/lon coordinate")
    faces = jobs.synth_cubefaces(64, 3)
    o, g = make_pair(euo.CUBEMAP, 64, 384, 90.0, faces, 1)
    assert_bits(ea.render(a, g, stage=2), jobs.oracle_render(a, o, stage=2), "cubemap pickup")


@pytest.mark.parametrize("degree", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("nch", [1, 3, 4])
def test_pixels_bit_exact_latlon_source(latlon, degree, nch):
    o, g = make_pair(euo.SPHERICAL, SRC_W, SRC_H, 360.0, latlon[nch], degree)
    for tprj, tw, th, thfov in TARGETS[:3]:
        a = ea.arguments(tprj, tw, th, thfov, yaw=12, pitch=-33, roll=4, spline_degree=degree)
        assert_bits(ea.render(a, g), jobs.oracle_render(a, o), f"pixels deg {degree} nch {nch} prj {tprj}")


@pytest.mark.parametrize("sprj,sw,sh,shfov", [
    (euo.RECTILINEAR, 200, 150, 80.0),      # misses: outside the window -> 0
    (euo.CYLINDRICAL, 256, 100, 360.0),
    (euo.SPHERICAL, 200, 80, 220.0),        # partial sphere: REFLECT, ordinary prefilter
    (euo.STEREOGRAPHIC, 160, 160, 150.0),
    (euo.FISHEYE, 180, 180, 190.0),
])
@pytest.mark.parametrize("lens", [None, dict(a=0.01, b=-0.03, c=0.02),
                                  dict(a=0.02, b=0.0, c=-0.01, h=0.01, v=-0.02, g=0.003, t=-0.002)])
def test_pixels_bit_exact_other_mounts(sprj, sw, sh, shfov, lens):
    img = jobs.synth_image(sw, sh, 3, seed=99)
    o, g = make_pair(sprj, sw, sh, shfov, img, 3, yaw=10, pitch=5, roll=-3, brighten=1.25, lens=lens)
    for tprj, tw, th, thfov in [(ea.SPHERICAL, 160, 80, 360.0), (ea.RECTILINEAR, 100, 80, 70.0)]:
        a = ea.arguments(tprj, tw, th, thfov, yaw=5, pitch=2, roll=1, spline_degree=3)
        got, ref = ea.render(a, g), jobs.oracle_render(a, o)
        assert_bits(got, ref, f"mount {sprj}")
        if tprj == ea.SPHERICAL and sprj not in (euo.CYLINDRICAL, euo.FISHEYE):
            # the source does not cover the sphere: the miss path was exercised
            assert (ref == 0).all(axis=2).any() and (ref != 0).any()


@pytest.mark.parametrize("sprj", [euo.CUBEMAP, euo.BIATAN6])
@pytest.mark.parametrize("degree", [1, 3])
def test_pixels_bit_exact_cubemap_source(sprj, degree):
    faces = jobs.synth_cubefaces(64, 3)
    o, g = make_pair(sprj, 64, 384, 90.0, faces, degree)
    for tprj, tw, th, thfov in [(ea.SPHERICAL, 256, 128, 360.0), (ea.CUBEMAP, 48, 288, 90.0)]:
        a = ea.arguments(tprj, tw, th, thfov, yaw=45, pitch=35.26, roll=0, spline_degree=degree)
        assert_bits(ea.render(a, g), jobs.oracle_render(a, o), "cubemap source")


@pytest.mark.parametrize("twine", [2, 3])
@pytest.mark.parametrize("tprj,tw,th,thfov", TARGETS[:5])
def test_twining_bit_exact(latlon, twine, tprj, tw, th, thfov):
    o, g = make_pair(euo.SPHERICAL, SRC_W, SRC_H, 360.0, latlon[3], 1)
    a = ea.arguments(tprj, tw, th, thfov, yaw=30, pitch=15, roll=7.5, spline_degree=1, twine=twine)
    assert_bits(ea.render(a, g), jobs.oracle_render(a, o), "twining")


def test_twining_gaussian_taps_and_rgba(latlon):
    o, g = make_pair(euo.SPHERICAL, SRC_W, SRC_H, 360.0, latlon[4], 3, brighten=0.8)
    a = ea.arguments(ea.SPHERICAL, 96, 48, 360.0, yaw=3, spline_degree=3, twine=5,
                     twine_width=1.3, twine_sigma=1.2, twine_threshold=0.03)
    assert len(a.twine_spread) < 25
    assert_bits(ea.render(a, g), jobs.oracle_render(a, o), "gaussian twining")


def test_row_tiles_equal_the_whole(latlon):
    o, g = make_pair(euo.SPHERICAL, SRC_W, SRC_H, 360.0, latlon[3], 3)
    a = ea.arguments(ea.CUBEMAP, 40, 240, 90.0, spline_degree=3)
    whole = ea.render(a, g)
    for r0, r1 in [(0, 30), (30, 61), (61, 240), (100, 100)]:
        assert_bits(ea.render(a, g, row_begin=r0, row_end=r1), whole[r0:r1], "row tile")
    assert_bits(whole, jobs.oracle_render(a, o), "whole")


# ---- device-side set-up: coefficients built in HBM ------------------------

@pytest.mark.parametrize("degree,pdeg", [(0, 0), (1, 1), (2, 2), (3, 3), (5, 5), (1, 3), (4, 3)])
def test_device_spherical_prefilter_bit_exact(latlon, degree, pdeg):
    for nch in (3, 4):
        fct = ea.facet_spec(ea.SPHERICAL, SRC_W, SRC_H, 360.0, nchannels=nch)
        g = ea.Source.load(fct, latlon[nch], degree, pdeg)
        o = jobs.OracleSource(euo.SPHERICAL, SRC_W, SRC_H, 360.0, latlon[nch], degree, pdeg)
        assert_bits(g.download(), o.container, f"spherical coefficients deg {degree}/{pdeg}")


@pytest.mark.parametrize("sprj,sw,sh,shfov", [(euo.RECTILINEAR, 200, 150, 80.0),
                                             (euo.CYLINDRICAL, 256, 100, 360.0),
                                             (euo.SPHERICAL, 30, 9, 100.0)])
@pytest.mark.parametrize("degree", [1, 2, 3, 7])
def test_device_ordinary_prefilter_bit_exact(sprj, sw, sh, shfov, degree):
    img = jobs.synth_image(sw, sh, 3, seed=5)
    g = ea.Source.load(ea.facet_spec(sprj, sw, sh, shfov), img, degree)
    o = jobs.OracleSource(sprj, sw, sh, shfov, img, degree)
    assert_bits(g.download(), o.container, "ordinary coefficients")


# odd face sizes: the frame is one pixel wider on the right / below; the frame pixels
# next to the face then tie with their own face's edge and the fill ORDER is part of
# the result (eu_setup.hip: fill_tie_kernel)
@pytest.mark.parametrize("face,degree", [(64, 1), (64, 3), (100, 3), (128, 2), (47, 0), (33, 3), (101, 1)])
def test_device_cubemap_ir_bit_exact(face, degree):
    faces = jobs.synth_cubefaces(face, 3)
    g = ea.Source.load(ea.facet_spec(ea.CUBEMAP, face, 6 * face, 90.0), faces, degree)
    o = jobs.OracleSource(euo.CUBEMAP, face, 6 * face, 90.0, faces, degree)
    assert_bits(g.download(), o.container, "cubemap IR")


def test_end_to_end_load_and_render_matches_oracle(latlon):
    """config-2/headline shape at test size: everything on the device"""
    fct = ea.facet_spec(ea.SPHERICAL, SRC_W, SRC_H, 360.0)
    g = ea.Source.load(fct, latlon[3], 3)
    o = jobs.OracleSource(euo.SPHERICAL, SRC_W, SRC_H, 360.0, latlon[3], 3)
    a = ea.arguments(ea.CUBEMAP, 64, 384, 90.0, spline_degree=3)
    dp = ea.get_dispatch().bind(a, g)
    assert dp.payload(3, 3, ea.CUBEMAP) == 0
    assert_bits(dp.result, jobs.oracle_render(a, o), "payload")


def test_identity_reprojection_property(latlon):
    """size-independent property: a full-sphere image reprojected onto its own
    grid with a prefiltered cubic spline reproduces itself to the prefilter
    tolerance (1e-4, environment.h:381-385)"""
    fct = ea.facet_spec(ea.SPHERICAL, SRC_W, SRC_H, 360.0)
    g = ea.Source.load(fct, latlon[3], 3)
    a = ea.arguments(ea.SPHERICAL, SRC_W, SRC_H, 360.0, spline_degree=3)
    assert np.abs(ea.render(a, g) - latlon[3]).max() < 2e-4


# ---- multi-facet jobs: fusion_t + voronoi synopsis --------------------------

def facet_set(prj, w, h, hfov, nch, degree, lens=None, seed=0):
    """six facets looking front/right/back/left/up/down (BASELINE config 5 shape)"""
    views = [(0, 0, 0), (90, 0, 3), (180, 0, -2), (270, 0, 1), (0, 90, 0), (0, -90, 5)]
    os_, gs = [], []
    for i, (yaw, pitch, roll) in enumerate(views):
        img = jobs.synth_image(w, h, nch, seed=seed + 17 * i)
        if nch in (2, 4):
            # feathered alpha so that compositing below the top layer matters
            yy, xx = np.mgrid[0:h, 0:w]
            r = np.hypot((xx - w / 2) / (w / 2), (yy - h / 2) / (h / 2))
            img[:, :, nch - 1] = np.clip(1.6 - 1.4 * r, 0.0, 1.0)
            img[:, :, :nch - 1] *= img[:, :, nch - 1:]
        o = jobs.OracleSource(prj, w, h, hfov, img, degree, yaw=yaw, pitch=pitch, roll=roll,
                              brighten=1.0 + 0.05 * i, lens=lens)
        g = ea.Source.adopt(ea.facet_spec(prj, w, h, hfov, nchannels=nch, yaw=yaw, pitch=pitch,
                                          roll=roll, brighten=1.0 + 0.05 * i, lens=lens),
                            o.container, degree, o.bc[0], o.bc[1])
        os_.append(o)
        gs.append(g)
    return os_, gs


@pytest.mark.parametrize("nch", [3, 4])
@pytest.mark.parametrize("degree", [1, 3])
def test_multi_facet_fisheye_stitch_bit_exact(nch, degree):
    """config 5 at test size: six circular-fisheye facets with the PTO lens
    polynomial -> spherical; voronoi_syn (RGB) / voronoi_syn_plus (RGBA)"""
    lens = dict(a=0.01, b=-0.03, c=0.02)
    os_, gs = facet_set(euo.FISHEYE, 96, 96, 130.0, nch, degree, lens)
    a = ea.arguments(ea.SPHERICAL, 300, 150, 360.0, yaw=10, pitch=4, roll=-2, spline_degree=degree)
    got, ref = ea.render(a, gs, nch), jobs.oracle_render(a, os_)
    assert_bits(got, ref, f"multi-facet nch {nch} degree {degree}")
    assert (ref[:, :, 0] != 0).mean() > 0.9


@pytest.mark.parametrize("nch", [1, 2, 3, 4])
def test_multi_facet_rectilinear_and_holes(nch):
    """rectilinear facets with 100 degree fov leave no hole; with 70 degrees
    they do (champion -1 -> 0); targets other than spherical"""
    for hfov in (100.0, 70.0):
        os_, gs = facet_set(euo.RECTILINEAR, 80, 80, hfov, nch, 1, seed=5)
        for tprj, tw, th, thfov in [(ea.CUBEMAP, 40, 240, 90.0), (ea.RECTILINEAR, 111, 77, 120.0)]:
            a = ea.arguments(tprj, tw, th, thfov, yaw=33, pitch=12, roll=5, spline_degree=1)
            assert_bits(ea.render(a, gs, nch), jobs.oracle_render(a, os_), f"rect facets {hfov} {tprj}")


@pytest.mark.parametrize("nch", [3, 4])
def test_multi_facet_twining_bit_exact(nch):
    os_, gs = facet_set(euo.FISHEYE, 64, 64, 140.0, nch, 1)
    a = ea.arguments(ea.SPHERICAL, 130, 65, 360.0, yaw=5, spline_degree=1, twine=2)
    assert_bits(ea.render(a, gs[:4], nch), jobs.oracle_render(a, os_[:4]), "multi-facet twining")


def test_multi_facet_mixed_projections():
    """facets of different projections in one job (a lat/lon backdrop under
    rectilinear insets)"""
    back = jobs.synth_image(256, 128, 3, seed=1)
    o0 = jobs.OracleSource(euo.SPHERICAL, 256, 128, 360.0, back, 3)
    g0 = ea.Source.adopt(ea.facet_spec(ea.SPHERICAL, 256, 128, 360.0), o0.container, 3, o0.bc[0], o0.bc[1])
    os_, gs = facet_set(euo.RECTILINEAR, 64, 48, 50.0, 3, 3, seed=9)
    a = ea.arguments(ea.SPHERICAL, 256, 128, 360.0, spline_degree=3)
    assert_bits(ea.render(a, [g0] + gs[:3], 3), jobs.oracle_render(a, [o0] + os_[:3]), "mixed facets")


@pytest.mark.parametrize("mix,out_n", [((3, 4), 4), ((1, 3), 3), ((2, 4, 1, 3), 4), ((4, 3), 3),
                                       ((3, 1), 1), ((4, 2), 2)])
@pytest.mark.parametrize("twine", [0, 2])
def test_multi_facet_mixed_channel_counts(mix, out_n, twine):
    """facets with different channel counts in one job: each facet's environment
    adapts through repix_t (environment.h:1846-1900); the job runs at the
    largest count (envutil_main.cc:1003-1157) or any other the caller asks for"""
    sets = {n: facet_set(euo.RECTILINEAR, 72, 72, 95.0, n, 1, seed=21) for n in set(mix)}
    os_ = [sets[mix[i % len(mix)]][0][i] for i in range(6)]
    gs = [sets[mix[i % len(mix)]][1][i] for i in range(6)]
    a = ea.arguments(ea.SPHERICAL, 180, 90, 360.0, yaw=20, pitch=-6, roll=3, spline_degree=1,
                     twine=twine)
    assert_bits(ea.render(a, gs, out_n), jobs.oracle_render(a, os_, nch=out_n), f"mixed channels {mix}->{out_n}")


# ---- channel adaption: repix_t ------------------------------------------------

@pytest.mark.parametrize("src_n", [1, 2, 3, 4])
@pytest.mark.parametrize("out_n", [1, 2, 3, 4])
def test_repix_channel_adaption_bit_exact(src_n, out_n):
    if src_n == out_n:
        pytest.skip("no adaption")
    img = jobs.synth_image(128, 64, src_n, seed=4)
    if src_n in (2, 4):
        # alpha with exact zeros (division by alpha is guarded in repix_t)
        img[:, :, src_n - 1] = (np.indices((64, 128))[1] % 7 != 0).astype(np.float32)
    o = jobs.OracleSource(euo.RECTILINEAR, 128, 64, 100.0, img, 1, brighten=1.3)
    g = ea.Source.adopt(ea.facet_spec(ea.RECTILINEAR, 128, 64, 100.0, nchannels=src_n, brighten=1.3),
                        o.container, 1, o.bc[0], o.bc[1])
    for twine in (0, 2):
        a = ea.arguments(ea.SPHERICAL, 150, 75, 360.0, yaw=20, spline_degree=1, twine=twine)
        assert_bits(ea.render(a, g, out_n), jobs.oracle_render(a, o, nch=out_n), f"repix {src_n}->{out_n}")


@pytest.mark.parametrize("tprj,tw,th,thfov", TARGETS[6:8])
@pytest.mark.parametrize("twine", [0, 3])
def test_pixels_polar_targets(latlon, tprj, tw, th, thfov, twine):
    """fisheye / stereographic targets (per-pixel transcendental steppers), all
    the way to pixels, single source and twined"""
    o, g = make_pair(euo.SPHERICAL, SRC_W, SRC_H, 360.0, latlon[3], 3)
    a = ea.arguments(tprj, tw, th, thfov, yaw=70, pitch=-25, roll=11, spline_degree=3, twine=twine)
    assert_bits(ea.render(a, g), jobs.oracle_render(a, o), f"pixels prj {tprj} twine {twine}")


# ---- store_cropped and the tethered put stage -----------------------------------

CROPS = [(512, 700, 10, 50),       # starts on a segment boundary
         (37, 649, 3, 41),         # ragged: segments of the crop differ from the frame's
         (600, 616, 0, 64),        # one vector wide
         (0, 700, 0, 64)]          # the whole frame, as a crop


@pytest.mark.parametrize("crop", CROPS)
@pytest.mark.parametrize("tprj,th,thfov", [(ea.SPHERICAL, 64, 360.0), (ea.CYLINDRICAL, 64, 300.0),
                                           (ea.RECTILINEAR, 64, 100.0)])
def test_cropped_output_bit_exact(latlon, crop, tprj, th, thfov):
    """args.store_cropped (envutil_payload.cc:440-474): crop-sized output, the
    stepper sees coordinates raised by the crop origin"""
    o, g = make_pair(euo.SPHERICAL, SRC_W, SRC_H, 360.0, latlon[3], 3)
    for twine in (0, 2):
        a = ea.arguments(tprj, 700, th, thfov, yaw=31, pitch=-12, roll=6, spline_degree=3,
                         twine=twine, crop=crop)
        got, ref = ea.render(a, g), jobs.oracle_render(a, o)
        assert got.shape == (crop[3] - crop[2], crop[1] - crop[0], 3)
        assert_bits(got, ref, f"crop {crop} prj {tprj} twine {twine}")


def test_cropped_cubemap_target_and_rows(latlon):
    """crop of a cubemap target straddling two faces (face = (y + y0) / width),
    rendered in two row tiles"""
    o, g = make_pair(euo.SPHERICAL, SRC_W, SRC_H, 360.0, latlon[3], 1)
    a = ea.arguments(ea.CUBEMAP, 48, 288, 90.0, yaw=5, spline_degree=1, crop=(5, 40, 30, 120))
    ref = jobs.oracle_render(a, o)
    got = np.concatenate([ea.render(a, g, row_begin=0, row_end=33), ea.render(a, g, row_begin=33)])
    assert_bits(got, ref, "cropped cubemap target")
    with pytest.raises(ea.EuError):
        ea.render(ea.arguments(ea.SPHERICAL, 64, 32, 360.0, crop=(0, 65, 0, 32)), g)


def test_cropped_multi_facet():
    os_, gs = facet_set(euo.RECTILINEAR, 72, 72, 95.0, 4, 1, seed=3)
    a = ea.arguments(ea.SPHERICAL, 640, 320, 360.0, yaw=20, spline_degree=1, crop=(101, 640, 17, 200))
    assert_bits(ea.render(a, gs, 4), jobs.oracle_render(a, os_), "cropped multi-facet")


@pytest.mark.parametrize("nch", [1, 2, 3, 4])
@pytest.mark.parametrize("twine", [0, 2])
def test_tethered_srgba8_words(nch, twine):
    """act + to_screen_t (envutil_payload.cc:251-413, :524-530): identical
    packed words, general and packed kernels, with values outside [0, 1]"""
    img = jobs.synth_image(SRC_W, SRC_H, nch) * 1.6 - 0.2
    if nch in (2, 4):
        img[:, :, nch - 1] = jobs.synth_image(SRC_W, SRC_H, 1, seed=99)[:, :, 0]
    o, g = make_pair(euo.SPHERICAL, SRC_W, SRC_H, 360.0, img, 3)
    for tprj, tw, th, thfov in [(ea.RECTILINEAR, 333, 77, 90.0), (ea.SPHERICAL, 300, 150, 360.0)]:
        a = ea.arguments(tprj, tw, th, thfov, yaw=12, pitch=7, spline_degree=3, twine=twine, tethered=True)
        got, ref = ea.render(a, g), jobs.oracle_render(a, o)
        assert got.dtype == np.uint32 and got.shape == (th, tw)
        assert np.array_equal(got, ref), f"{(got != ref).sum()} words differ"
        assert len(np.unique(got)) > 100
    # general kernel (a rectilinear source is not on the packed path), cropped
    o, g = make_pair(euo.RECTILINEAR, 200, 150, 80.0, jobs.synth_image(200, 150, nch), 1)
    a = ea.arguments(ea.SPHERICAL, 300, 150, 360.0, spline_degree=1, twine=twine, tethered=True,
                     crop=(100, 220, 40, 110))
    assert np.array_equal(ea.render(a, g), jobs.oracle_render(a, o))


def test_tethered_multi_facet_and_repix():
    os_, gs = facet_set(euo.RECTILINEAR, 72, 72, 95.0, 4, 1, seed=3)
    a = ea.arguments(ea.SPHERICAL, 200, 100, 360.0, yaw=20, spline_degree=1, tethered=True)
    assert np.array_equal(ea.render(a, gs, 4), jobs.oracle_render(a, os_))
    # 3-channel source shown through a 4-channel job: repix, then to_screen
    o, g = make_pair(euo.SPHERICAL, SRC_W, SRC_H, 360.0, jobs.synth_image(SRC_W, SRC_H, 3), 1)
    a = ea.arguments(ea.RECTILINEAR, 120, 90, 80.0, spline_degree=1, tethered=True)
    got, ref = ea.render(a, g, 4), jobs.oracle_render(a, o, nch=4)
    assert np.array_equal(got, ref) and (got >> 24 == 255).all()


# ---- interleaved row bands (multi-GPU tiling) -------------------------------------

@pytest.mark.parametrize("world,band", [(2, 4), (3, 8), (8, 4)])
def test_band_parts_assemble_the_frame(latlon, world, band):
    """eu_target.band_*: every part's compacted rows, put back at their frame
    rows, give the oracle's frame bit for bit (packed, general and multi kernels)"""
    o, g = make_pair(euo.SPHERICAL, SRC_W, SRC_H, 360.0, latlon[3], 3)
    jobs_ = [(ea.arguments(ea.CUBEMAP, 40, 240, 90.0, yaw=3, spline_degree=3), g, o, 3),
             (ea.arguments(ea.FISHEYE, 90, 101, 180.0, spline_degree=3, twine=2), g, o, 3)]
    os_, gs = facet_set(euo.RECTILINEAR, 72, 72, 95.0, 4, 1, seed=3)
    jobs_.append((ea.arguments(ea.SPHERICAL, 120, 75, 360.0, spline_degree=1), gs, os_, 4))
    for a, gg, oo, nch in jobs_:
        ref = jobs.oracle_render(a, oo)
        frame = np.full_like(ref, np.nan)
        for part in range(world):
            rows = ea.band_frame_rows(a.height, band, world, part)
            got = ea.render(a, gg, nch, band=(band, world, part))
            assert got.shape[0] == len(rows) == ea.band_rows(a.height, band, world, part)
            frame[rows] = got
        assert_bits(frame, ref, f"bands {world}x{band} prj {a.projection}")
    # local row ranges inside a part
    a = jobs_[0][0]
    whole = ea.render(a, g, 3, band=(band, world, 1))
    n = whole.shape[0]
    part = np.concatenate([ea.render(a, g, 3, 0, n // 2, band=(band, world, 1)),
                           ea.render(a, g, 3, n // 2, n, band=(band, world, 1))])
    assert_bits(part, whole, "row ranges of a band part")
    with pytest.raises(ea.EuError):
        ea.render(a, g, 3, band=(6, world, 0))            # not a power of two


@pytest.mark.parametrize("nch", [3, 4])
def test_multi_facet_more_than_sixteen(nch):
    """24 facets: the synopsis walks the facets with run-time loops (up to 64);
    beyond 16 the coordinates are recomputed for the winners instead of kept"""
    os_, gs = [], []
    for k in range(4):
        o6, g6 = facet_set(euo.RECTILINEAR, 48, 48, 70.0 + 5 * k, nch, 1, seed=40 + k)
        os_ += o6
        gs += g6
    a = ea.arguments(ea.SPHERICAL, 160, 80, 360.0, yaw=11, pitch=3, spline_degree=1)
    assert_bits(ea.render(a, gs, nch), jobs.oracle_render(a, os_), f"24 facets nch {nch}")
    a = ea.arguments(ea.SPHERICAL, 96, 48, 360.0, spline_degree=1, twine=2)
    assert_bits(ea.render(a, gs[:17], nch), jobs.oracle_render(a, os_[:17]), f"17 facets twined nch {nch}")
    # 72 > 64: voronoi_syn keeps no per-facet state; alpha compositing beyond the 64 mask bits takes the
    # mask-free form (eu_synopsis_big). Every facet three times over: equal z scores, the earlier one on top
    assert_bits(ea.render(a, gs * 3, nch), jobs.oracle_render(a, os_ * 3), "72 facets")


# ---- targets whose rows share stepper constants (cube faces, unpitched targets) -----

@pytest.mark.parametrize("ypr", [(0, 0, 0), (37.5, 0, 0), (-90, 0, 0), (0, 0.5, 0)])
@pytest.mark.parametrize("degree", [1, 2, 3])
def test_column_invariant_longitude_paths(latlon, ypr, degree):
    """unrotated and yaw-only targets (the source column then depends on the target
    column only - the common lat/lon -> cubemap conversion) next to a slightly
    pitched one, all packed-kernel degrees and channel counts, cropped and banded"""
    for nch in (1, 3, 4):
        o, g = make_pair(euo.SPHERICAL, SRC_W, SRC_H, 360.0, latlon[nch], degree)
        for tprj, tw, th, thfov in [(ea.CUBEMAP, 70, 420, 90.0), (ea.RECTILINEAR, 300, 200, 100.0),
                                    (ea.CYLINDRICAL, 333, 130, 360.0), (ea.BIATAN6, 66, 396, 90.0),
                                    (ea.SPHERICAL, 256, 128, 360.0)]:
            a = ea.arguments(tprj, tw, th, thfov, yaw=ypr[0], pitch=ypr[1], roll=ypr[2], spline_degree=degree)
            assert_bits(ea.render(a, g), jobs.oracle_render(a, o), f"ypr {ypr} deg {degree} nch {nch} trg {tprj}")
    # cropped and band-tiled
    o, g = make_pair(euo.SPHERICAL, SRC_W, SRC_H, 360.0, latlon[3], degree)
    a = ea.arguments(ea.CUBEMAP, 70, 420, 90.0, yaw=ypr[0], pitch=ypr[1], roll=ypr[2], spline_degree=degree,
                     crop=(3, 69, 50, 400))
    assert_bits(ea.render(a, g), jobs.oracle_render(a, o), "unpitched, cropped")
    a = ea.arguments(ea.CUBEMAP, 70, 420, 90.0, yaw=ypr[0], pitch=ypr[1], roll=ypr[2], spline_degree=degree)
    ref = jobs.oracle_render(a, o)
    for part in range(3):
        rows = ea.band_frame_rows(a.height, 8, 3, part)
        assert_bits(ea.render(a, g, 3, band=(8, 3, part)), ref[rows], "unpitched, bands")


# ---- windowed sources (cropped PTO images) ------------------------------------------

@pytest.mark.parametrize("sprj,shfov", [(euo.RECTILINEAR, 90.0), (euo.FISHEYE, 180.0), (euo.CYLINDRICAL, 200.0)])
@pytest.mark.parametrize("window", [(120, 120, 40, 30), (90, 60, 0, 70), (200, 150, 0, 0)])
def test_windowed_source_bit_exact(sprj, shfov, window):
    """facet_spec.window_*: the image is a window of a larger frame
    (envutil_basic.h:447-470); extents follow environment.h:617-633"""
    ww, wh, xo, yo = window
    img = jobs.synth_image(ww, wh, 3, seed=8)
    o = jobs.OracleSource(sprj, 200, 150, shfov, img, 3, yaw=20, pitch=-10, window=window)
    fct = ea.facet_spec(sprj, 200, 150, shfov, nchannels=3, yaw=20, pitch=-10, window=window)
    g = ea.Source.adopt(fct, o.container, 3, o.bc[0], o.bc[1])
    gl = ea.Source.load(fct, img, 3)
    assert_bits(gl.download(), o.container, "windowed source, device set-up")
    for tprj, tw, th, thfov in [(ea.SPHERICAL, 240, 120, 360.0), (ea.RECTILINEAR, 150, 110, 100.0)]:
        for twine in (0, 2):
            a = ea.arguments(tprj, tw, th, thfov, yaw=15, pitch=-5, spline_degree=3, twine=twine)
            ref = jobs.oracle_render(a, o)
            assert_bits(ea.render(a, g), ref, f"windowed source {window} prj {sprj} -> {tprj}")
            assert (ref != 0).any()
    # two windowed facets in one job
    o2 = jobs.OracleSource(sprj, 200, 150, shfov, img, 3, yaw=-70, pitch=5, window=window)
    g2 = ea.Source.adopt(ea.facet_spec(sprj, 200, 150, shfov, nchannels=3, yaw=-70, pitch=5, window=window),
                         o2.container, 3, o2.bc[0], o2.bc[1])
    a = ea.arguments(ea.SPHERICAL, 240, 120, 360.0, spline_degree=3)
    assert_bits(ea.render(a, [g, g2], 3), jobs.oracle_render(a, [o, o2]), "windowed facets, synopsis")


def test_explicit_init_then_tethered_in_a_fresh_process():
    """eu_hip_init() as the FIRST call of a process (bench.py, one process per GPU) must set up
    everything the implicit initialisation does - the sRGB table of the tethered path was once
    missing behind it - and a second, different device is refused"""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np
sys.path.insert(0, "tests")
import envutil_amd as ea, euo, jobs
L = ea.lib()
assert L.eu_hip_init(0) == 0
assert L.eu_hip_init(0) == 0
assert L.eu_hip_init(10 ** 6) != 0
img = jobs.synth_image(128, 64, 3)
o = jobs.OracleSource(euo.SPHERICAL, 128, 64, 360.0, img, 1)
g = ea.Source.adopt(ea.facet_spec(euo.SPHERICAL, 128, 64, 360.0), o.container, 1, o.bc[0], o.bc[1])
a = ea.arguments(ea.SPHERICAL, 90, 45, 360.0, spline_degree=1, tethered=True)
got, ref = ea.render(a, g), jobs.oracle_render(a, o)
assert got.dtype == np.uint32 and (got == ref).all()
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_resident_source_follows_its_facet_spec(latlon):
    """asset_handler semantics (environment.h:84-227): the coefficients stay resident, the
    facet's orientation / hfov / brighten / lens are read fresh for every job"""
    img = jobs.synth_image(128, 128, 3)
    o0, g = make_pair(euo.FISHEYE, 128, 128, 160.0, img, 1)
    a = ea.arguments(ea.SPHERICAL, 160, 80, 360.0, spline_degree=1)
    assert_bits(ea.render(a, g), jobs.oracle_render(a, o0), "as loaded")
    kw = dict(yaw=40.0, pitch=-15.0, roll=20.0, brighten=1.5, lens=dict(a=0.01, b=-0.03, c=0.02))
    o1 = jobs.OracleSource(euo.FISHEYE, 128, 128, 150.0, img, 1, **kw)
    g.update_facet(ea.facet_spec(euo.FISHEYE, 128, 128, 150.0, nchannels=3, **kw))
    assert_bits(ea.render(a, g), jobs.oracle_render(a, o1), "after the facet changed")
    with pytest.raises(ea.EuError):
        g.update_facet(ea.facet_spec(euo.FISHEYE, 64, 64, 150.0, nchannels=3))
