"""Facets with PTO translation parameters (TrX, TrY, TrZ, Tpy, Tpp): the reference steps them with
generic_stepper over tf_ex_facet (envutil_payload.cc:1628-1885, :2095-2110, :2145-2158;
geometry.h:1850-1941). CPU part: properties of the oracle's restatement; GPU part (marked): the HIP
kernels (single-facet general kernel, multi-facet kernel) against the oracle, bit for bit."""
import math

import numpy as np
import pytest

import envutil_amd as ea
import euo
import jobs

TARGETS = [(ea.SPHERICAL, 200, 100, 360.0), (ea.CYLINDRICAL, 180, 90, 200.0), (ea.RECTILINEAR, 150, 100, 80.0),
           (ea.STEREOGRAPHIC, 120, 120, 200.0), (ea.FISHEYE, 120, 120, 170.0), (ea.CUBEMAP, 48, 288, 90.0),
           (ea.BIATAN6, 40, 240, 90.0)]
TR = dict(x=0.12, y=-0.07, z=0.05, tp_y=4.0, tp_p=-3.0, tp_r=1.5)


def facet(prj, w, h, hfov, nch, degree, translation, gpu, seed=2, **kw):
    img = jobs.synth_image(w, h, nch, seed=seed)
    o = jobs.OracleSource(prj, w, h, hfov, img, degree, translation=translation, **kw)
    g = None
    if gpu:
        g = ea.Source.adopt(ea.facet_spec(prj, w, h, hfov, nchannels=nch, translation=translation, **kw),
                            o.container, degree, o.bc[0], o.bc[1])
    return o, g


# ---- properties of the restatement (no GPU) --------------------------------------------------------

def test_vanishing_translation_is_the_plain_facet():
    plain, _ = facet(euo.RECTILINEAR, 120, 90, 70.0, 3, 1, None, False, yaw=5)
    for tprj, tw, th, hf in TARGETS:
        a = ea.arguments(tprj, tw, th, hf, yaw=3, pitch=-2, spline_degree=1)
        ref = jobs.oracle_render(a, plain)
        tiny, _ = facet(euo.RECTILINEAR, 120, 90, 70.0, 3, 1, dict(x=1e-7), False, yaw=5)
        got = jobs.oracle_render(a, tiny)
        inner = (ref[:, :, 0] != 0) & (got[:, :, 0] != 0)
        assert inner.mean() > 0.02
        assert np.abs(got - ref)[inner].max() < 2e-5, tprj


def test_translation_along_the_view_axis_is_a_change_of_scale():
    """camera moved by tz along the facet's axis, translation plane facing it: the ray (x, y, z) picks
    up at (x/z, y/z) / (1 - tz), which is what the untranslated facet with extent tan(hfov/2) * (1 - tz)
    shows"""
    tz, hfov = 0.2, 70.0
    moved, _ = facet(euo.RECTILINEAR, 160, 120, hfov, 3, 1, dict(z=tz), False)
    hf2 = math.degrees(2.0 * math.atan(math.tan(math.radians(hfov) / 2.0) * (1.0 - tz)))
    scaled, _ = facet(euo.RECTILINEAR, 160, 120, hf2, 3, 1, None, False)
    a = ea.arguments(ea.RECTILINEAR, 120, 90, 50.0, spline_degree=1)
    got, ref = jobs.oracle_render(a, moved), jobs.oracle_render(a, scaled)
    assert (ref[:, :, 0] != 0).all()
    np.testing.assert_allclose(got, ref, rtol=0, atol=3e-5)


def test_rays_behind_the_translation_plane_see_nothing():
    o, _ = facet(euo.RECTILINEAR, 100, 80, 80.0, 3, 1, dict(x=0.1), False)
    a = ea.arguments(ea.SPHERICAL, 240, 120, 360.0, spline_degree=1)
    out = jobs.oracle_render(a, o)
    assert np.isfinite(out).all()
    assert (out[:, :40] == 0).all() and (out[:, 200:] == 0).all()     # looking backwards
    assert (out[40:80, 100:140, 0] != 0).all()


# ---- HIP against the oracle ------------------------------------------------------------------------

def assert_bits(got, ref, what):
    d = jobs.bits(got) != jobs.bits(ref)
    assert not d.any(), f"{what}: {int(d.sum())} of {d.size} values differ, first at {np.argwhere(d)[0]}"


@pytest.mark.gpu
@pytest.mark.parametrize("tprj,tw,th,hf", TARGETS)
@pytest.mark.parametrize("twine", [0, 2])
def test_single_translated_facet_bit_exact(tprj, tw, th, hf, twine):
    for sprj, sw, sh, shf, deg in [(euo.RECTILINEAR, 120, 90, 70.0, 1), (euo.FISHEYE, 100, 100, 150.0, 3)]:
        o, g = facet(sprj, sw, sh, shf, 3, deg, TR, True, yaw=12, pitch=-5, roll=3, brighten=1.2)
        a = ea.arguments(tprj, tw, th, hf, yaw=20, pitch=7, roll=-4, spline_degree=deg, twine=twine)
        assert_bits(ea.render(a, g, 3), jobs.oracle_render(a, o), f"translated facet {sprj} -> target {tprj} twine {twine}")
        # stage 1: the rays themselves (generic_stepper's output)
        assert_bits(ea.render(a, g, 3, stage=1), jobs.oracle_render(a, o, stage=1), "rays")


@pytest.mark.gpu
def test_translation_without_plane_rotation_and_with_lens(tmp_path):
    o, g = facet(euo.RECTILINEAR, 140, 100, 60.0, 4, 1, dict(x=-0.2, z=-0.1), True, yaw=-8,
                 lens=dict(a=0.01, b=-0.02, c=0.01, h=3.0, v=-2.0))
    a = ea.arguments(ea.SPHERICAL, 260, 130, 360.0, spline_degree=1)
    assert_bits(ea.render(a, g, 4), jobs.oracle_render(a, o), "translation + lens")


@pytest.mark.gpu
@pytest.mark.parametrize("nch", [3, 4])
@pytest.mark.parametrize("synopsis", ["panorama", "hdr_merge"])
def test_mosaic_of_translated_and_plain_facets_bit_exact(nch, synopsis):
    """a PTO mosaic: some facets moved sideways (translated), others plain - per facet the reference picks
    generic_stepper or the target's own stepper (envutil_payload.cc:2145-2158)"""
    os_, gs = [], []
    for k in range(5):
        tr = None if k % 2 == 0 else dict(x=0.15 * (k - 2), y=0.05 * k, z=0.03 * k, tp_y=2.0 * k, tp_p=-1.0 * k)
        o, g = facet(euo.RECTILINEAR, 100, 80, 65.0, nch, 1, tr, True, seed=10 + k, yaw=25.0 * (k - 2), pitch=3.0 * k,
                     brighten=1.0 + 0.1 * k)
        os_.append(o)
        gs.append(g)
    for tprj, tw, th, hf in TARGETS[:3]:
        for twine in (0, 2):
            a = ea.arguments(tprj, tw, th, hf, yaw=5, spline_degree=1, twine=twine, synopsis=synopsis)
            assert_bits(ea.render(a, gs, nch), jobs.oracle_render(a, os_), f"mosaic target {tprj} twine {twine} {synopsis}")


# ---- --single: the target recreates a facet (inverse lens correction, inverse translation) -------------

def single_job(prj, w, h, hfov, lens, translation, gpu, nfacets=3, nch=3, degree=1, **kw):
    """the facet to recreate (its pixels are not used: only its geometry) and a few source facets around it"""
    fspec = ea.facet_spec(prj, w, h, hfov, nchannels=nch, yaw=4.0, pitch=-2.0, roll=1.0, lens=lens, translation=translation)
    single_o = jobs.OracleSource(prj, w, h, hfov, jobs.synth_image(w, h, nch, seed=1), degree, yaw=4.0, pitch=-2.0, roll=1.0,
                                 lens=lens, translation=translation)
    a = ea.arguments.for_single(fspec, spline_degree=degree, **kw)
    a.single_oracle = single_o
    os_, gs = [], []
    for k in range(nfacets):
        tr = None if k != 1 else dict(x=0.1, y=0.05, z=-0.04, tp_y=3.0, tp_p=2.0)
        o, g = facet(euo.RECTILINEAR, 110, 90, 75.0, nch, degree, tr, gpu, seed=30 + k, yaw=4.0 + 20.0 * (k - 1), pitch=-2.0 + 3.0 * k,
                     brighten=1.0 + 0.1 * k)
        os_.append(o)
        gs.append(g)
    return a, os_, gs


def test_single_without_lens_or_translation_is_an_ordinary_job():
    a, os_, _ = single_job(euo.RECTILINEAR, 120, 90, 70.0, None, None, False, nfacets=1)
    b = ea.arguments(ea.RECTILINEAR, 120, 90, 70.0, yaw=4.0, pitch=-2.0, roll=1.0, spline_degree=1)
    assert (jobs.bits(jobs.oracle_render(a, os_)) == jobs.bits(jobs.oracle_render(b, os_))).all()


def test_single_undoes_the_lens_correction_of_the_facet_it_recreates():
    """a facet seen through its own lens parameters, recreated with --single from itself, is the facet:
    the inverse planar transformation of the target and the forward one of the source cancel"""
    lens = dict(a=0.01, b=-0.03, c=0.02, h=0.02, v=-0.01)
    img = jobs.synth_image(160, 120, 3, seed=7)
    src = jobs.OracleSource(euo.RECTILINEAR, 160, 120, 70.0, img, 1, yaw=4.0, pitch=-2.0, roll=1.0, lens=lens)
    fspec = ea.facet_spec(euo.RECTILINEAR, 160, 120, 70.0, yaw=4.0, pitch=-2.0, roll=1.0, lens=lens)
    a = ea.arguments.for_single(fspec, spline_degree=1)
    a.single_oracle = src
    out = jobs.oracle_render(a, src)
    inner = out[10:-10, 10:-10]
    assert (inner[:, :, 0] != 0).all()
    np.testing.assert_allclose(inner, img[10:-10, 10:-10], rtol=0, atol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("lens,translation", [
    (dict(a=0.01, b=-0.03, c=0.02), None),
    (dict(h=0.03, v=-0.02, g=0.01, t=-0.02), None),
    (dict(a=0.02, b=0.0, c=-0.01, h=0.01, v=0.02, g=0.005, t=0.004), dict(x=0.08, y=-0.03, z=0.05, tp_y=2.0, tp_p=-1.0)),
    (None, dict(x=-0.1, y=0.02, z=0.0)),
])
@pytest.mark.parametrize("prj,w,h,hfov", [(euo.RECTILINEAR, 130, 100, 70.0), (euo.FISHEYE, 110, 110, 140.0),
                                          (euo.SPHERICAL, 160, 80, 120.0)])
def test_single_jobs_bit_exact(prj, w, h, hfov, lens, translation):
    for nf, twine, syn in [(1, 0, "panorama"), (3, 0, "panorama"), (3, 2, "panorama"), (3, 0, "hdr_merge")]:
        a, os_, gs = single_job(prj, w, h, hfov, lens, translation, True, nfacets=nf, twine=twine, synopsis=syn)
        assert_bits(ea.render(a, gs, 3), jobs.oracle_render(a, os_), f"--single {prj} lens {lens} tr {translation} nf {nf} twine {twine} {syn}")


@pytest.mark.gpu
def test_single_with_a_lens_polynomial_that_has_no_inverse_is_an_error():
    a, os_, gs = single_job(euo.RECTILINEAR, 100, 80, 70.0, dict(a=-0.3, b=0.0, c=0.0), None, True, nfacets=1)
    with pytest.raises(Exception):
        ea.render(a, gs, 3)
