"""args.synopsis == "hdr_merge": _hdr_merge_syn (envutil_payload.cc:1325-1626), the quality-weighted
sum of ALL facets of a job. CPU part: the oracle's restatement against an independent numpy
model built from single-facet renders; GPU part (marked): the HIP multi-facet kernel against
the oracle, bit for bit."""
import math

import numpy as np
import pytest

import envutil_amd as ea
import euo
import jobs


def bracket(prj, w, h, hfov, nch, degree, brightens, seed=3, with_gpu=False, alpha_holes=False):
    """an exposure bracket: the same scene seen by the same camera at several exposures. A longer
    exposure's image is brighter (values clip at 1) and gets a SMALLER brighten factor."""
    scene = jobs.synth_image(w, h, nch, seed=seed)
    ncol = nch - 1 if nch in (2, 4) else nch
    scene[:, :, :ncol] = scene[:, :, :ncol] ** 2 * 1.5          # some range: 0 .. 1.5
    os_, gs = [], []
    for i, b in enumerate(brightens):
        img = scene.copy()
        img[:, :, :ncol] = np.clip(img[:, :, :ncol] / np.float32(b), 0.0, 1.0)
        if nch in (2, 4):
            a = np.ones((h, w), np.float32)
            if alpha_holes:
                yy, xx = np.mgrid[0:h, 0:w]
                a = np.clip(1.5 - 1.6 * np.hypot((xx - w / 2) / (w / 2), (yy - h / 2) / (h / 2)), 0.0, 1.0).astype(np.float32)
                a[(xx + 3 * i) % 11 == 0] = 0.0
            img[:, :, nch - 1] = a
            img[:, :, :ncol] *= a[:, :, None]
        o = jobs.OracleSource(prj, w, h, hfov, img, degree, yaw=0.5 * i, pitch=-0.25 * i, brighten=b)
        os_.append(o)
        if with_gpu:
            gs.append(ea.Source.adopt(ea.facet_spec(prj, w, h, hfov, nchannels=nch, yaw=0.5 * i, pitch=-0.25 * i, brighten=b),
                                      o.container, degree, o.bc[0], o.bc[1]))
    return os_, gs


def f32(x):
    return np.asarray(x, np.float32)


def numpy_hdr_merge(px, brightens, nch):
    """the synopsis from the facets' pixels px[f] (H, W, nch), float32 operation by operation"""
    lowest, highest, low, high = np.float32(100000.0), np.float32(-1.0), -1, -1
    for f, b in enumerate(brightens):
        b = np.float32(b)
        if b < lowest:
            lowest, low = b, f
        if b > highest:
            highest, high = b, f
    alpha = nch in (2, 4)
    ncol = nch - 1 if alpha else nch
    trg = np.zeros_like(px[0])
    qsum = np.zeros(px[0].shape[:2], np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        for f, v in enumerate(px):
            opt = np.float32(0.5) * np.float32(brightens[f])
            grey = v[:, :, 0] if ncol == 1 else np.maximum(v[:, :, 0], np.maximum(v[:, :, 1], v[:, :, 2]))
            large = grey > opt
            dist = np.abs(opt - grey)
            if f == low:
                dist = np.where(~large, np.float32(0), dist)
            elif f == high:
                dist = np.where(large, np.float32(0), dist)
            q = f32(f32(opt - dist) / f32(opt * opt))
            if alpha:
                a = v[:, :, nch - 1]
                q = f32(a * q)
                for c in range(ncol):
                    d = np.where(a > np.float32(0.000001), f32(v[:, :, c] / a), np.float32(0))
                    trg[:, :, c] = f32(trg[:, :, c] + f32(d * q))
                trg[:, :, nch - 1] = np.maximum(trg[:, :, nch - 1], a)
            else:
                for c in range(ncol):
                    trg[:, :, c] = f32(trg[:, :, c] + f32(v[:, :, c] * q))
            qsum = f32(qsum + q)
        for c in range(ncol):
            t = f32(trg[:, :, c] / qsum)
            t = np.where(qsum > 0, t, np.float32(0))
            if alpha:
                t = f32(t * trg[:, :, nch - 1])
            trg[:, :, c] = t
    return trg


@pytest.mark.parametrize("nch", [1, 2, 3, 4])
def test_oracle_hdr_merge_against_numpy_model(nch):
    """degree 0 (nearest texel) keeps the facets' pixels insensitive to the last bits of the ray, so
    the single-facet renders (other stepper normalisation) give the pixels the synopsis sees"""
    brightens = (4.0, 1.0, 0.25)
    os_, _ = bracket(euo.RECTILINEAR, 96, 64, 70.0, nch, 0, brightens, alpha_holes=True)
    for o in os_:
        o.s.yaw = o.s.pitch = 0.0                      # identical cameras: every facet sees every pixel alike
    a = ea.arguments(ea.RECTILINEAR, 60, 40, 50.0, spline_degree=0, synopsis="hdr_merge")
    a1 = ea.arguments(ea.RECTILINEAR, 60, 40, 50.0, spline_degree=0)
    px = [jobs.oracle_render(a1, o) for o in os_]
    ref = jobs.oracle_render(a, os_)
    model = numpy_hdr_merge(px, brightens, nch)
    same = jobs.bits(ref) == jobs.bits(model)
    assert same.mean() > 0.995, f"{(~same).sum()} of {same.size} values differ"
    np.testing.assert_allclose(ref, model, rtol=2e-6, atol=1e-7)
    assert np.isfinite(ref).all() and ref.max() > 0.5


def test_oracle_hdr_merge_of_identical_facets_is_the_facet():
    img = jobs.synth_image(80, 60, 3, seed=8)
    os_ = [jobs.OracleSource(euo.RECTILINEAR, 80, 60, 80.0, img, 1) for _ in range(3)]
    a = ea.arguments(ea.RECTILINEAR, 50, 30, 60.0, spline_degree=1, synopsis="hdr_merge")
    one = jobs.oracle_render(ea.arguments(ea.RECTILINEAR, 50, 30, 60.0, spline_degree=1, synopsis="hdr_merge"), os_[:2])
    three = jobs.oracle_render(a, os_)
    np.testing.assert_allclose(one, three, rtol=1e-5, atol=1e-6)


def test_arguments_refuse_an_unknown_synopsis():
    with pytest.raises(ValueError):
        ea.arguments(ea.SPHERICAL, 64, 32, 360.0, synopsis="median")


# ---- HIP against the oracle ---------------------------------------------------------------------

def assert_bits(got, ref, what):
    d = jobs.bits(got) != jobs.bits(ref)
    assert not d.any(), f"{what}: {int(d.sum())} of {d.size} values differ, first at {np.argwhere(d)[0]}"


@pytest.mark.gpu
@pytest.mark.parametrize("nch", [1, 2, 3, 4])
@pytest.mark.parametrize("degree,twine", [(1, 0), (3, 0), (1, 2)])
def test_hdr_merge_bracket_bit_exact(nch, degree, twine):
    os_, gs = bracket(euo.RECTILINEAR, 120, 90, 75.0, nch, degree, (4.0, 1.0, 0.25), with_gpu=True,
                      alpha_holes=True)
    a = ea.arguments(ea.RECTILINEAR, 150, 100, 90.0, yaw=2, pitch=1, roll=-3, spline_degree=degree, twine=twine,
                     synopsis="hdr_merge")
    got, ref = ea.render(a, gs, nch), jobs.oracle_render(a, os_)
    assert_bits(got, ref, f"hdr_merge nch {nch} degree {degree} twine {twine}")
    assert (ref[:, :, 0] != 0).mean() > 0.3          # the target is wider than the facets: holes too
    # the same facets as a panorama give something else: the synopsis switch reaches the kernel
    b = ea.arguments(ea.RECTILINEAR, 150, 100, 90.0, yaw=2, pitch=1, roll=-3, spline_degree=degree, twine=twine)
    assert (jobs.bits(ea.render(b, gs, nch)) != jobs.bits(got)).any()


@pytest.mark.gpu
@pytest.mark.parametrize("nch", [3, 4])
def test_hdr_merge_of_facets_looking_elsewhere(nch):
    """six fisheye facets with lens polynomial, different orientations and brighten: most pixels see
    one or two facets, the others contribute zero pixels (and, for the LOW facet, their quality)"""
    import test_gpu_parity as tp
    os_, gs = tp.facet_set(euo.FISHEYE, 96, 96, 130.0, nch, 1, dict(a=0.01, b=-0.03, c=0.02))
    a = ea.arguments(ea.SPHERICAL, 300, 150, 360.0, yaw=10, pitch=4, roll=-2, spline_degree=1, synopsis="hdr_merge")
    assert_bits(ea.render(a, gs, nch), jobs.oracle_render(a, os_), f"hdr_merge six facets nch {nch}")


@pytest.mark.gpu
def test_hdr_merge_mixed_channel_counts_and_many_facets():
    import test_gpu_parity as tp
    sets = {n: tp.facet_set(euo.RECTILINEAR, 72, 72, 95.0, n, 1, seed=21) for n in (3, 4)}
    os_ = [sets[(3, 4)[i % 2]][0][i] for i in range(6)]
    gs = [sets[(3, 4)[i % 2]][1][i] for i in range(6)]
    a = ea.arguments(ea.SPHERICAL, 180, 90, 360.0, yaw=20, pitch=-6, roll=3, spline_degree=1, synopsis="hdr_merge")
    for out_n in (3, 4):
        assert_bits(ea.render(a, gs, out_n), jobs.oracle_render(a, os_, nch=out_n), f"hdr_merge mixed -> {out_n}")
    # 20 facets (beyond the 16 whose coordinates the panorama kernels keep in LDS)
    os20, gs20 = [], []
    for k in range(20):
        img = jobs.synth_image(48, 48, 3, seed=100 + k)
        o = jobs.OracleSource(euo.RECTILINEAR, 48, 48, 60.0, img, 1, yaw=18.0 * k, pitch=10.0 * ((k % 3) - 1),
                              brighten=0.5 + 0.1 * k)
        os20.append(o)
        gs20.append(ea.Source.adopt(ea.facet_spec(ea.RECTILINEAR, 48, 48, 60.0, yaw=18.0 * k, pitch=10.0 * ((k % 3) - 1),
                                                  brighten=0.5 + 0.1 * k), o.container, 1, o.bc[0], o.bc[1]))
    a = ea.arguments(ea.SPHERICAL, 200, 100, 360.0, spline_degree=1, synopsis="hdr_merge")
    assert_bits(ea.render(a, gs20, 3), jobs.oracle_render(a, os20), "hdr_merge 20 facets")


@pytest.mark.gpu
@pytest.mark.parametrize("synopsis,nch", [("panorama", 3), ("panorama", 1), ("hdr_merge", 3), ("hdr_merge", 4)])
def test_more_than_sixty_four_facets(synopsis, nch):
    """voronoi_syn and hdr_merge keep no per-facet state on the device: 90 facets in one job (alpha
    compositing - voronoi_syn_plus - stays at 64: one mask bit per facet)"""
    os90, gs90 = [], []
    for k in range(90):
        img = jobs.synth_image(40, 40, nch, seed=300 + k)
        kw = dict(yaw=4.0 * k, pitch=25.0 * math.sin(k), brighten=0.6 + 0.01 * k)
        o = jobs.OracleSource(euo.RECTILINEAR, 40, 40, 40.0, img, 1, **kw)
        os90.append(o)
        gs90.append(ea.Source.adopt(ea.facet_spec(ea.RECTILINEAR, 40, 40, 40.0, nchannels=nch, **kw), o.container, 1, o.bc[0], o.bc[1]))
    a = ea.arguments(ea.SPHERICAL, 240, 120, 360.0, spline_degree=1, synopsis=synopsis)
    assert_bits(ea.render(a, gs90, nch), jobs.oracle_render(a, os90), f"90 facets {synopsis} nch {nch}")


@pytest.mark.gpu
@pytest.mark.parametrize("nch,twine", [(4, 0), (2, 0), (4, 2)])
def test_alpha_compositing_of_more_than_sixty_four_facets(nch, twine):
    """voronoi_syn_plus beyond the 64 mask bits: the mask-free form (eu_synopsis_big) finds every layer by a
    pass over all facets; 70 overlapping facets with feathered alpha, some with equal z scores (same step)"""
    os70, gs70 = [], []
    for k in range(70):
        w = 36
        img = jobs.synth_image(w, w, nch, seed=500 + k)
        yy, xx = np.mgrid[0:w, 0:w]
        a = np.clip(1.5 - 1.6 * np.hypot((xx - w / 2) / (w / 2), (yy - w / 2) / (w / 2)), 0.0, 1.0).astype(np.float32)
        img[:, :, nch - 1] = a
        img[:, :, :nch - 1] *= a[:, :, None]
        kw = dict(yaw=5.2 * k, pitch=20.0 * math.sin(1.7 * k), roll=3.0 * k, brighten=0.7 + 0.01 * k)
        hf = 50.0 if k % 3 else 50.0 + 0.5 * (k % 7)
        o = jobs.OracleSource(euo.RECTILINEAR, w, w, hf, img, 1, **kw)
        os70.append(o)
        gs70.append(ea.Source.adopt(ea.facet_spec(ea.RECTILINEAR, w, w, hf, nchannels=nch, **kw), o.container, 1, o.bc[0], o.bc[1]))
    a = ea.arguments(ea.SPHERICAL, 200, 100, 360.0, yaw=7, spline_degree=1, twine=twine)
    got, ref = ea.render(a, gs70, nch), jobs.oracle_render(a, os70)
    assert_bits(got, ref, f"70 facets with alpha nch {nch} twine {twine}")
    assert (ref[:, :, nch - 1] > 0).mean() > 0.3
    # and the same 64 of them through the mask-based kernel agree with the oracle too (the boundary)
    assert_bits(ea.render(a, gs70[:64], nch), jobs.oracle_render(a, os70[:64]), "64 facets with alpha")
