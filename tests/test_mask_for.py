"""--mask_for N (envutil_main.cc:1077-1092, masking.h:70-135, mono_t environment.h:1325-1383): facet N
is painted white and every other facet black at the inner evaluation, so that the job's output is
the mask of where facet N shows in the ordinary rendition. CPU part: properties of the oracle's
restatement. GPU part (marked): the HIP kernels against the oracle, bit for bit."""
import copy

import numpy as np
import pytest

import envutil_amd as ea
import euo
import jobs

VIEWS = [(0.0, 0.0, 0.0), (55.0, 10.0, 5.0), (-60.0, -15.0, 0.0)]


def facets(prj, w, h, hfov, nch, degree, with_gpu, holes=False):
    os_, gs = [], []
    for i, (y, p, r) in enumerate(VIEWS):
        img = jobs.synth_image(w, h, nch, seed=11 + i)
        if nch in (2, 4):
            a = np.ones((h, w), np.float32)
            if holes:
                yy, xx = np.mgrid[0:h, 0:w]
                a = np.clip(1.4 - 1.5 * np.hypot((xx - w / 2) / (w / 2), (yy - h / 2) / (h / 2)), 0, 1).astype(np.float32)
            img[:, :, nch - 1] = a
            img[:, :, :nch - 1] *= a[:, :, None]
        o = jobs.OracleSource(prj, w, h, hfov, img, degree, yaw=y, pitch=p, roll=r)
        os_.append(o)
        if with_gpu:
            gs.append(ea.Source.adopt(ea.facet_spec(prj, w, h, hfov, nchannels=nch, yaw=y, pitch=p, roll=r),
                                      o.container, degree, o.bc[0], o.bc[1]))
    return os_, gs


def set_mask_for(os_, gs, k):
    """--mask_for k; k = -1: an ordinary job"""
    for i, o in enumerate(os_):
        o.s.mask_paint = 0 if k < 0 else (2 if i == k else 1)
    for i, s in enumerate(gs):
        f = copy.copy(s.fct)
        f.masked = -1 if k < 0 else int(i == k)
        s.update_facet(f)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# ------------------------------------------------------------------------------------- CPU

def test_masks_partition_the_covered_area():
    """opaque facets: every output pixel shows exactly one facet (voronoi_syn), so the masks of all
    facets are 0/1 images that add up to the coverage of the ordinary rendition"""
    os_, _ = facets(ea.RECTILINEAR, 96, 72, 70.0, 3, 1, False)
    args = ea.arguments(ea.SPHERICAL, 200, 100, 360.0, spline_degree=1)
    set_mask_for(os_, [], -1)
    plain = jobs.oracle_render(args, os_)
    total = np.zeros(plain.shape[:2], np.float32)
    for k in range(len(os_)):
        set_mask_for(os_, [], k)
        m = jobs.oracle_render(args, os_)
        assert set(np.unique(m)) <= {0.0, 1.0}
        assert (m[..., 0] == m[..., 1]).all() and (m[..., 0] == m[..., 2]).all()
        total += m[..., 0]
        # single facet, painted white: its footprint
        one = jobs.oracle_render(args, os_[k])
        assert ((one[..., 0] == 1) >= (m[..., 0] == 1)).all()
    covered = (plain != 0).any(axis=2)
    assert (total[covered] == 1).all() and (total[~covered] == 0).all()
    set_mask_for(os_, [], -1)


def test_alpha_facets_paint_their_alpha():
    """a facet with alpha: colour = paint * alpha, alpha kept - the white rendition of a single RGBA
    facet has all four channels equal to the ordinary rendition's alpha channel"""
    os_, _ = facets(ea.RECTILINEAR, 80, 60, 60.0, 4, 1, False, holes=True)
    args = ea.arguments(ea.SPHERICAL, 160, 80, 360.0, spline_degree=1)
    o = os_[0]
    o.s.mask_paint = 0
    plain = jobs.oracle_render(args, o)
    o.s.mask_paint = 2
    white = jobs.oracle_render(args, o)
    o.s.mask_paint = 1
    black = jobs.oracle_render(args, o)
    for c in range(4):
        assert (bits(white[..., c]) == bits(plain[..., 3])).all()
    assert (black[..., :3] == 0).all() and (bits(black[..., 3]) == bits(plain[..., 3])).all()
    # mono_t: a two-channel target takes (colour, alpha), a one-channel target colour / alpha
    two = jobs.oracle_render(args, o, nch=2)
    assert (two[..., 0] == 0).all() and (bits(two[..., 1]) == bits(plain[..., 3])).all()
    o.s.mask_paint = 2
    one = jobs.oracle_render(args, o, nch=1)
    assert set(np.unique(one)) <= {0.0, 1.0} and ((one[..., 0] == 1) == (plain[..., 3] != 0)).all()
    o.s.mask_paint = 0


# ------------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
@pytest.mark.parametrize("nch,degree,holes", [(3, 1, False), (4, 1, True), (1, 3, False), (2, 2, True), (4, 3, True)])
def test_mask_jobs_bit_exact(nch, degree, holes):
    os_, gs = facets(ea.RECTILINEAR, 96, 72, 70.0, nch, degree, True, holes)
    for twine in (0, 2):
        args = ea.arguments(ea.SPHERICAL, 256, 128, 360.0, yaw=5, pitch=-3, spline_degree=degree, twine=twine)
        for k in range(len(os_)):
            set_mask_for(os_, gs, k)
            ref = jobs.oracle_render(args, os_)
            got = ea.render(args, gs)
            assert (bits(got) == bits(ref)).all(), (nch, degree, twine, k)
            # a single masked facet (the packed kernels must not take the job)
            ref1 = jobs.oracle_render(args, os_[k])
            got1 = ea.render(args, gs[k])
            assert (bits(got1) == bits(ref1)).all(), (nch, degree, twine, k, "single")
            assert got1.max() > 0
    set_mask_for(os_, gs, -1)
    assert (bits(ea.render(args, gs)) == bits(jobs.oracle_render(args, os_))).all()


@pytest.mark.gpu
@pytest.mark.parametrize("nch,out_n", [(3, 1), (3, 2), (4, 2), (4, 1), (1, 2), (2, 1)])
def test_mask_jobs_with_mono(nch, out_n):
    os_, gs = facets(ea.FISHEYE, 80, 80, 150.0, nch, 1, True, holes=True)
    args = ea.arguments(ea.SPHERICAL, 200, 100, 360.0, spline_degree=1)
    for k in (0, 2):
        set_mask_for(os_, gs, k)
        ref = jobs.oracle_render(args, os_, nch=out_n)
        got = ea.render(args, gs, out_n)
        assert (bits(got) == bits(ref)).all(), (nch, out_n, k)
        ref1 = jobs.oracle_render(args, os_[k], nch=out_n)
        got1 = ea.render(args, gs[k], out_n)
        assert (bits(got1) == bits(ref1)).all(), (nch, out_n, k, "single")


@pytest.mark.gpu
def test_mask_of_a_cubemap_facet_and_refusals():
    faces = jobs.synth_cubefaces(48, 4)
    faces[..., 3] = 1.0
    faces[10:30, 10:30, 3] = 0.5
    faces[10:30, 10:30, :3] *= 0.5
    o = jobs.OracleSource(euo.CUBEMAP, 48, 288, 90.0, faces, 2, masked=1)
    s = ea.Source.adopt(ea.facet_spec(ea.CUBEMAP, 48, 288, 90.0, nchannels=4, masked=1), o.container, 2)
    args = ea.arguments(ea.SPHERICAL, 192, 96, 360.0, spline_degree=2)
    assert (bits(ea.render(args, s)) == bits(jobs.oracle_render(args, o))).all()
    # mono_t knows one- and two-channel targets only: the reference asserts, the library refuses
    s3 = ea.Source.load(ea.facet_spec(ea.SPHERICAL, 64, 32, 360.0, nchannels=3, masked=1), jobs.synth_image(64, 32, 3), 1)
    with pytest.raises(ea.EuError):
        ea.render(args, s3, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_random_mask_jobs_bit_identical(seed):
    """--mask_for over the randomised facets of the round-2 fuzz (all mounts, lens polynomials, translated facets,
    alpha holes, brighten), twining, both synopsis modes, every target projection, mono_t where channel counts differ"""
    import test_gpu_fuzz_round2 as F
    rng = np.random.default_rng(91000 + seed)
    for k in range(5):
        out_n = int(rng.integers(1, 5))
        degree = int(rng.choice([0, 1, 1, 2, 3]))
        nf = int(rng.choice([1, 2, 3, 4]))
        # facets of the target's channel count, or - for one- and two-channel targets - of any (mono_t)
        fn = [out_n if (out_n > 2 or rng.random() < 0.5) else int(rng.integers(1, 5)) for _ in range(nf)]
        facets = [F.draw_facet(rng, fn[i], degree, seed * 1000 + 10 * k + i) for i in range(nf)]
        os_, gs = [f[0] for f in facets], [f[1] for f in facets]
        which = int(rng.integers(nf))
        set_mask_for(os_, gs, which)
        kw = dict(spline_degree=degree, twine=int(rng.choice([0, 0, 2, 3])),
                  synopsis=str(rng.choice(["panorama", "panorama", "hdr_merge"])))
        tprj = F.GENERIC_TRG[rng.integers(len(F.GENERIC_TRG))]
        if tprj in (ea.CUBEMAP, ea.BIATAN6):
            tw = int(rng.integers(8, 40)); th, thf = 6 * tw, 90.0
        else:
            tw, th = int(rng.integers(8, 160)), int(rng.integers(8, 80))
            thf = float(rng.uniform(30.0, {ea.RECTILINEAR: 130.0, ea.STEREOGRAPHIC: 280.0}.get(tprj, 360.0)))
        a = ea.arguments(tprj, tw, th, thf, yaw=float(rng.uniform(-180, 180)), pitch=float(rng.uniform(-60, 60)),
                         roll=float(rng.uniform(-30, 30)), **kw)
        what = f"seed {seed} job {k}: facets {fn} -> {out_n} channels, mask_for {which}, deg {degree} {kw} target {tprj} {tw}x{th}"
        got = ea.render(a, gs, out_n)
        ref = jobs.oracle_render(a, os_, nch=out_n)
        same = bits(got) == bits(ref)
        assert same.all(), f"{what}: {int((~same).sum())} of {same.size} words differ"
        for g in gs:
            g.release()
