"""Randomised differential test of what round 2 added to the path: facets with PTO translation,
`--single` targets (inverse lens polynomial / shift / shear, inverse translation), the hdr_merge
synopsis, jobs of many facets - drawn at random, mixed with twining, crops, channel adaption and
tethered output, rendered by the HIP library and by the oracle; every float / word identical.
EU_FUZZ2_SEEDS=a:b runs seeds a..b-1 instead of the default few."""
import os

import numpy as np
import pytest

import envutil_amd as ea
import euo
import jobs

pytestmark = pytest.mark.gpu

FACET_PRJ = [euo.SPHERICAL, euo.CYLINDRICAL, euo.RECTILINEAR, euo.STEREOGRAPHIC, euo.FISHEYE]
GENERIC_TRG = [ea.SPHERICAL, ea.CYLINDRICAL, ea.RECTILINEAR, ea.STEREOGRAPHIC, ea.FISHEYE, ea.CUBEMAP, ea.BIATAN6]
_s = os.environ.get("EU_FUZZ2_SEEDS")
SEEDS = range(*[int(v) for v in _s.split(":")]) if _s else range(6)


def hfov_for(rng, prj):
    return float(rng.uniform(40.0, {euo.RECTILINEAR: 120.0, euo.STEREOGRAPHIC: 220.0, euo.FISHEYE: 250.0,
                                    euo.CYLINDRICAL: 300.0, euo.SPHERICAL: 300.0}[prj]))


def draw_lens(rng):
    lens = {}
    if rng.random() < 0.6:
        lens.update(a=float(rng.uniform(-0.015, 0.015)), b=float(rng.uniform(-0.03, 0.03)), c=float(rng.uniform(-0.02, 0.02)))
    if rng.random() < 0.5:
        lens.update(h=float(rng.uniform(-0.03, 0.03)), v=float(rng.uniform(-0.03, 0.03)))
    if rng.random() < 0.3:
        lens.update(g=float(rng.uniform(-0.02, 0.02)), t=float(rng.uniform(-0.02, 0.02)))
    return lens or None


def draw_translation(rng):
    tr = dict(x=float(rng.uniform(-0.3, 0.3)), y=float(rng.uniform(-0.2, 0.2)), z=float(rng.uniform(-0.2, 0.2)))
    if rng.random() < 0.6:
        tr.update(tp_y=float(rng.uniform(-20, 20)), tp_p=float(rng.uniform(-15, 15)))
    if rng.random() < 0.2:
        tr["tp_r"] = float(rng.uniform(-10, 10))
    return tr


def draw_facet(rng, nch, degree, seed, allow_translation=True):
    prj = FACET_PRJ[rng.integers(len(FACET_PRJ))]
    w, h = int(rng.integers(24, 120)), int(rng.integers(24, 100))
    kw = dict(yaw=float(rng.uniform(-60, 60)), pitch=float(rng.uniform(-40, 40)), roll=float(rng.uniform(-20, 20)),
              brighten=float(rng.choice([1.0, 0.5, 2.0, 1.3])))
    if prj != euo.SPHERICAL and rng.random() < 0.3:
        kw["lens"] = draw_lens(rng)
    if allow_translation and rng.random() < 0.5:
        kw["translation"] = draw_translation(rng)
    hf = hfov_for(rng, prj)
    img = jobs.synth_image(w, h, nch, seed=seed)
    if nch in (2, 4) and rng.random() < 0.7:
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.clip(1.4 - 1.5 * np.hypot((xx - w / 2) / (w / 2), (yy - h / 2) / (h / 2)), 0.0, 1.0).astype(np.float32)
        img[:, :, nch - 1] = a
        img[:, :, :nch - 1] *= a[:, :, None]
    o = jobs.OracleSource(prj, w, h, hf, img, degree, **kw)
    g = ea.Source.adopt(ea.facet_spec(prj, w, h, hf, nchannels=nch, **kw), o.container, degree, o.bc[0], o.bc[1])
    return o, g


@pytest.mark.parametrize("seed", SEEDS)
def test_random_round2_jobs_bit_identical(seed):
    rng = np.random.default_rng(7000 + seed)
    for k in range(5):
        nch = int(rng.integers(1, 5))
        degree = int(rng.choice([0, 1, 1, 2, 3, 5]))
        nf = int(rng.choice([1, 1, 2, 3, 5]))
        facets = [draw_facet(rng, nch if rng.random() < 0.85 else int(rng.integers(1, 5)), degree, seed * 1000 + 10 * k + i)
                  for i in range(nf)]
        os_, gs = [f[0] for f in facets], [f[1] for f in facets]
        kw = dict(spline_degree=degree, twine=int(rng.choice([0, 0, 2, 3])),
                  synopsis=str(rng.choice(["panorama", "panorama", "hdr_merge"])))
        if rng.random() < 0.15:
            kw["tethered"] = True
        single = rng.random() < 0.4
        if single:
            # the target recreates a facet of its own geometry (not one of the sources: only its parameters count)
            sprj = FACET_PRJ[rng.integers(len(FACET_PRJ))]
            w, h = int(rng.integers(20, 140)), int(rng.integers(20, 100))
            skw = dict(yaw=float(rng.uniform(-30, 30)), pitch=float(rng.uniform(-20, 20)), roll=float(rng.uniform(-10, 10)))
            lens = draw_lens(rng) if rng.random() < 0.7 else None
            if lens and any(k_ in lens for k_ in "abc"):
                # keep the polynomial invertible over the facet (the reference asserts otherwise)
                lens.update(a=lens["a"] / 2, b=lens["b"] / 2, c=lens["c"] / 2)
            tr = draw_translation(rng) if rng.random() < 0.5 else None
            hf = hfov_for(rng, sprj)
            fspec = ea.facet_spec(sprj, w, h, hf, nchannels=nch, lens=lens, translation=tr, **skw)
            a = ea.arguments.for_single(fspec, **kw)
            a.single_oracle = jobs.OracleSource(sprj, w, h, hf, jobs.synth_image(w, h, nch, seed=5), degree, lens=lens,
                                                translation=tr, **skw)
            tdesc = f"single {sprj} {w}x{h} fov {hf:.1f} lens {lens} tr {tr}"
        else:
            tprj = GENERIC_TRG[rng.integers(len(GENERIC_TRG))]
            if tprj in (ea.CUBEMAP, ea.BIATAN6):
                tw = int(rng.integers(8, 40)); th, thf = 6 * tw, 90.0
            else:
                tw, th = int(rng.integers(8, 180)), int(rng.integers(8, 90))
                thf = float(rng.uniform(30.0, {ea.RECTILINEAR: 130.0, ea.STEREOGRAPHIC: 280.0}.get(tprj, 360.0)))
            if rng.random() < 0.25 and tw >= 8 and th >= 8:
                x0, y0 = int(rng.integers(0, tw // 2)), int(rng.integers(0, th // 2))
                kw["crop"] = (x0, int(rng.integers(x0 + 1, tw + 1)), y0, int(rng.integers(y0 + 1, th + 1)))
            a = ea.arguments(tprj, tw, th, thf, yaw=float(rng.uniform(-180, 180)), pitch=float(rng.uniform(-60, 60)),
                             roll=float(rng.uniform(-30, 30)), **kw)
            tdesc = f"target {tprj} {tw}x{th} fov {thf:.1f}"
        what = f"seed {seed} job {k}: {nf} facets nch {nch} deg {degree} {kw} {tdesc}"
        try:
            got = ea.render(a, gs, nch)
        except ea.EuError as e:
            # the one legitimate refusal here: a lens polynomial without inverse over the facet
            assert single and "inverse" in str(e), what + ": " + str(e)
            for g in gs:
                g.release()
            continue
        ref = jobs.oracle_render(a, os_, nch=nch)
        assert got.shape == ref.shape, what
        same = got.view(np.uint32) == ref.view(np.uint32)
        assert same.all(), f"{what}: {int((~same).sum())} of {same.size} words differ, first at {np.argwhere(~same)[0]}"
        for g in gs:
            g.release()
