"""Randomised differential test: jobs drawn from the whole parameter space
(source and target projections, fields of view, orientations, spline degrees,
channel counts, twining, crop windows, band tilings, tethered output) rendered
by the HIP library and by the oracle; every float / word must be identical."""
import numpy as np
import pytest

import envutil_amd as ea
import euo
import jobs

pytestmark = pytest.mark.gpu

SRC_PRJ = [euo.SPHERICAL, euo.CYLINDRICAL, euo.RECTILINEAR, euo.STEREOGRAPHIC, euo.FISHEYE,
           euo.CUBEMAP, euo.BIATAN6]
TRG_PRJ = [ea.SPHERICAL, ea.CYLINDRICAL, ea.RECTILINEAR, ea.STEREOGRAPHIC, ea.FISHEYE, ea.CUBEMAP,
           ea.BIATAN6]


import os
TINY = os.environ.get("EU_FUZZ_TINY") == "1"     # sources of a few pixels (tests/fuzz_wide.py)


def draw_job(rng):
    sprj = SRC_PRJ[rng.integers(len(SRC_PRJ))]
    nch = int(rng.integers(1, 5))
    degree = int(rng.choice([0, 1, 1, 2, 3, 3, 4, 5]))
    if sprj in (euo.CUBEMAP, euo.BIATAN6):
        face = int(rng.integers(2, 9)) if TINY else int(rng.integers(16, 48))
        sw, sh, shfov = face, 6 * face, 90.0
    else:
        sw, sh = int(rng.integers(24, 160)), int(rng.integers(24, 120))
        if TINY:
            sw, sh = int(rng.integers(1, 13)), int(rng.integers(1, 13))
        full = sprj in (euo.SPHERICAL, euo.CYLINDRICAL) and rng.random() < 0.5
        if sprj == euo.SPHERICAL and full:
            sh = max(1 if TINY else 12, sw // 2)
            sw = 2 * sh
        shfov = 360.0 if full else float(rng.uniform(40.0, {euo.RECTILINEAR: 130.0, euo.STEREOGRAPHIC: 250.0,
                                                            euo.FISHEYE: 300.0}.get(sprj, 300.0)))
    tprj = TRG_PRJ[rng.integers(len(TRG_PRJ))]
    if tprj in (ea.CUBEMAP, ea.BIATAN6):
        tw = int(rng.integers(8, 40))
        th, thfov = 6 * tw, 90.0
    else:
        tw, th = int(rng.integers(1, 200)), int(rng.integers(1, 90))
        if rng.random() < 0.15:                      # rows longer than one and two 512-pixel segments
            tw, th = int(rng.integers(500, 1300)), int(rng.integers(1, 12))
        thfov = float(rng.uniform(30.0, {ea.RECTILINEAR: 140.0, ea.STEREOGRAPHIC: 300.0}.get(tprj, 360.0)))
    kw = dict(yaw=float(rng.uniform(-180, 180)), pitch=float(rng.uniform(-90, 90)),
              roll=float(rng.uniform(-180, 180)), spline_degree=degree,
              twine=int(rng.choice([0, 0, 2, 3])))
    if rng.random() < 0.25 and tw >= 4 and th >= 4:
        x0, y0 = int(rng.integers(0, tw // 2)), int(rng.integers(0, th // 2))
        kw["crop"] = (x0, int(rng.integers(x0 + 1, tw + 1)), y0, int(rng.integers(y0 + 1, th + 1)))
    if rng.random() < 0.2:
        kw["tethered"] = True
    src_kw = dict(yaw=float(rng.uniform(-180, 180)), pitch=float(rng.uniform(-90, 90)),
                  roll=float(rng.uniform(-30, 30)), brighten=float(rng.choice([1.0, 1.0, 0.8, 1.3])))
    if sprj in (euo.RECTILINEAR, euo.FISHEYE, euo.STEREOGRAPHIC, euo.CYLINDRICAL) and rng.random() < 0.3:
        src_kw["lens"] = dict(a=float(rng.uniform(-0.02, 0.02)), b=float(rng.uniform(-0.04, 0.04)),
                              c=float(rng.uniform(-0.03, 0.03)), h=float(rng.choice([0.0, 0.01])),
                              v=float(rng.choice([0.0, -0.02])))
    out_n = nch if rng.random() < 0.8 else int(rng.integers(1, 5))
    band = None
    if rng.random() < 0.25:
        cnt = int(rng.integers(2, 5))
        band = (int(rng.choice([4, 8, 16])), cnt, int(rng.integers(0, cnt)))
    return sprj, sw, sh, shfov, nch, degree, tprj, tw, th, thfov, kw, src_kw, out_n, band


@pytest.mark.parametrize("seed", range(8))
def test_random_jobs_bit_identical(seed):
    rng = np.random.default_rng(1000 + seed)
    for k in range(6):
        sprj, sw, sh, shfov, nch, degree, tprj, tw, th, thfov, kw, src_kw, out_n, band = draw_job(rng)
        img = jobs.synth_image(sw, sh, nch, seed=seed * 100 + k)
        o = jobs.OracleSource(sprj, sw, sh, shfov, img, degree, **src_kw)
        g = ea.Source.adopt(ea.facet_spec(sprj, sw, sh, shfov, nchannels=nch, yaw=src_kw["yaw"],
                                          pitch=src_kw["pitch"], roll=src_kw["roll"],
                                          brighten=src_kw["brighten"], lens=src_kw.get("lens")),
                            o.container, degree, o.bc[0], o.bc[1])
        a = ea.arguments(tprj, tw, th, thfov, **kw)
        what = f"seed {seed} job {k}: src {sprj} {sw}x{sh} fov {shfov:.1f} nch {nch} deg {degree} -> " \
               f"trg {tprj} {tw}x{th} fov {thfov:.1f} {kw} out {out_n} band {band} lens {src_kw.get('lens')}"
        ref = jobs.oracle_render(a, o, nch=out_n)
        if band is None:
            got = ea.render(a, g, out_n)
        else:
            rows = ea.band_frame_rows(a.out_height, *band)
            got = ea.render(a, g, out_n, band=band)
            ref = ref[rows]
        assert got.shape == ref.shape, what
        same = got.view(np.uint32) == ref.view(np.uint32)
        assert same.all(), f"{what}: {int((~same).sum())} of {same.size} words differ"
        g.release()


@pytest.mark.parametrize("seed", range(4))
def test_random_multi_facet_jobs_bit_identical(seed):
    """2-9 facets of mixed projections, channel counts, orientations, lens
    parameters; voronoi_syn / voronoi_syn_plus by the job's channel count"""
    rng = np.random.default_rng(5000 + seed)
    for k in range(2):
        nf = int(rng.integers(2, 10))
        degree = int(rng.choice([0, 1, 1, 2, 3, 4]))
        out_n = int(rng.integers(1, 5))
        os_, gs = [], []
        for f in range(nf):
            sprj, sw, sh, shfov, nch, _, _, _, _, _, _, src_kw, _, _ = draw_job(rng)
            if rng.random() < 0.7:
                nch = out_n                         # mostly uniform channel counts, sometimes mixed
            img = jobs.synth_image(sw, sh, nch, seed=seed * 1000 + k * 50 + f)
            if nch in (2, 4):
                yy, xx = np.mgrid[0:sh, 0:sw]
                r = np.hypot((xx - sw / 2) / (sw / 2), (yy - sh / 2) / (sh / 2))
                img[:, :, nch - 1] = np.clip(1.5 - 1.3 * r, 0.0, 1.0)
            o = jobs.OracleSource(sprj, sw, sh, shfov, img, degree, **src_kw)
            gs.append(ea.Source.adopt(ea.facet_spec(sprj, sw, sh, shfov, nchannels=nch, yaw=src_kw["yaw"],
                                                    pitch=src_kw["pitch"], roll=src_kw["roll"],
                                                    brighten=src_kw["brighten"], lens=src_kw.get("lens")),
                                      o.container, degree, o.bc[0], o.bc[1]))
            os_.append(o)
        tprj = TRG_PRJ[rng.integers(len(TRG_PRJ))]
        if tprj in (ea.CUBEMAP, ea.BIATAN6):
            tw = int(rng.integers(8, 32))
            th, thfov = 6 * tw, 90.0
        else:
            tw, th = int(rng.integers(1, 150)), int(rng.integers(1, 70))
            thfov = float(rng.uniform(30.0, {ea.RECTILINEAR: 140.0, ea.STEREOGRAPHIC: 300.0}.get(tprj, 360.0)))
        kw = {}
        if rng.random() < 0.3 and tw >= 4 and th >= 4:
            x0, y0 = int(rng.integers(0, tw // 2)), int(rng.integers(0, th // 2))
            kw["crop"] = (x0, int(rng.integers(x0 + 1, tw + 1)), y0, int(rng.integers(y0 + 1, th + 1)))
        if rng.random() < 0.25:
            kw["tethered"] = True
        a = ea.arguments(tprj, tw, th, thfov, yaw=float(rng.uniform(-180, 180)),
                         pitch=float(rng.uniform(-90, 90)), roll=float(rng.uniform(-180, 180)),
                         spline_degree=degree, twine=int(rng.choice([0, 0, 2])), **kw)
        ref = jobs.oracle_render(a, os_, nch=out_n)
        if rng.random() < 0.3:
            cnt = int(rng.integers(2, 5))
            band = (int(rng.choice([4, 8, 16])), cnt, int(rng.integers(0, cnt)))
            got = ea.render(a, gs, out_n, band=band)
            ref = ref[ea.band_frame_rows(a.out_height, *band)]
        else:
            got = ea.render(a, gs, out_n)
        assert got.shape == ref.shape
        same = got.view(np.uint32) == ref.view(np.uint32)
        assert same.all(), f"seed {seed} job {k}: {nf} facets deg {degree} out {out_n} trg {tprj} {tw}x{th}: " \
                           f"{int((~same).sum())} of {same.size} words differ"
        for g in gs:
            g.release()


@pytest.mark.parametrize("seed", range(4))
def test_random_device_setup_bit_identical(seed):
    """source set-up ON THE DEVICE (container, bracing, prefilter with blocked
    loads, spherical two-axis scheme, cubemap IR) against the oracle's, for
    random sizes / degrees / channel counts: the coefficient arrays must be
    bit-identical"""
    rng = np.random.default_rng(9000 + seed)
    for k in range(5):
        sprj, sw, sh, shfov, nch, degree, *_ = draw_job(rng)
        pdeg = int(rng.choice([degree, degree, 0, 1, 3, 5]))
        img = jobs.synth_image(sw, sh, nch, seed=seed * 77 + k)
        # cubemap IR geometry: --support_min / --tile_size (cubemap.h:233-400)
        smin, tile = int(rng.choice([8, 8, 4, 12, 1])), int(rng.choice([64, 64, 16, 32]))
        o = jobs.OracleSource(sprj, sw, sh, shfov, img, degree, pdeg, support_min=smin, tile=tile)
        try:
            g = ea.Source.load(ea.facet_spec(sprj, sw, sh, shfov, nchannels=nch), img, degree, pdeg,
                               support_min=smin, tile_size=tile)
        except ea.EuError as e:
            # a cubemap whose support frame is narrower than the spline reaches is refused (the reference reads
            # outside its IR array there: nothing to compare)
            assert "support frame" in str(e) and sprj in (euo.CUBEMAP, euo.BIATAN6) and smin < degree // 2 + 1, str(e)
            continue
        got = g.download().reshape(-1)
        ref = np.ascontiguousarray(o.container, np.float32).reshape(-1)
        assert got.shape == ref.shape, (sprj, sw, sh, nch, degree, pdeg)
        same = got.view(np.uint32) == ref.view(np.uint32)
        assert same.all(), f"seed {seed} job {k}: prj {sprj} {sw}x{sh} fov {shfov:.1f} nch {nch} degree {degree} " \
                           f"prefilter {pdeg} support {smin} tile {tile}: {int((~same).sum())} of {same.size} " \
                           "coefficients differ"
        # and a render from the device-built source (pick-up geometry of that IR)
        a = ea.arguments(ea.SPHERICAL, 96, 48, 360.0, yaw=float(rng.uniform(-180, 180)),
                         pitch=float(rng.uniform(-60, 60)), spline_degree=degree)
        got, ref = ea.render(a, g), jobs.oracle_render(a, o)
        assert (got.view(np.uint32) == ref.view(np.uint32)).all(), f"seed {seed} job {k}: render from device set-up"
        g.release()
