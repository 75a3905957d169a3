"""Round 3's run-time switches, each against the CPU oracle bit for bit and against the other setting:
  EU_HIP_COLMAJOR  the walk of the XCD units (column by column for rotated / twined jobs, row by row for upright
                   ones; eu_xcd_tile, eu_render_dev.h) - forced both ways on jobs of both kinds, incl. a frame whose
                   last unit is partial,
  EU_HIP_REJ       the multi-facet kernels' second early-miss stage (eu_multi_maybe: a conservative table test in
                   front of the exact hit test; off by default) - fisheye facets with and without the lens
                   polynomial and with a shift, voronoi_syn / voronoi_syn_plus / hdr_merge, twining.
Both are read on every launch."""
import os

import numpy as np
import pytest

import envutil_amd as ea
import euo
import jobs
from test_gpu_parity import assert_bits, make_pair, facet_set

pytestmark = pytest.mark.gpu


class env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("degree", [1, 3])
@pytest.mark.parametrize("twine", [0, 2])
def test_column_walk_matches_row_walk_and_oracle(degree, twine):
    img = jobs.synth_image(512, 256, 3)
    o, g = make_pair(euo.SPHERICAL, 512, 256, 360.0, img, degree)
    # 700 x 333: 6 tile columns of 128 (the last ragged), 84 tile rows of 4 = 10 units of 8 and a partial one
    for ypr in ((0, 0, 0), (25, -10, 5)):
        a = ea.arguments(ea.SPHERICAL, 700, 333, 360.0, yaw=ypr[0], pitch=ypr[1], roll=ypr[2],
                         spline_degree=degree, twine=twine)
        ref = jobs.oracle_render(a, o)
        outs = []
        for cm in ("0", "1", None):
            with env(EU_HIP_COLMAJOR=cm, EU_HIP_R4="0"):
                outs.append(ea.render(a, g, 3))
            assert_bits(outs[-1], ref, f"walk {cm} ypr {ypr} twine {twine}")


@pytest.mark.parametrize("nch", [3, 4])
@pytest.mark.parametrize("lens", [None, dict(a=0.01, b=-0.03, c=0.02), dict(a=0.0, b=0.02, c=-0.01, h=0.07, v=-0.04)])
def test_early_miss_tables_change_nothing(nch, lens):
    os_, gs = facet_set(euo.FISHEYE, 96, 96, 130.0, nch, 1, lens)
    for twine in (0, 2):
        a = ea.arguments(ea.SPHERICAL, 300, 150, 360.0, yaw=10, pitch=4, roll=-2, spline_degree=1, twine=twine)
        ref = jobs.oracle_render(a, os_)
        with env(EU_HIP_REJ="1"):
            got = ea.render(a, gs, nch)
        assert_bits(got, ref, f"early-miss tables on, nch {nch} lens {lens} twine {twine}")
        with env(EU_HIP_REJ="2"):
            assert_bits(ea.render(a, gs, nch), ref, f"table-free early miss, nch {nch} lens {lens} twine {twine}")
        with env(EU_HIP_REJ=None):
            assert_bits(ea.render(a, gs, nch), got, "tables off against tables on")


def test_early_miss_tables_hdr_merge_and_narrow_facets():
    # narrow facets (most rays miss every facet: the tables decide for most of the frame) and hdr_merge,
    # where a miss still takes part in the sum
    os_, gs = facet_set(euo.FISHEYE, 64, 64, 60.0, 3, 1)
    for syn in (None, "hdr_merge"):
        kw = dict(synopsis=syn) if syn else {}
        a = ea.arguments(ea.SPHERICAL, 256, 128, 360.0, spline_degree=1, **kw)
        ref = jobs.oracle_render(a, os_)
        for mode in ("1", "2"):
            with env(EU_HIP_REJ=mode):
                assert_bits(ea.render(a, gs, 3), ref, f"early miss {mode}, synopsis {syn}")
    assert (ref == 0).mean() > 0.3


def test_staged_jobs_on_alternating_streams():
    """The staged kernels' work list and queues are the library's, one launch pair at a time: jobs issued back to
    back on two different streams (no synchronisation between them) must come out as they do one by one."""
    import ctypes as C
    torch = pytest.importorskip("torch")
    img = jobs.synth_image(512, 256, 3)
    o, g = make_pair(euo.SPHERICAL, 512, 256, 360.0, img, 3)
    a1 = ea.arguments(ea.CUBEMAP, 96, 576, 90.0, spline_degree=3)
    a2 = ea.arguments(ea.CUBEMAP, 80, 480, 90.0, spline_degree=3)
    dev = torch.device("cuda", 0)
    with env(EU_HIP_R4="1"):
        want = [ea.render(a, g, 3) for a in (a1, a2)]
        outs = [torch.zeros((a.height, a.width, 3), device=dev, dtype=torch.float32) for a in (a1, a2)]
        streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        tg = [a.target(3, 0, a.height, 0, None) for a in (a1, a2)]
        srcs = (C.c_void_p * 1)(g.handle)
        for rep in range(6):
            k = rep & 1
            rc = ea.lib().eu_hip_render(C.byref(tg[k]), srcs, 1, C.c_void_p(outs[k].data_ptr()), tg[k].width * 3 * 4,
                                        1, C.c_void_p(streams[k].cuda_stream))
            assert rc == 0, ea.lib().eu_hip_last_error()
        ea.lib().eu_hip_sync()
        torch.cuda.synchronize()
    for k in (0, 1):
        assert_bits(outs[k].cpu().numpy(), want[k], f"job {k} on its own stream")


@pytest.mark.parametrize("degree", [1, 2, 3])
@pytest.mark.parametrize("w,h", [(2, 1), (4, 2), (6, 3), (8, 4), (10, 5)])
def test_full_sphere_sources_smaller_than_their_frame(w, h, degree):
    """A 360 x 180 degree image lower / narrower than the spline's frame (refused until round 3): the over-the-pole
    rows in the reference's alternating order (environment.h:455-516, sources may be frame rows written a round
    earlier), the horizontal bracing slice by slice - the device-built container and a render against the oracle."""
    img = jobs.synth_image(w, h, 3, seed=w)
    src = ea.Source.load(ea.facet_spec(ea.SPHERICAL, w, h, 360.0), img, degree)
    osrc = jobs.OracleSource(euo.SPHERICAL, w, h, 360.0, img, degree)
    assert_bits(src.download(), osrc.container, f"container {w}x{h} degree {degree}")
    for a in (ea.arguments(ea.CUBEMAP, 16, 96, 90.0, spline_degree=degree),
              ea.arguments(ea.SPHERICAL, 40, 20, 360.0, yaw=20, pitch=30, roll=-10, spline_degree=degree)):
        assert_bits(ea.render(a, src, 3), jobs.oracle_render(a, osrc), f"render from {w}x{h} degree {degree}")


def test_cubemap_support_narrower_than_the_spline_is_refused():
    """--support_min 1 --tile_size 16 on 45-pixel faces leaves a frame of 1 / 2 pixels; a degree-4 spline reaches 3
    texels beyond a pick-up at the face's edge: the reference reads outside its IR array there, the library says
    so (found by tests/fuzz_wide.py at seed 1033). With the default support of 8 the same source loads."""
    faces = jobs.synth_cubefaces(45, 3)
    with pytest.raises(ea.EuError, match="support frame"):
        ea.Source.load(ea.facet_spec(ea.BIATAN6, 45, 270, 90.0), faces, 4, support_min=1, tile_size=16)
    g = ea.Source.load(ea.facet_spec(ea.BIATAN6, 45, 270, 90.0), faces, 4)
    o = jobs.OracleSource(euo.BIATAN6, 45, 270, 90.0, faces, 4)
    a = ea.arguments(ea.SPHERICAL, 96, 48, 360.0, yaw=158.5, pitch=45.2, spline_degree=4)
    assert_bits(ea.render(a, g, 3), jobs.oracle_render(a, o), "degree 4 from a biatan6 source with the default support")
