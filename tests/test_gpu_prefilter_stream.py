"""The streamed form of the device prefilter (eu_setup.hip: LDS-DMA tiles, one wavefront per
group of lines) against the oracle and against the one-thread-per-line kernels it replaces
(EU_HIP_IIR_STREAM=0). Sizes are chosen so that every remainder path runs: lines longer than a
whole number of 64-sample blocks, rows beyond the last group of 8, columns beyond the last
group of 32 floats, several poles (degree >= 4), every boundary condition the set-up uses."""
import os

import numpy as np
import pytest

import envutil_amd as ea
import euo
import jobs

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same(a, b, what):
    d = bits(a) != bits(b)
    assert not d.any(), f"{what}: {int(d.sum())} of {d.size} coefficients differ, first at {np.argwhere(d)[0]}"


class generic_kernels:
    """EU_HIP_IIR_STREAM: 0 one thread per line; 3 streamed, four passes per axis; 7 (default) streamed
    with checkpoints, the causal result recomputed in the backward sweep"""

    def __init__(self, mode="0"):
        self.mode = mode

    def __enter__(self):
        self.keep = os.environ.get("EU_HIP_IIR_STREAM")
        os.environ["EU_HIP_IIR_STREAM"] = self.mode

    def __exit__(self, *a):
        if self.keep is None:
            del os.environ["EU_HIP_IIR_STREAM"]
        else:
            os.environ["EU_HIP_IIR_STREAM"] = self.keep


@pytest.mark.parametrize("nch", [1, 2, 3, 4])
@pytest.mark.parametrize("sw,sh,degree", [(640, 320, 3), (1000, 500, 3), (1000, 500, 5), (330, 166, 2),
                                          (128, 64, 7), (772, 386, 4)])
def test_spherical_streamed_prefilter(sw, sh, degree, nch):
    """full sphere: PERIODIC rows, stacked PERIODIC columns (environment.h:356-522)"""
    img = jobs.synth_image(sw, sh, nch, seed=21 + nch)
    fct = ea.facet_spec(ea.SPHERICAL, sw, sh, 360.0, nchannels=nch)
    g = ea.Source.load(fct, img, degree)
    got = g.download()
    g.release()
    o = jobs.OracleSource(euo.SPHERICAL, sw, sh, 360.0, img, degree)
    same(got, o.container, f"spherical {sw}x{sh}x{nch} degree {degree} vs oracle")
    for mode in ("0", "3"):
        with generic_kernels(mode):
            g2 = ea.Source.load(fct, img, degree)
            same(got, g2.download(), f"streamed vs EU_HIP_IIR_STREAM={mode}")
            g2.release()


@pytest.mark.parametrize("nch", [1, 3, 4])
@pytest.mark.parametrize("sprj,sw,sh,shfov,degree", [(euo.RECTILINEAR, 333, 217, 80.0, 3),
                                                    (euo.CYLINDRICAL, 512, 130, 360.0, 3),
                                                    (euo.SPHERICAL, 200, 100, 100.0, 5),
                                                    (euo.FISHEYE, 450, 450, 180.0, 2),
                                                    (euo.STEREOGRAPHIC, 97, 401, 120.0, 4)])
def test_ordinary_streamed_prefilter(sprj, sw, sh, shfov, degree, nch):
    """bspline::prefilter along rows, then columns (REFLECT / PERIODIC)"""
    img = jobs.synth_image(sw, sh, nch, seed=5 + nch)
    g = ea.Source.load(ea.facet_spec(sprj, sw, sh, shfov, nchannels=nch), img, degree)
    o = jobs.OracleSource(sprj, sw, sh, shfov, img, degree)
    same(g.download(), o.container, f"projection {sprj} {sw}x{sh}x{nch} degree {degree}")
    g.release()


@pytest.mark.parametrize("face,degree,nch", [(100, 3, 3), (128, 2, 4), (201, 3, 3), (64, 5, 1)])
def test_cubemap_ir_streamed_prefilter(face, degree, nch):
    """NATURAL x NATURAL per section of the IR image (cubemap.h:921-946)"""
    faces = jobs.synth_cubefaces(face, nch)
    g = ea.Source.load(ea.facet_spec(ea.CUBEMAP, face, 6 * face, 90.0, nchannels=nch), faces, degree)
    o = jobs.OracleSource(euo.CUBEMAP, face, 6 * face, 90.0, faces, degree)
    same(g.download(), o.container, f"cubemap IR face {face} degree {degree}")
    g.release()


def test_large_source_streamed_equals_generic():
    """8192 x 4096 x 3 (config 2's source): both forms of the kernels, whole container"""
    rng = np.random.default_rng(3)
    img = rng.random((4096, 8192, 3), dtype=np.float32)
    fct = ea.facet_spec(ea.SPHERICAL, 8192, 4096, 360.0)
    g = ea.Source.load(fct, img, 3)
    a = g.download()
    g.release()
    with generic_kernels():
        g2 = ea.Source.load(fct, img, 3)
        b = g2.download()
        g2.release()
    same(a, b, "8192x4096 streamed vs one thread per line")


@pytest.mark.parametrize("sw,sh", [(2, 40), (3, 50), (4, 33), (40, 2), (50, 3), (3, 3), (2, 2), (5, 4), (1, 30), (30, 1)])
@pytest.mark.parametrize("degree", [2, 3, 5, 7])
def test_sources_narrower_than_the_frame(sw, sh, degree):
    """a core of 2-4 pixels on an axis: zimt braces slice by slice outward and reads slices it filled a
    step before (brace.h:134-330); the device does the same for such cores (brace_seq_kernel)"""
    for sprj, hfov in ((euo.RECTILINEAR, 60.0), (euo.CYLINDRICAL, 360.0), (euo.SPHERICAL, 90.0)):
        for nch in (1, 3):
            img = jobs.synth_image(sw, sh, nch, seed=sw * 31 + sh)
            g = ea.Source.load(ea.facet_spec(sprj, sw, sh, hfov, nchannels=nch), img, degree)
            o = jobs.OracleSource(sprj, sw, sh, hfov, img, degree)
            same(g.download(), o.container, f"narrow {sw}x{sh}x{nch} projection {sprj} degree {degree}")
            g.release()
