"""The LDS-staging render kernels (envutil_amd/csrc/eu_render4.hip) against the CPU oracle,
bit for bit. The library uses them by default only where they measured faster (cubic jobs on
cubemap sources); EU_HIP_R4=1 (read on every call) sends every job they cover through them, so
that the staged kernel, its column-plan variant (upright cubemap / rectilinear targets of a
lat/lon source), the work list and the direct-gather kernel behind it all see the lat/lon cases
too: tiles on the +-180 degree seam and at the poles, ragged frame edges, row ranges, bands."""
import os

import numpy as np
import pytest

import envutil_amd as ea
import euo
import jobs
from test_gpu_parity import assert_bits, make_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def staged_everywhere():
    old = os.environ.get("EU_HIP_R4")
    os.environ["EU_HIP_R4"] = "1"
    yield
    if old is None:
        os.environ.pop("EU_HIP_R4", None)
    else:
        os.environ["EU_HIP_R4"] = old


@pytest.fixture(scope="module")
def latlon():
    return {n: jobs.synth_image(512, 256, n) for n in (3, 4)}


TARGETS = [
    (ea.CUBEMAP, 96, 576, 90.0),          # equatorial faces: column plans; polar faces: work list
    (ea.CUBEMAP, 41, 246, 90.0),          # ragged tiles on both axes
    (ea.RECTILINEAR, 200, 120, 100.0),    # one column plan for the whole frame
    (ea.SPHERICAL, 300, 150, 360.0),      # BCA form: no column plan, seam and poles
    (ea.BIATAN6, 32, 192, 90.0),
]


@pytest.mark.parametrize("tprj,tw,th,thfov", TARGETS)
@pytest.mark.parametrize("degree", [1, 2, 3])
@pytest.mark.parametrize("nch", [3, 4])
def test_latlon_source(latlon, tprj, tw, th, thfov, degree, nch):
    o, g = make_pair(euo.SPHERICAL, 512, 256, 360.0, latlon[nch], degree)
    for ypr in ((0, 0, 0), (0, 0, 0, "yaw"), (25, -10, 5)):
        yaw, pitch, roll = (40, 0, 0) if len(ypr) == 4 else ypr
        a = ea.arguments(tprj, tw, th, thfov, yaw=yaw, pitch=pitch, roll=roll, spline_degree=degree)
        assert_bits(ea.render(a, g, nch), jobs.oracle_render(a, o), f"pixels ypr {yaw, pitch, roll}")


@pytest.mark.parametrize("degree", [2, 3])
@pytest.mark.parametrize("sprj", [euo.CUBEMAP, euo.BIATAN6])
def test_cubemap_source(degree, sprj):
    faces = jobs.synth_cubefaces(96, 3)
    o, g = make_pair(sprj, 96, 576, 90.0, faces, degree)
    for a in (ea.arguments(ea.SPHERICAL, 400, 200, 360.0, spline_degree=degree),
              ea.arguments(ea.RECTILINEAR, 150, 110, 80.0, yaw=33, pitch=21, roll=-8, spline_degree=degree)):
        assert_bits(ea.render(a, g), jobs.oracle_render(a, o), "pixels")


def test_partial_source_misses_are_zero(latlon):
    """a 120 degree lat/lon window: tiles with misses, tiles without any hit"""
    img = jobs.synth_image(256, 128, 3)
    o, g = make_pair(euo.SPHERICAL, 256, 128, 120.0, img, 3)
    a = ea.arguments(ea.CUBEMAP, 64, 384, 90.0, spline_degree=3)
    got, ref = ea.render(a, g), jobs.oracle_render(a, o)
    assert_bits(got, ref, "pixels")
    assert (ref == 0).any() and (ref != 0).any()


def test_row_ranges_and_bands_assemble(latlon):
    o, g = make_pair(euo.SPHERICAL, 512, 256, 360.0, latlon[3], 3)
    a = ea.arguments(ea.CUBEMAP, 64, 384, 90.0, spline_degree=3)
    ref = jobs.oracle_render(a, o)
    parts = [ea.render(a, g, 3, r0, r1) for r0, r1 in ((0, 101), (101, 102), (102, 384))]
    assert_bits(np.concatenate(parts, 0), ref, "row ranges")
    frame = np.zeros_like(ref)
    for k in range(3):
        band = (8, 3, k)
        rows = ea.band_frame_rows(384, *band)
        frame[rows] = ea.render(a, g, 3, 0, len(rows), band=band)
    assert_bits(frame, ref, "bands")


def test_switch_off_gives_the_same_frame(latlon):
    o, g = make_pair(euo.SPHERICAL, 512, 256, 360.0, latlon[3], 3)
    a = ea.arguments(ea.CUBEMAP, 96, 576, 90.0, spline_degree=3)
    on = ea.render(a, g)
    os.environ["EU_HIP_R4"] = "0"
    off = ea.render(a, g)
    assert_bits(on, off, "EU_HIP_R4=1 vs 0")
