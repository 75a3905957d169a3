"""Parity at BASELINE.json's full sizes (headline: 16384x8192 lat/lon -> 6x4096
cubemap, degree 3): the oracle renders bands of rows of the real frame from the
coefficients the GPU built, and the multi-GPU tilings reassemble the frame."""
import os
import sys

import numpy as np
import pytest

import envutil_amd as ea
import euo
import jobs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def headline():
    import torch
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda:0")
    sw, sh, nch, deg = 16384, 8192, 3, 3
    img = bench.synth_on_device(torch, dev, sw, sh, nch).cpu().numpy()
    torch.cuda.empty_cache()
    src = ea.Source.load(ea.facet_spec(ea.SPHERICAL, sw, sh, 360.0), img, deg)   # device prefilter
    del img
    args = ea.arguments(ea.CUBEMAP, 4096, 24576, 90.0, spline_degree=deg)
    yield src, args
    src.release()


def test_full_size_rows_bit_identical_to_oracle(headline):
    src, args = headline
    cont = src.download()
    g, _ = src.info()
    osrc = jobs.oracle_source_from_container(euo.SPHERICAL, 16384, 8192, 360.0, cont, g, 3, 3)
    # face 0 top, face seam 1|2, the pole of face 2, the +-180 degree seam region of
    # face 5, the last rows of the frame
    for r0 in (0, 8192 - 8, 8192 + 2048 - 8, 20480 + 2040, 24576 - 16):
        ref = jobs.oracle_render(args, osrc, 0, r0, r0 + 16, nthreads=16)
        got = ea.render(args, src, 3, r0, r0 + 16)
        assert (jobs.bits(got) == jobs.bits(ref)).all(), f"rows {r0}..{r0 + 16}"


def test_full_size_tilings_reassemble(headline):
    """row strips and interleaved bands (what N GPUs render) against one launch"""
    import torch
    src, args = headline
    dev = torch.device("cuda:0")
    th, tw, nch = 24576, 4096, 3

    def on_device(rows, **kw):
        out = torch.empty((rows, tw, nch), device=dev, dtype=torch.float32)
        t = args.target(nch, kw.get("r0", 0), kw.get("r1"), 0, kw.get("band"))
        arr = (ea.api.C.c_void_p * 1)(src.handle)
        rc = ea.lib().eu_hip_render(ea.api.C.byref(t), arr, 1, ea.api.C.c_void_p(out.data_ptr()),
                                    tw * nch * 4, 1, None)
        assert rc == 0
        ea.lib().eu_hip_sync()
        return out

    whole = on_device(th)
    # 8 interleaved band parts
    frame = torch.full_like(whole, float("nan"))
    for part in range(8):
        n = ea.band_rows(th, 64, 8, part)
        rows = torch.from_numpy(ea.band_frame_rows(th, 64, 8, part)).to(dev)
        frame[rows] = on_device(n, band=(64, 8, part))
    assert torch.equal(frame.view(torch.int32), whole.view(torch.int32))
    # 3 contiguous strips with ragged boundaries
    frame.fill_(float("nan"))
    for r0, r1 in ((0, 8190), (8190, 16391), (16391, th)):
        out = torch.empty((r1 - r0, tw, nch), device=dev, dtype=torch.float32)
        t = args.target(nch, r0, r1, 0)
        arr = (ea.api.C.c_void_p * 1)(src.handle)
        assert ea.lib().eu_hip_render(ea.api.C.byref(t), arr, 1, ea.api.C.c_void_p(out.data_ptr()),
                                      tw * nch * 4, 1, None) == 0
        ea.lib().eu_hip_sync()
        frame[r0:r1] = out
    assert torch.equal(frame.view(torch.int32), whole.view(torch.int32))


def test_full_size_cost_partition_and_layout_segments(headline):
    """what bench.py does for N > 1 on this job: the library's layout segments, strips of
    equal estimated cost, every strip rendered (with the launch-level layout choice) and
    put back: the single-launch frame"""
    import torch
    from envutil_amd.distributed import cost_partition
    src, args = headline
    dev = torch.device("cuda:0")
    th, tw, nch = 24576, 4096, 3
    seg_rows, flags = ea.layout_segments(args, src, nch)
    assert seg_rows == 512 and flags.size == th // 512
    # the inner halves of the two polar faces, nothing on the equatorial faces
    assert flags[18:22].all() and flags[26:30].all() and not flags[:16].any() and not flags[32:].any()

    def on_device(r0, r1):
        out = torch.empty((r1 - r0, tw, nch), device=dev, dtype=torch.float32)
        t = args.target(nch, r0, r1, 0)
        arr = (ea.api.C.c_void_p * 1)(src.handle)
        assert ea.lib().eu_hip_render(ea.api.C.byref(t), arr, 1, ea.api.C.c_void_p(out.data_ptr()),
                                      tw * nch * 4, 1, None) == 0
        ea.lib().eu_hip_sync()
        return out

    whole = on_device(0, th)
    for world in (2, 8):
        frame = torch.full_like(whole, float("nan"))
        for r0, r1 in cost_partition(th, world, seg_rows, flags):
            frame[r0:r1] = on_device(r0, r1)
        assert torch.equal(frame.view(torch.int32), whole.view(torch.int32))
