"""Parity at BASELINE.json's full sizes for configs 1 - 5 (the headline is in
test_gpu_fullsize.py): the sources are built on the GPU as bench.py builds them, the oracle
renders bands of rows of the real frame from the coefficients the GPU built (downloaded), and
the GPU's rows must be the same bits. Config 4 (9 coordinate chains per pixel on a 6.4 GB
source) is checked on a few rows only."""
import math
import os
import sys

import numpy as np
import pytest

import envutil_amd as ea
import euo
import jobs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VIEWS5 = [(0, 0, 0), (90, 0, 0), (180, 0, 0), (270, 0, 0), (0, 90, 0), (0, -90, 0)]
LENS5 = dict(a=0.01, b=-0.03, c=0.02)


def build(name):
    """sources on the GPU + their oracle twins over the downloaded coefficients + the job"""
    import torch
    import bench
    from envutil_amd.api import PROJECTION_NAMES
    (sname, sw, sh, shfov), (tname, tw, th, thfov), nch, degree, twine, ypr = bench.WORKLOADS[name]
    sprj, tprj = PROJECTION_NAMES.index(sname), PROJECTION_NAMES.index(tname)
    dev = torch.device("cuda:0")
    views = VIEWS5 if name == "config5" else [(0, 0, 0)]
    gs, os_ = [], []
    for v in views:
        kw = dict(yaw=v[0], pitch=v[1], roll=v[2])
        if name == "config5":
            kw["lens"] = LENS5
        img = bench.synth_on_device(torch, dev, sw, sh, nch)
        if nch in (2, 4):
            img[:, :, nch - 1] = 1.0
        host = img.cpu().numpy()
        del img
        torch.cuda.empty_cache()
        g = ea.Source.load(ea.facet_spec(sprj, sw, sh, shfov, nchannels=nch, **kw), host, degree)
        del host
        geom, _ = g.info()
        metrics = ea.cubemap_metrics(sw) if sprj in (ea.CUBEMAP, ea.BIATAN6) else None
        os_.append(jobs.oracle_source_from_container(sprj, sw, sh, shfov, g.download(), geom, degree, nch,
                                                     metrics, **kw))
        gs.append(g)
    args = ea.arguments(tprj, tw, th, thfov, yaw=ypr[0], pitch=ypr[1], roll=ypr[2],
                        spline_degree=degree, twine=twine)
    return gs, (os_ if len(os_) > 1 else os_[0]), args, nch, th


def check_rows(name, starts, rows=8):
    gs, osrc, args, nch, th = build(name)
    try:
        for r0 in starts:
            r0 = min(r0, th - rows)
            ref = jobs.oracle_render(args, osrc, 0, r0, r0 + rows, nthreads=16)
            for r4 in ("0", "1", None):
                if r4 is None:
                    os.environ.pop("EU_HIP_R4", None)
                else:
                    os.environ["EU_HIP_R4"] = r4
                got = ea.render(args, gs, nch, r0, r0 + rows)
                assert (jobs.bits(got) == jobs.bits(ref)).all(), f"{name} rows {r0}..{r0 + rows} (EU_HIP_R4={r4})"
    finally:
        os.environ.pop("EU_HIP_R4", None)
        for g in gs:
            g.release()


def test_config1_whole_frame_bit_identical():
    """2048x1024 lat/lon -> 1024x1024 rectilinear, hfov 90, bilinear (BASELINE config 1, the reference's own
    CPU-runnable case): 1 Mpix - the WHOLE frame against the oracle, under every kernel selection"""
    check_rows("config1", (0,), rows=1024)


def test_config2_rows_bit_identical():
    """8192x4096 lat/lon -> 6x2048 cubemap, bilinear: face tops, a pole, the seam, the end"""
    check_rows("config2", (0, 2048 - 4, 4096 + 1020, 10240 + 1020, 12288 - 8))


def test_config3_rows_bit_identical():
    """6x2048 cubemap -> 16384x8192 spherical, degree 3: poles, cube edges, the equator"""
    check_rows("config3", (0, 2048 - 4, 2896, 4096 - 4, 6144, 8192 - 8))


def test_config5_rows_bit_identical():
    """six RGBA fisheye facets with lens polynomial -> 16384x8192 spherical, voronoi_syn_plus"""
    check_rows("config5", (0, 2048, 4096 - 4, 7000), rows=4)


def test_config4_rows_bit_identical():
    """32768x16384 -> same, yaw / pitch / roll, 3x3 twining: two bands of two rows"""
    check_rows("config4", (5, 8191), rows=2)
