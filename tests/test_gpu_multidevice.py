"""One process, several device slots behind the C ABI (include/eu_hip.h: eu_hip_init_devices /
eu_hip_render_devices): the frame's rows tiled over the slots, the source replicated by a peer copy,
the strips gathered into one buffer. On a one-GPU box the same device is listed twice and three times (separate
streams, stepper tables, plan caches and strip buffers per slot): every frame must be the single-launch frame
bit for bit - host output and device output, single-facet (the staged kernel's strips), bilinear, twined and
multi-facet jobs."""
import numpy as np
import pytest
import torch

import envutil_amd as ea
import jobs
from test_gpu_parity import assert_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def slots():
    ea.lib()
    ea.init_devices([0, 0, 0])
    assert ea.device_slots() == 3
    return 3


def test_single_facet_jobs(slots):
    img = jobs.synth_image(512, 256, 3)
    for degree in (1, 3):
        src = ea.Source.load(ea.facet_spec(ea.SPHERICAL, 512, 256, 360.0), img, degree)
        for a in (ea.arguments(ea.CUBEMAP, 128, 768, 90.0, spline_degree=degree),
                  ea.arguments(ea.SPHERICAL, 320, 160, 360.0, yaw=30, pitch=15, roll=7.5, spline_degree=degree, twine=2),
                  ea.arguments(ea.RECTILINEAR, 200, 131, 100.0, spline_degree=degree)):
            want = ea.render(a, src, 3)
            strips = ea.device_strips(a, src, 3)
            assert strips[0][0] == 0 and strips[-1][1] == want.shape[0]
            assert all(s[1] == t[0] for s, t in zip(strips, strips[1:]))
            assert_bits(ea.render_devices(a, src, 3), want, f"host output, degree {degree}")
            out = torch.zeros(want.shape, device="cuda:0", dtype=torch.float32)
            ea.render_devices(a, src, 3, out_dev_ptr=out.data_ptr())
            torch.cuda.synchronize()
            assert_bits(out.cpu().numpy(), want, f"device output, degree {degree}")


def test_multi_facet_job(slots):
    rng = np.random.default_rng(5)
    srcs = []
    for k, (yaw, pitch) in enumerate([(0, 0), (70, 10), (-80, -15)]):
        img = rng.random((96, 128, 4), dtype=np.float32)
        img[..., 3] = 1.0
        srcs.append(ea.Source.load(ea.facet_spec(ea.RECTILINEAR, 128, 96, 80.0, nchannels=4, yaw=yaw, pitch=pitch), img, 1))
    a = ea.arguments(ea.SPHERICAL, 256, 128, 360.0, spline_degree=1)
    assert_bits(ea.render_devices(a, srcs, 4), ea.render(a, srcs, 4), "three facets over three slots")


def test_after_facet_update(slots):
    """a replica follows the original's new orientation (eu_hip_source_update_facet)"""
    img = jobs.synth_image(256, 128, 3)
    src = ea.Source.load(ea.facet_spec(ea.SPHERICAL, 256, 128, 360.0), img, 1)
    a = ea.arguments(ea.RECTILINEAR, 160, 120, 90.0, spline_degree=1)
    assert_bits(ea.render_devices(a, src, 3), ea.render(a, src, 3), "before")
    src.update_facet(ea.facet_spec(ea.SPHERICAL, 256, 128, 360.0, yaw=33.0, pitch=-12.0))
    assert_bits(ea.render_devices(a, src, 3), ea.render(a, src, 3), "after the update")
