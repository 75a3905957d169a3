"""On the device: the range-restricted FMA division / square root and the packed
atan2f of the two-pixel kernel against hipcc's correctly rounded `/`, sqrtf and
the scalar glibc restatement; and the packed kernel against the general one."""
import ctypes as C

import numpy as np
import pytest

import envutil_amd as ea
import euo
import jobs

pytestmark = pytest.mark.gpu


def test_device_math_selftest():
    L = ea.lib()
    L.eu_hip_selftest_math.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_void_p]
    bad = (C.c_uint64 * 4)()
    # 2048 blocks x 256 threads x 2048 iterations x 2 lanes = 2.1e9 samples per function
    assert L.eu_hip_selftest_math(20251226, 2048, 2048, bad) == 0, L.eu_hip_last_error()
    names = ["div2_safe vs /", "sqrt2_safe vs sqrtf", "atan2f_2 vs scalar", "const div vs /"]
    assert list(bad) == [0, 0, 0, 0], dict(zip(names, list(bad)))


@pytest.mark.parametrize("degree", [1, 2, 3])
@pytest.mark.parametrize("nch", [1, 3, 4])
def test_packed_kernel_bit_exact(degree, nch):
    """the two-pixel kernel covers lat/lon sources without twining: every
    target form, odd widths, rotation"""
    img = jobs.synth_image(512, 256, nch)
    o = jobs.OracleSource(euo.SPHERICAL, 512, 256, 360.0, img, degree)
    g = ea.Source.adopt(ea.facet_spec(ea.SPHERICAL, 512, 256, 360.0, nchannels=nch), o.container,
                        degree, o.bc[0], o.bc[1])
    for tprj, tw, th, thfov, ypr in [(ea.CUBEMAP, 200, 1200, 90.0, (0, 0, 0)),
                                     (ea.CUBEMAP, 65, 390, 90.0, (10, 20, 30)),
                                     (ea.SPHERICAL, 777, 123, 360.0, (170, 88, 3)),
                                     (ea.RECTILINEAR, 130, 131, 100.0, (0, -90, 0)),
                                     (ea.BIATAN6, 33, 198, 90.0, (0, 0, 0))]:
        a = ea.arguments(tprj, tw, th, thfov, yaw=ypr[0], pitch=ypr[1], roll=ypr[2],
                         spline_degree=degree)
        got, ref = ea.render(a, g), jobs.oracle_render(a, o)
        bad = int((jobs.bits(got) != jobs.bits(ref)).sum())
        assert bad == 0, (tprj, tw, bad, jobs.ulp_diff(got, ref).max())


def test_packed_kernel_partial_sphere_misses():
    img = jobs.synth_image(300, 100, 3, seed=3)
    o = jobs.OracleSource(euo.SPHERICAL, 300, 100, 200.0, img, 3, yaw=20, pitch=-10, roll=5, brighten=1.5)
    g = ea.Source.adopt(ea.facet_spec(ea.SPHERICAL, 300, 100, 200.0, yaw=20, pitch=-10, roll=5, brighten=1.5),
                        o.container, 3, o.bc[0], o.bc[1])
    a = ea.arguments(ea.SPHERICAL, 400, 200, 360.0, spline_degree=3)
    got, ref = ea.render(a, g), jobs.oracle_render(a, o)
    assert (jobs.bits(got) == jobs.bits(ref)).all()
    assert (ref == 0).all(axis=2).any() and (ref != 0).any()


def test_device_pointer_as_torch_tensor():
    """bench.py's multi-GPU path broadcasts straight into the library's device
    buffer through a torch view of the raw pointer (__cuda_array_interface__):
    the view must alias the container"""
    import torch
    import bench
    img = jobs.synth_image(64, 32, 3)
    src = ea.Source.load(ea.facet_spec(ea.SPHERICAL, 64, 32, 360.0), img, 3)
    ptr, n = src.device_ptr()
    t = torch.as_tensor(bench._DevBuf(ptr, n), device=torch.device("cuda:0"))
    host = src.download().reshape(-1)
    assert t.numel() == host.size and np.array_equal(t.cpu().numpy(), host)
    # a write through the view is what a broadcast does
    t.mul_(2.0)
    torch.cuda.synchronize()
    assert np.array_equal(src.download().reshape(-1), host * 2.0)


@pytest.mark.parametrize("tw,th", [(1, 1), (3, 2), (17, 5), (513, 3), (64, 1), (65, 4), (129, 9)])
def test_ragged_target_sizes(tw, th):
    """targets narrower than a vector, a tile, a segment; odd leftovers"""
    img = jobs.synth_image(128, 64, 3)
    o = jobs.OracleSource(euo.SPHERICAL, 128, 64, 360.0, img, 3)
    g = ea.Source.adopt(ea.facet_spec(ea.SPHERICAL, 128, 64, 360.0), o.container, 3, o.bc[0], o.bc[1])
    for twine in (0, 2):
        a = ea.arguments(ea.RECTILINEAR, tw, th, 80.0, yaw=15, pitch=-8, roll=3, spline_degree=3, twine=twine)
        got, ref = ea.render(a, g), jobs.oracle_render(a, o)
        assert (jobs.bits(got) == jobs.bits(ref)).all(), (tw, th, twine)
    # the same through the general kernel and the 2-D tile kernels
    a = ea.arguments(ea.SPHERICAL, tw, th, 360.0, spline_degree=3)
    ref = jobs.oracle_render(a, o)
    assert (jobs.bits(ea.render(a, g)) == jobs.bits(ref)).all()


def test_error_codes_not_aborts():
    img = jobs.synth_image(64, 32, 3)
    src = ea.Source.load(ea.facet_spec(ea.SPHERICAL, 64, 32, 360.0), img, 1)
    # a projection id the library has no stepper for: an error code, the process lives
    a = ea.arguments(ea.SPHERICAL, 32, 16, 360.0, spline_degree=1)
    a.projection = 99
    with pytest.raises(ea.EuError):
        ea.render(a, src)
    # cubemap target that is not 1:6
    a = ea.arguments(ea.CUBEMAP, 32, 100, 90.0, spline_degree=1)
    with pytest.raises(ea.EuError, match="-2"):
        ea.render(a, src)
    # and the library still works afterwards
    a = ea.arguments(ea.SPHERICAL, 32, 16, 360.0, spline_degree=1)
    assert ea.render(a, src).shape == (16, 32, 3)


def test_device_setup_of_images_narrower_than_the_frame():
    """zimt braces slice by slice, outward; a 2-pixel-wide image gets slices copied from slices filled
    before. The device set-up does the same for such cores (brace_seq_kernel) - since round 3 also for a
    full-sphere image that small (tests/test_gpu_round3_switches.py has the sizes and degrees)"""
    img = jobs.synth_image(2, 9, 3)
    g = ea.Source.load(ea.facet_spec(ea.RECTILINEAR, 2, 9, 60.0), img, 3)
    o = jobs.OracleSource(euo.RECTILINEAR, 2, 9, 60.0, img, 3)
    assert (jobs.bits(g.download().reshape(-1)) == jobs.bits(np.asarray(o.container).reshape(-1))).all()
    a = ea.arguments(ea.RECTILINEAR, 40, 30, 50.0, spline_degree=3)
    assert (jobs.bits(ea.render(a, g)) == jobs.bits(jobs.oracle_render(a, o))).all()
    img = jobs.synth_image(4, 2, 3)
    gs = ea.Source.load(ea.facet_spec(ea.SPHERICAL, 4, 2, 360.0), img, 3)
    os_ = jobs.OracleSource(euo.SPHERICAL, 4, 2, 360.0, img, 3)
    assert (jobs.bits(gs.download().reshape(-1)) == jobs.bits(np.asarray(os_.container).reshape(-1))).all()
    # one pixel wide is fine (a constant along that axis)
    img = jobs.synth_image(1, 20, 3)
    g1 = ea.Source.load(ea.facet_spec(ea.RECTILINEAR, 1, 20, 5.0), img, 3)
    o1 = jobs.OracleSource(euo.RECTILINEAR, 1, 20, 5.0, img, 3)
    assert (jobs.bits(g1.download().reshape(-1)) == jobs.bits(np.asarray(o1.container).reshape(-1))).all()
