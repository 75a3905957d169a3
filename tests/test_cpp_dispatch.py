"""The C++ host mirror (include/eu_dispatch.hpp: arguments / facet_spec /
get_dispatch()->payload()) compiles against the C ABI, fails loudly without a
GPU and - on a GPU - produces the oracle's frame."""
import math
import os
import subprocess

import numpy as np
import pytest

import envutil_amd as ea

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "envutil_amd", "build", "dispatch_demo")


def build_demo():
    if not os.path.exists(ea.lib_path()):
        ea.build()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "csrc", "dispatch_demo.cc"), "-o", EXE,
                           "-L" + os.path.join(ROOT, "envutil_amd", "lib"), "-leu_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "envutil_amd", "lib")])


def fnv1a(a):
    h = 1469598103934665603
    a = np.ascontiguousarray(a)
    if a.dtype != np.uint32:
        a = a.astype(np.float32, copy=False).view(np.uint32)
    for u in a.ravel().tolist():
        h = ((h ^ u) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return f"{h:016x}"


def test_cpp_mirror_builds_and_fails_loudly_without_gpu():
    build_demo()
    if ea.device_count() > 0:
        pytest.skip("a HIP device is present")
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("twine", [0, 2])
def test_cpp_payload_matches_oracle(twine):
    import euo
    import jobs
    build_demo()
    r = subprocess.run([EXE] + (["twine"] if twine else []), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    got = r.stdout.split("fnv1a")[1].strip()
    sw, sh, tw = 256, 128, 64
    y, x, c = np.indices((sh, sw, 3))
    img = (np.float32(0.5) + np.float32(0.25) * ((x * 7 + y * 13 + c * 29) % 97).astype(np.float32)
           / np.float32(97.0)).astype(np.float32)
    o = jobs.OracleSource(euo.SPHERICAL, sw, sh, 360.0, img, 3)
    a = ea.arguments(ea.CUBEMAP, tw, 6 * tw, 90.0, yaw=math.degrees(0.3), pitch=math.degrees(-0.2),
                     roll=math.degrees(0.1), spline_degree=3, twine=twine)
    # the demo passes radians straight through; the Python mirror converts degrees
    a.yaw, a.pitch, a.roll = math.degrees(0.3), math.degrees(-0.2), math.degrees(0.1)
    assert fnv1a(jobs.oracle_render(a, o)) == got


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [("crop",), ("screen",), ("crop", "screen", "twine")])
def test_cpp_payload_crop_and_tethered(mode):
    """args.store_cropped / args.tethered through the C++ mirror"""
    import euo
    import jobs
    build_demo()
    r = subprocess.run([EXE] + list(mode), capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    got = r.stdout.split("fnv1a")[1].strip()
    sw, sh, tw = 256, 128, 64
    y, x, c = np.indices((sh, sw, 3))
    img = (np.float32(0.5) + np.float32(0.25) * ((x * 7 + y * 13 + c * 29) % 97).astype(np.float32)
           / np.float32(97.0)).astype(np.float32)
    o = jobs.OracleSource(euo.SPHERICAL, sw, sh, 360.0, img, 3)
    a = ea.arguments(ea.CUBEMAP, tw, 6 * tw, 90.0, yaw=math.degrees(0.3), pitch=math.degrees(-0.2),
                     roll=math.degrees(0.1), spline_degree=3, twine=2 if "twine" in mode else 0,
                     crop=(5, 60, 30, 301) if "crop" in mode else None, tethered="screen" in mode)
    assert fnv1a(jobs.oracle_render(a, o)) == got
