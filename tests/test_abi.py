"""CPU-side checks of the product library: it builds, loads, exports every
symbol include/eu_hip.h declares, its host set-up arithmetic agrees with the
oracle, and without a GPU the compute entry points fail loudly."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import envutil_amd as ea
import euo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    if not os.path.exists(ea.lib_path()):
        ea.build()
    return ea.lib()


def test_exports_every_declared_symbol(L):
    hdr = open(os.path.join(ROOT, "include", "eu_hip.h")).read()
    names = sorted(set(re.findall(r"\b(eu_hip_\w+)\s*\(", hdr)))
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), f"{n} declared in eu_hip.h but not exported"


@pytest.mark.parametrize("prj", range(7))
def test_extent_and_step_match_oracle(L, prj):
    for w, h, hfov in [(1024, 512, 360.0), (640, 480, 90.0), (300, 200, 65.5), (64, 384, 90.0)]:
        hf = math.radians(hfov)
        assert np.array_equal(ea.get_extent(prj, w, h, hf), euo.get_extent(prj, w, h, hf))
        assert ea.get_step(prj, w, h, hf) == euo.lib().euo_get_step(prj, w, h, hf)


def test_make_spread_matches_oracle(L):
    for w, h, d, sigma, th in [(2, 0, 1.0, 0.0, 0.0), (3, 3, 1.0, 0.0, 0.0), (1, 0, 1.0, 0, 0),
                               (5, 4, 1.5, 0.0, 0.0), (7, 7, 1.0, 1.5, 0.0),
                               (7, 7, 1.0, 1.5, 0.02), (4, 0, 2.0, 0.7, 0.05)]:
        a = ea.make_spread(w, h, d, sigma, th)
        b = euo.make_spread(w, h, d, sigma, th)
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (w, h)
    # box filter: weights sum to one, taps centred (envutil_main.cc:1262-1268)
    t = ea.make_spread(3, 3)
    assert len(t) == 9 and abs(t[:, 2].sum() - 1) < 1e-6 and abs(t[:, 0].sum()) < 1e-6


def test_cubemap_metrics_match_oracle(L):
    for face, fov, smin, tile in [(2048, 90.0, 8, 64), (64, 90.0, 8, 64), (512, 90.0, 4, 64),
                                  (100, 90.0, 8, 32), (256, 100.0, 8, 64), (333, 93.0, 16, 64)]:
        m = ea.cubemap_metrics(face, math.radians(fov), smin, tile)
        o = euo.metrics(face, math.radians(fov), smin, tile)
        assert m["section_px"] == o.section_px and m["left_frame_px"] == o.left_frame_px
        assert m["refc_md"] == o.refc_md and m["model_to_px"] == o.model_to_px
    # face 2048, support 8, tile 64
    m = ea.cubemap_metrics(2048)
    assert (m["section_px"], m["left_frame_px"], m["model_to_px"], m["refc_md"]) == (2112, 32, 1024.0, 1.03125)


def test_container_geometry_matches_oracle(L):
    for deg in range(10):
        for b0, b1 in [(1, 2), (2, 2), (0, 0), (3, 3), (1, 1), (4, 4)]:
            g = ea.container_geometry(deg, b0, b1, 100, 50)
            o = (C.c_long * 6)()
            euo.lib().euo_spline_geometry(deg, b0, b1, 100, 50, o)
            assert [g.shape[0], g.shape[1], g.left[0], g.left[1], g.right[0], g.right[1]] == list(o)
    # headline container
    g = ea.container_geometry(3, ea.BC_PERIODIC, ea.BC_REFLECT, 16384, 8192)
    assert (g.shape[0], g.shape[1]) == (16384 + 2 + 3, 8192 + 2 + 2)


def test_no_gpu_means_loud_failure(L):
    if ea.device_count() > 0:
        pytest.skip("a HIP device is present")
    fct = ea.facet_spec(ea.SPHERICAL, 64, 32, 360.0)
    with pytest.raises(ea.EuError, match="no HIP device"):
        ea.Source.load(fct, np.zeros((32, 64, 3), np.float32), 1)


def test_argument_errors_do_not_abort(L):
    e = np.zeros(4)
    assert L.eu_hip_get_extent(99, 10, 10, 1.0, e.ctypes.data_as(C.c_void_p)) == -2
    assert L.eu_hip_container_geometry(99, 0, 0, 10, 10, None) == -2
    assert b"" != L.eu_hip_last_error()


def test_ctypes_structs_match_the_header(tmp_path):
    """sizeof / offsetof of every struct of include/eu_hip.h, as gcc lays them
    out, against the ctypes mirrors in envutil_amd/api.py"""
    import subprocess
    from envutil_amd import api
    structs = {"eu_facet": api.Facet, "eu_container": api.Container, "eu_target": api.Target}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "eu_hip.h"', 'int main(void) {']
    for name, cls in structs.items():
        lines.append(f'  printf("{name} size %zu\\n", sizeof({name}));')
        for fld, _ in cls._fields_:
            lines.append(f'  printf("{name} {fld} %zu\\n", offsetof({name}, {fld}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split("\n")
    seen = 0
    for ln in out:
        if not ln:
            continue
        name, fld, val = ln.split()
        cls = structs[name]
        if fld == "size":
            assert C.sizeof(cls) == int(val), (name, C.sizeof(cls), val)
        else:
            assert getattr(cls, fld).offset == int(val), (name, fld)
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in structs.values())
