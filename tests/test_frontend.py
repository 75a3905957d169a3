"""The OpenImageIO-free front end (include/eu_frontend.hpp): envutil command lines - the
reference README's own examples among them - and PTO scripts are parsed into project::args by a
small C++ program (tests/csrc/frontend_demo.cc) and compared with the jobs the parity tests
build by hand (envutil_amd.arguments / facet_spec) and with values worked out here from the
reference's rules (envutil_main.cc:178-1251 arguments::init, :1405-1616 twine_setup,
pto.h:63-200). No GPU needed: the set-up functions of the library run on the host."""
import json
import math
import os
import subprocess

import numpy as np
import pytest

import envutil_amd as ea

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "envutil_amd", "build", "frontend_demo")
RAD = math.pi / 180.0


@pytest.fixture(scope="module")
def demo():
    if not os.path.exists(ea.lib_path()):
        ea.build()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "csrc", "frontend_demo.cc"), "-o", EXE,
                           "-L" + os.path.join(ROOT, "envutil_amd", "lib"), "-leu_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "envutil_amd", "lib")])

    def run(argv, images, cwd=None, render=False):
        env = dict(os.environ, EU_TEST_IMAGES=";".join(f"{k}={w}x{h}x{c}" for k, (w, h, c) in images.items()))
        if render:
            env["EU_TEST_RENDER"] = "1"
        r = subprocess.run([EXE] + list(argv), capture_output=True, text=True, env=env, cwd=cwd, timeout=300)
        assert r.stdout, r.stderr
        if render:
            head, _, tail = r.stdout.partition("\nrc ")
            return json.loads(head), "rc " + tail
        return json.loads(r.stdout)
    return run


def f32(x):
    return float(np.float32(x))


def test_single_facet_command_line_equals_the_hand_built_job(demo):
    """envutil --facet pano.tif spherical 360 0 0 0 --projection cubemap --width 512 --degree 3 ..."""
    j = demo(["--facet", "pano.tif", "spherical", "360", "0", "0", "0", "--output", "cube.exr",
              "--projection", "cubemap", "--width", "512", "--hfov", "90", "--degree", "3",
              "--yaw", "33.3", "--pitch", "-12.5", "--roll", "7", "--twine", "0"],
             {"pano.tif": (4000, 2000, 3)})
    assert j["ok"] and j["output"] == "cube.exr"
    a = ea.arguments(ea.CUBEMAP, 512, 3072, 90.0, spline_degree=3)
    assert (j["projection"], j["width"], j["height"], j["degree"], j["prefilter"]) == (ea.CUBEMAP, 512, 3072, 3, 3)
    assert [j["x0"], j["x1"], j["y0"], j["y1"]] == [float(v) for v in a.extent] and j["step"] == a.step
    # ArgParse's get<float>: the angles pass through float before they become radians
    assert j["yaw"] == f32(33.3) * RAD and j["pitch"] == f32(-12.5) * RAD and j["roll"] == 7.0 * RAD
    assert j["twine"] == 0 and j["spread"] == [] and j["nchannels"] == 3 and j["solo"] == 0
    f = j["facets"][0]
    fs = ea.facet_spec(ea.SPHERICAL, 4000, 2000, 360.0)
    assert f["hfov"] == 2 * math.pi and (f["width"], f["height"]) == (4000, 2000) and f["window"] == [4000, 2000, 0, 0]
    assert [f["x0"], f["x1"], f["y0"], f["y1"]] == list(ea.get_extent(ea.SPHERICAL, 4000, 2000, 2 * math.pi))
    assert f["step"] == ea.get_step(ea.SPHERICAL, 4000, 2000, 2 * math.pi) == fs.c_struct().step
    assert f["brighten"] == 1.0 and f["asset_key"] == "pano.tif"


def test_defaults(demo):
    """no --width: 1024; spherical: height = width / 2; no --projection: rectilinear, hfov 90,
    degree 1, support_min 8, tile_size 64 (envutil_main.cc:438-505)"""
    j = demo(["--facet", "a.jpg", "rectilinear", "65", "10", "-5", "2", "--output", "o.jpg"], {"a.jpg": (3000, 2000, 3)})
    assert (j["projection"], j["width"], j["height"]) == (ea.RECTILINEAR, 1024, 1024)
    assert j["hfov"] == 90.0 * RAD and (j["degree"], j["prefilter"], j["support_min"], j["tile_size"]) == (1, 1, 8, 64)
    assert j["facets"][0]["yaw"] == 10 * RAD and j["facets"][0]["pitch"] == -5 * RAD and j["facets"][0]["roll"] == 2 * RAD
    j = demo(["--facet", "a.jpg", "rectilinear", "65", "0", "0", "0", "--output", "o.jpg", "--projection", "spherical",
              "--width", "1001", "--hfov", "360"], {"a.jpg": (3000, 2000, 3)})
    assert (j["width"], j["height"]) == (1002, 501)


@pytest.mark.parametrize("tw,degree,nfacets,want", [
    (250, 1, 1, ("down", None)),          # minification: twine = min(twine_max, int(1 + 1 / mag))
    (1200, 3, 1, ("down", None)),
    (8000, 1, 1, ("up_bilinear", None)),  # magnification, bilinear: twine = min(5, int(1 + mag)), width = mag
    (8000, 3, 1, ("up_spline1", None)),   # magnification, spline, one facet: mag >= 2 -> 1, else 2
    (3000, 3, 1, ("up_spline1", None)),
    (8000, 3, 2, ("up_spline_multi", None)),
])
def test_automatic_twining(demo, tw, degree, nfacets, want):
    """arguments::twine_setup without --twine (envutil_main.cc:1437-1560): mag = smallest facet
    step / target step"""
    argv = ["--output", "o.exr", "--projection", "spherical", "--width", str(tw), "--hfov", "360", "--degree", str(degree)]
    for i in range(nfacets):
        argv += ["--facet", f"f{i}.tif", "spherical", "360", str(30 * i), "0", "0"]
    j = demo(argv, {f"f{i}.tif": (2000, 1000, 3) for i in range(nfacets)})
    src_step, trg_step = 2 * math.pi / 2000, 2 * math.pi / (tw + (tw & 1))
    mag = src_step / trg_step
    kind = want[0]
    if kind == "down":
        twine, width = min(8, int(1.0 + 1.0 / mag)), 1.0
    elif kind == "up_bilinear":
        twine, width = min(5, int(1.0 + mag)), f32(mag)
    elif kind == "up_spline1":
        twine, width = (2 if mag < 2.0 else 1), 1.0
    else:
        twine, width = 3, 1.0
    assert j["twine"] == twine and j["twine_width"] == pytest.approx(width, rel=1e-7)
    spread = ea.make_spread(twine, twine, j["twine_width"], 0.0, 0.0)
    assert np.allclose(np.array(j["spread"], np.float32), spread, rtol=0, atol=0)


def test_explicit_twine_and_density(demo):
    j = demo(["--facet", "a.tif", "spherical", "360", "0", "0", "0", "--output", "o", "--twine", "3", "--twine_width", "1.5",
              "--twine_sigma", "1.2", "--twine_threshold", "0.01", "--twine_density", "2"], {"a.tif": (256, 128, 3)})
    assert j["twine"] == 6
    want = ea.make_spread(6, 6, 1.5, f32(1.2), f32(0.01))
    assert np.array_equal(np.array(j["spread"], np.float32), want)


PTO = """# hugin project file
p f2 w4000 h2000 v360  E11.5 R0 S100,3900,50,1950 n"TIFF_m c:LZW"
m i2

# image lines
i w3000 h2000 f0 v50 Ra0 Eev12 y0 p0 r0 a0.01 b-0.03 c0.02 d5 e-3 g0 t0 TrX0 TrY0 TrZ0 n"img0.jpg"
i w3000 h2000 f0 v=0 Eev13 y40.5 p-3.25 r1.5 a=0 b=0 c=0 d0 e0 g12 t-6 n"img1.jpg"
i w2000 h2000 f3 v180 Eev12.5 y-90 p0 r0 n"fish.tif"
"""


def test_pto_script(demo, tmp_path):
    """a PTO with a p-line (projection, size, hfov, output crop), three i-lines with back
    references, lens polynomial, shift, shear and exposure values"""
    (tmp_path / "pano.pto").write_text(PTO)
    j = demo(["--pto", "pano.pto", "--output", "out.tif", "--degree", "1", "--twine", "0"],
             {"img0.jpg": (3000, 2000, 3), "img1.jpg": (3000, 2000, 3), "fish.tif": (2000, 2000, 4)}, cwd=str(tmp_path))
    assert j["ok"], j
    # the p-line sets the target (no --width on the command line)
    assert (j["projection"], j["width"], j["height"]) == (ea.SPHERICAL, 4000, 2000) and j["hfov"] == 360 * RAD
    assert j["store_cropped"] == 1 and j["crop"] == [100, 3900, 50, 1950]
    assert [j["x0"], j["x1"], j["y0"], j["y1"]] == list(ea.get_extent(ea.SPHERICAL, 4000, 2000, 2 * math.pi))
    assert j["nfacets"] == 3 and j["nchannels"] == 4 and j["solo"] == -1
    f0, f1, f2 = j["facets"]
    assert (f0["projection"], f1["projection"], f2["projection"]) == (ea.RECTILINEAR, ea.RECTILINEAR, ea.FISHEYE)
    assert f0["hfov"] == 50 * RAD and f1["hfov"] == 50 * RAD          # v=0: the value of image 0
    assert (f1["a"], f1["b"], f1["c"]) == (0.01, -0.03, 0.02) and f1["has_lcp"] == 1
    assert (f0["h"], f0["v"], f0["has_shift"]) == (5.0, -3.0, 1) and f1["has_shift"] == 0
    assert f1["yaw"] == 40.5 * RAD and f1["pitch"] == -3.25 * RAD and f1["roll"] == 1.5 * RAD
    assert f1["shear_g"] == 12 / 2000 and f1["shear_t"] == -6 / 3000 and f1["has_shear"] == 1
    # process_geometry: s = the smaller half extent (envutil_basic.h:499-521)
    e = ea.get_extent(ea.RECTILINEAR, 3000, 2000, 50 * RAD)
    assert f0["s"] == min(abs(e[1] - e[0]), abs(e[3] - e[2])) / 2
    # Eev -> brighten = 2^(Eev - Eev_out); the p-line's E is not 'Eev': the mean of the facets' counts
    mean = f32((12 + 13 + 12.5) / 3)
    for f, eev in ((f0, 12), (f1, 13), (f2, 12.5)):
        assert f["brighten"] == pytest.approx(2.0 ** (eev - mean), rel=1e-6)
    assert f2["hfov"] == math.pi and f2["nchannels"] == 4 and f2["filename"] == "fish.tif"
    # what the hand-built job of the parity tests reads: the same numbers
    fs = ea.facet_spec(ea.RECTILINEAR, 3000, 2000, 50.0, lens=dict(a=0.01, b=-0.03, c=0.02, h=5.0, v=-3.0)).c_struct()
    assert (fs.a, fs.b, fs.c, fs.h, fs.v) == (f0["a"], f0["b"], f0["c"], f0["h"], f0["v"])
    assert fs.step == f0["step"]


def test_readme_unstitching_examples(demo, tmp_path):
    """README.md:620-622, :631-633, :705-707, :725-727: a stitched panorama added as a facet of
    its own - by --facet, by a PTO line, as a cropped image (W clause), by the Pano shortcut"""
    (tmp_path / "pano.pto").write_text(PTO)
    imgs = {"img0.jpg": (3000, 2000, 3), "img1.jpg": (3000, 2000, 3), "fish.tif": (2000, 2000, 4),
            "pano.tif": (4000, 2000, 3)}
    j = demo(["--pto", "pano.pto", "--facet", "pano.tif", "spherical", "360", "0", "0", "0",
              "--solo", "3", "--single", "1", "--output", "image1.tif"], imgs, cwd=str(tmp_path))
    assert j["ok"] and j["nfacets"] == 4 and j["solo"] == 3 and j["single"] == 1
    assert j["facets"][3]["filename"] == "pano.tif" and j["facets"][3]["projection"] == ea.SPHERICAL
    # --single: the target takes the facet's geometry
    assert (j["projection"], j["width"], j["height"]) == (ea.RECTILINEAR, 3000, 2000) and j["hfov"] == 50 * RAD
    assert j["yaw"] == 40.5 * RAD
    j2 = demo(["--pto", "pano.pto", "--pto_line", 'i f4 v360 n"pano.tif" Eev13.5',
               "--solo", "3", "--single", "1", "--output", "image1.tif"], imgs, cwd=str(tmp_path))
    assert j2["ok"] and j2["facets"][3]["projection"] == ea.SPHERICAL and j2["facets"][3]["hfov"] == 2 * math.pi
    # cropped input: the file holds the window, w / h the whole image
    imgs["pano.tif"] = (1980, 1490, 3)
    j3 = demo(["--pto", "pano.pto", "--pto_line", 'i f4 v360 n"pano.tif" W20,2000,10,1500 w4000 h2000',
               "--solo", "3", "--split", "img_%02d.tif"], imgs, cwd=str(tmp_path))
    assert j3["ok"], j3
    f = j3["facets"][3]
    assert (f["width"], f["height"]) == (4000, 2000) and f["window"] == [1980, 1490, 20, 10]
    # the Pano shortcut takes everything from the p-line (here with its S crop) and sets solo
    imgs["pano.tif"] = (3800, 1900, 3)
    j4 = demo(["--pto", "pano.pto", "--pto_line", 'i Pano"pano.tif"', "--split", "img_%02d.tif"], imgs, cwd=str(tmp_path))
    assert j4["ok"], j4
    f = j4["facets"][3]
    assert j4["solo"] == 3 and f["projection"] == ea.SPHERICAL and f["hfov"] == 2 * math.pi
    assert (f["width"], f["height"]) == (4000, 2000) and f["window"] == [3800, 1900, 100, 50]


def test_errors_are_reported_not_asserted(demo):
    for argv, what in ((["--output", "o"], "no facet"), (["--facet", "a", "spherical", "360", "0", "0", "0"], "no --output"),
                       (["--facet", "nope.tif", "spherical", "360", "0", "0", "0", "--output", "o"], "failed to open"),
                       (["--facet", "a.tif", "mercator", "360", "0", "0", "0", "--output", "o"], "unknown facet projection"),
                       (["--bogus", "1"], "unknown option")):
        j = demo(argv, {"a.tif": (64, 32, 3)})
        assert j["ok"] is False and what in j["error"], (argv, j)


@pytest.mark.gpu
def test_command_line_to_pixels(demo, tmp_path):
    """front end -> get_dispatch()->payload() -> HIP kernels, against the oracle on the job the
    command line describes: two facets from a PTO (lens polynomial on one), automatic twining"""
    import euo
    import jobs
    from test_cpp_dispatch import fnv1a
    (tmp_path / "two.pto").write_text(
        'p f2 w300 h150 v360 n"TIFF"\n'
        'i w200 h150 f0 v70 y10 p5 r2 a0.01 b-0.03 c0.02 d0 e0 n"a.tif"\n'
        'i w160 h160 f3 v170 y-100 p-20 r0 n"b.tif"\n')
    j, tail = demo(["--pto", "two.pto", "--output", "o.tif", "--degree", "1"],
                   {"a.tif": (200, 150, 3), "b.tif": (160, 160, 3)}, cwd=str(tmp_path), render=True)
    assert j["ok"] and "rc 0" in tail, tail
    got = tail.split("fnv1a")[1].strip()

    def pixels(k, w, h, n):
        y, x, c = np.indices((h, w, n))
        return (np.float32(0.5) + np.float32(0.25) * ((x * 7 + y * 13 + c * 29 + k * 5) % 97).astype(np.float32)
                / np.float32(97.0)).astype(np.float32)
    o0 = jobs.OracleSource(euo.RECTILINEAR, 200, 150, 70.0, pixels(0, 200, 150, 3), 1, yaw=10.0, pitch=5.0, roll=2.0,
                           lens=dict(a=0.01, b=-0.03, c=0.02))
    o1 = jobs.OracleSource(euo.FISHEYE, 160, 160, 170.0, pixels(1, 160, 160, 3), 1, yaw=-100.0, pitch=-20.0)
    a = ea.arguments(ea.SPHERICAL, 300, 150, 360.0, spline_degree=1, twine=j["twine"], twine_width=j["twine_width"])
    assert j["twine"] >= 1 and len(j["spread"]) == len(a.twine_spread)
    assert fnv1a(jobs.oracle_render(a, [o0, o1])) == got


@pytest.mark.gpu
def test_hdr_merge_command_line_to_pixels(demo, tmp_path):
    """--synopsis hdr_merge on an exposure bracket described by a PTO (Eev per image -> brighten,
    envutil_main.cc:1003-1060): front end -> payload() -> the HIP hdr_merge kernel == the oracle"""
    import euo
    import jobs
    from test_cpp_dispatch import fnv1a
    (tmp_path / "bracket.pto").write_text(
        'p f0 w240 h160 v80 n"TIFF"\n'
        'i w200 h150 f0 v70 y0 p0 r0 Eev10 n"a.tif"\n'
        'i w200 h150 f0 v70 y0.5 p0 r0 Eev12 n"b.tif"\n'
        'i w200 h150 f0 v70 y0 p0.5 r0 Eev14 n"c.tif"\n')
    j, tail = demo(["--pto", "bracket.pto", "--output", "o.tif", "--degree", "1", "--twine", "0",
                    "--synopsis", "hdr_merge"],
                   {"a.tif": (200, 150, 3), "b.tif": (200, 150, 3), "c.tif": (200, 150, 3)}, cwd=str(tmp_path), render=True)
    assert j["ok"] and j["synopsis"] == "hdr_merge" and "rc 0" in tail, tail
    got = tail.split("fnv1a")[1].strip()
    br = [f["brighten"] for f in j["facets"]]
    assert br == [f32(2.0 ** (e - 12.0)) for e in (10, 12, 14)]

    def pixels(k, w, h, n):
        y, x, c = np.indices((h, w, n))
        return (np.float32(0.5) + np.float32(0.25) * ((x * 7 + y * 13 + c * 29 + k * 5) % 97).astype(np.float32)
                / np.float32(97.0)).astype(np.float32)
    views = [(0.0, 0.0), (0.5, 0.0), (0.0, 0.5)]
    os_ = [jobs.OracleSource(euo.RECTILINEAR, 200, 150, 70.0, pixels(k, 200, 150, 3), 1, yaw=views[k][0], pitch=views[k][1],
                             brighten=br[k]) for k in range(3)]
    a = ea.arguments(ea.RECTILINEAR, 240, 160, 80.0, spline_degree=1, synopsis="hdr_merge")
    assert fnv1a(jobs.oracle_render(a, os_)) == got


@pytest.mark.gpu
def test_pto_mosaic_with_translation_to_pixels(demo, tmp_path):
    """TrX/TrY/TrZ/Tpy/Tpp of a PTO image line reach the kernel: front end -> payload() -> generic_stepper
    on the device == the oracle"""
    import euo
    import jobs
    from test_cpp_dispatch import fnv1a
    (tmp_path / "mosaic.pto").write_text(
        'p f0 w260 h180 v90 n"TIFF"\n'
        'i w200 h150 f0 v60 y-10 p0 r0 n"a.tif"\n'
        'i w200 h150 f0 v60 y12 p2 r1 TrX0.2 TrY-0.05 TrZ0.1 Tpy5 Tpp-3 n"b.tif"\n')
    j, tail = demo(["--pto", "mosaic.pto", "--output", "o.tif", "--degree", "1", "--twine", "0"],
                   {"a.tif": (200, 150, 3), "b.tif": (200, 150, 3)}, cwd=str(tmp_path), render=True)
    assert j["ok"] and "rc 0" in tail, tail
    assert j["facets"][1]["tr"] == [0.2, -0.05, -0.1]          # TrZ changes sign, envutil_main.cc:788
    got = tail.split("fnv1a")[1].strip()

    def pixels(k, w, h, n):
        y, x, c = np.indices((h, w, n))
        return (np.float32(0.5) + np.float32(0.25) * ((x * 7 + y * 13 + c * 29 + k * 5) % 97).astype(np.float32)
                / np.float32(97.0)).astype(np.float32)
    o0 = jobs.OracleSource(euo.RECTILINEAR, 200, 150, 60.0, pixels(0, 200, 150, 3), 1, yaw=-10.0)
    o1 = jobs.OracleSource(euo.RECTILINEAR, 200, 150, 60.0, pixels(1, 200, 150, 3), 1, yaw=12.0, pitch=2.0, roll=1.0,
                           translation=dict(x=0.2, y=-0.05, z=-0.1, tp_y=5.0, tp_p=-3.0))
    a = ea.arguments(ea.RECTILINEAR, 260, 180, 90.0, spline_degree=1)
    assert fnv1a(jobs.oracle_render(a, [o0, o1])) == got


@pytest.mark.gpu
def test_single_command_line_to_pixels(demo, tmp_path):
    """--single 1 on a PTO whose second image has a lens polynomial, a shift and a translation: the front end
    takes the facet's geometry over as target geometry, payload() hands its lens and translation parameters to
    the library (eu_target.single), every facet goes through generic_stepper == the oracle"""
    import euo
    import jobs
    from test_cpp_dispatch import fnv1a
    (tmp_path / "un.pto").write_text(
        'p f2 w400 h200 v360 n"TIFF"\n'
        'i w200 h150 f0 v60 y-10 p0 r0 n"a.tif"\n'
        'i w180 h140 f0 v55 y8 p3 r2 a0.01 b-0.02 c0.015 d4 e-3 TrX0.1 TrY-0.04 TrZ0.05 Tpy2 Tpp-1 n"b.tif"\n')
    j, tail = demo(["--pto", "un.pto", "--single", "1", "--output", "o.tif", "--degree", "1", "--twine", "0"],
                   {"a.tif": (200, 150, 3), "b.tif": (180, 140, 3)}, cwd=str(tmp_path), render=True)
    assert j["ok"] and "rc 0" in tail, tail
    assert (j["projection"], j["width"], j["height"]) == (ea.RECTILINEAR, 180, 140)
    got = tail.split("fnv1a")[1].strip()
    fb = j["facets"][1]

    def pixels(k, w, h, n):
        y, x, c = np.indices((h, w, n))
        return (np.float32(0.5) + np.float32(0.25) * ((x * 7 + y * 13 + c * 29 + k * 5) % 97).astype(np.float32)
                / np.float32(97.0)).astype(np.float32)
    lens = dict(a=fb["a"], b=fb["b"], c=fb["c"], h=fb["h"], v=fb["v"])
    tr = dict(x=0.1, y=-0.04, z=-0.05, tp_y=2.0, tp_p=-1.0)
    o0 = jobs.OracleSource(euo.RECTILINEAR, 200, 150, 60.0, pixels(0, 200, 150, 3), 1, yaw=-10.0)
    o1 = jobs.OracleSource(euo.RECTILINEAR, 180, 140, 55.0, pixels(1, 180, 140, 3), 1, yaw=8.0, pitch=3.0, roll=2.0, lens=lens,
                           translation=tr)
    a = ea.arguments.for_single(ea.facet_spec(ea.RECTILINEAR, 180, 140, 55.0, yaw=8.0, pitch=3.0, roll=2.0, lens=lens,
                                              translation=tr), spline_degree=1)
    a.single_oracle = o1
    assert fnv1a(jobs.oracle_render(a, [o0, o1])) == got


def test_twf_file_is_the_tap_table(demo, tmp_path):
    """--twf_file (read_twf_file, envutil_main.cc:1357-1403): x y weight triples from a text file; x and y
    are scaled by twine_width, the weights divided by their sum under --twine_normalize; twine becomes 1"""
    (tmp_path / "k.twf").write_text("-0.25 -0.25 1\n0.25 -0.25 2\n-0.25 0.25 3\n0.25 0.25 2\n")
    base = ["--facet", "pano.tif", "spherical", "360", "0", "0", "0", "--output", "o.exr", "--width", "256",
            "--twf_file", "k.twf", "--twine_width", "1.5"]
    j = demo(base, {"pano.tif": (4000, 2000, 3)}, cwd=str(tmp_path))
    assert j["ok"] and j["twine"] == 1
    w = np.float32(1.5)
    want = [[f32(np.float32(x) * w), f32(np.float32(y) * w), float(k)] for x, y, k in
            ((-0.25, -0.25, 1), (0.25, -0.25, 2), (-0.25, 0.25, 3), (0.25, 0.25, 2))]
    assert j["spread"] == want
    j = demo(base + ["--twine_normalize"], {"pano.tif": (4000, 2000, 3)}, cwd=str(tmp_path))
    assert [t[2] for t in j["spread"]] == [f32(np.float32(k) / 8.0) for k in (1, 2, 3, 2)]


def test_k_lines_become_mask_polygons(demo, tmp_path):
    """k-lines (envutil_main.cc:827-905): the polygon of an exclude mask goes to its image's facet_spec, the
    facet gains an alpha channel and a mask-specific asset key; an S clause is the lens crop"""
    (tmp_path / "m.pto").write_text(
        'p f2 w300 h150 v360 n"TIFF"\n'
        'i w200 h150 f0 v70 y10 p5 r2 n"a.tif"\n'
        'i w160 h160 f3 v170 y-100 p-20 r0 S10,150,12,148 n"b.tif"\n'
        'k i0 t0 p"30 20 120.5 25 140 110 40 100"\n'
        'k i0 t1 p"1 2 3 4 5 6"\n')
    j = demo(["--pto", "m.pto", "--output", "o.tif", "--mask_for", "1"],
             {"a.tif": (200, 150, 3), "b.tif": (160, 160, 3)}, cwd=str(tmp_path))
    assert j["ok"] and j["nchannels"] == 4
    fa, fb = j["facets"]
    assert fa["has_pto_mask"] == 1 and fa["nchannels"] == 4 and fa["asset_key"] == "a.tif.m.pto.0.1"
    assert fa["masks"] == [{"variant": 0, "xy": [30, 20, 120.5, 25, 140, 110, 40, 100]},
                           {"variant": 1, "xy": [1, 2, 3, 4, 5, 6]}]
    assert fb["has_lens_crop"] == 1 and fb["lens_crop"] == [10, 150, 12, 148] and fb["nchannels"] == 4 and fb["masks"] == []
    assert (fa["masked"], fb["masked"]) == (0, 1)


def test_photo_and_metadata_facets_default_to_rectilinear_65(demo):
    """--photo IMAGE is --facet IMAGE metadata -1 0 0 0; without metadata in the file: rectilinear, 65 degrees
    (get_image_metrics, envutil_basic.h:589-625); an hfov <= 0 other than -1 is refused"""
    j = demo(["--photo", "p.tif", "--output", "o.tif"], {"p.tif": (300, 200, 3)})
    assert j["ok"]
    f = j["facets"][0]
    assert f["projection"] == ea.RECTILINEAR and f["hfov"] == 65.0 * RAD and (f["yaw"], f["pitch"], f["roll"]) == (0, 0, 0)
    j = demo(["--facet", "p.tif", "metadata", "40", "10", "0", "0", "--output", "o.tif"], {"p.tif": (300, 200, 3)})
    assert j["ok"] and j["facets"][0]["projection"] == ea.RECTILINEAR and j["facets"][0]["hfov"] == 40.0 * RAD
    j = demo(["--facet", "p.tif", "spherical", "0", "0", "0", "0", "--output", "o.tif"], {"p.tif": (300, 200, 3)})
    assert j["ok"] is False and "hfov invalid" in j["error"]
