"""The stand-alone command line (tools/envutil_hip.cc -> envutil_amd/bin/envutil_hip): envutil's
options and PTO handling (include/eu_frontend.hpp), PFM / PNM / PAM image files
(include/eu_image_io.hpp), PTO masks and lens crops applied to the loaded pixels
(include/eu_imageprep.hpp), payload() on the HIP library. CPU tests: the program builds, reports
errors instead of asserting, and fails loudly without a device. GPU tests: the files it writes hold
the same bits as the same jobs rendered through the Python binding."""
import os
import subprocess

import numpy as np
import pytest

import envutil_amd as ea

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "envutil_amd", "bin", "envutil_hip")


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(EXE) or not os.path.exists(ea.lib_path()):
        ea.build()

    def run(argv, cwd, stdin=None):
        return subprocess.run([EXE] + list(argv), capture_output=True, text=True, cwd=str(cwd), input=stdin, timeout=600)
    return run


def write_pfm(path, img):
    h, w = img.shape[:2]
    n = 1 if img.ndim == 2 else img.shape[2]
    magic = {1: b"Pf", 3: b"PF", 4: b"PF4"}[n]
    with open(path, "wb") as f:
        f.write(magic + b"\n%d %d\n-1.0\n" % (w, h))
        f.write(np.ascontiguousarray(img[::-1], "<f4").tobytes())


def read_pfm(path):
    with open(path, "rb") as f:
        magic = f.readline().strip()
        w, h = map(int, f.readline().split())
        scale = float(f.readline())
        n = {b"Pf": 1, b"PF": 3, b"PF4": 4}[magic]
        a = np.frombuffer(f.read(), "<f4" if scale < 0 else ">f4").reshape(h, w, n)
    return np.ascontiguousarray(a[::-1]).astype(np.float32)


def synth(w, h, n, seed=0):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([0.5 + 0.3 * np.sin(x / (5.0 + c)) * np.cos(y / (7.0 + c)) for c in range(n)], 2)
    return (img + 0.05 * rng.random((h, w, n))).astype(np.float32)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# ---------------------------------------------------------------------------------- CPU

def test_errors_are_messages(cli, tmp_path):
    write_pfm(tmp_path / "a.pfm", synth(32, 16, 3))
    r = cli(["--facet", "nope.pfm", "spherical", "360", "0", "0", "0", "--output", "o.pfm"], tmp_path)
    assert r.returncode == 2 and "failed to open facet image" in r.stderr
    r = cli(["--facet", "a.pfm", "spherical", "360", "0", "0", "0", "--bogus", "1", "--output", "o.pfm"], tmp_path)
    assert r.returncode == 2 and "unknown option" in r.stderr
    (tmp_path / "junk.pfm").write_bytes(b"GIF89a....")
    r = cli(["--facet", "junk.pfm", "spherical", "360", "0", "0", "0", "--output", "o.pfm"], tmp_path)
    assert r.returncode == 2
    assert cli([], tmp_path).returncode == 2


def test_no_device_is_a_loud_failure(cli, tmp_path):
    if ea.device_count() > 0:
        pytest.skip("a HIP device is present")
    write_pfm(tmp_path / "a.pfm", synth(32, 16, 3))
    r = cli(["--facet", "a.pfm", "spherical", "360", "0", "0", "0", "--projection", "cubemap", "--hfov", "90",
             "--width", "8", "--output", "o.pfm"], tmp_path)
    assert r.returncode == 1 and "no HIP device" in r.stderr and not (tmp_path / "o.pfm").exists()


# ---------------------------------------------------------------------------------- GPU

@pytest.mark.gpu
def test_latlon_to_cubemap_files(cli, tmp_path):
    img = synth(256, 128, 3)
    write_pfm(tmp_path / "pano.pfm", img)
    r = cli(["--facet", "pano.pfm", "spherical", "360", "0", "0", "0", "--projection", "cubemap", "--hfov", "90",
             "--width", "64", "--degree", "3", "--twine", "0", "--output", "cube.pfm", "-v"], tmp_path)
    assert r.returncode == 0, r.stderr
    src = ea.Source.load(ea.facet_spec(ea.SPHERICAL, 256, 128, 360.0), img, 3)
    want = ea.render(ea.arguments(ea.CUBEMAP, 64, 384, 90.0, spline_degree=3), src)
    assert (bits(read_pfm(tmp_path / "cube.pfm")) == bits(want)).all()
    # six face files by a format string, 16-bit PAM: the same pixels quantised as OpenImageIO does
    r = cli(["--facet", "pano.pfm", "spherical", "360", "0", "0", "0", "--projection", "cubemap", "--hfov", "90",
             "--width", "64", "--degree", "3", "--twine", "0", "--output", "face_%s.pam"], tmp_path)
    assert r.returncode == 0, r.stderr
    for i, name in enumerate(["left", "right", "top", "bottom", "front", "back"]):
        raw = (tmp_path / f"face_{name}.pam").read_bytes()
        head, _, data = raw.partition(b"ENDHDR\n")
        assert b"WIDTH 64" in head and b"DEPTH 3" in head and b"MAXVAL 65535" in head
        q = np.frombuffer(data, ">u2").reshape(64, 64, 3)
        face = np.clip(want[64 * i:64 * (i + 1)], 0, 1)
        assert (q == (face * np.float32(65535) + np.float32(0.5)).astype(np.uint32)).all()
    # ... and the six faces read back as a cubemap source
    # (the faces were written in the working space: told so, they are read without a conversion)
    r = cli(["--facet", "face_%s.pam", "cubemap", "90", "0", "0", "0", "--projection", "spherical", "--hfov", "360",
             "--width", "128", "--degree", "1", "--twine", "0", "--input_colour_space", "Linear", "--output", "back.pfm"], tmp_path)
    assert r.returncode == 0, r.stderr
    faces = np.concatenate([np.frombuffer((tmp_path / f"face_{n}.pam").read_bytes().partition(b"ENDHDR\n")[2], ">u2")
                            .reshape(64, 64, 3) for n in ["left", "right", "top", "bottom", "front", "back"]])
    fimg = (faces.astype(np.float32) / np.float32(65535)).astype(np.float32)
    csrc = ea.Source.load(ea.facet_spec(ea.CUBEMAP, 64, 384, 90.0), fimg, 1)
    want2 = ea.render(ea.arguments(ea.SPHERICAL, 128, 64, 360.0, spline_degree=1), csrc)
    assert (bits(read_pfm(tmp_path / "back.pfm")) == bits(want2)).all()


@pytest.mark.gpu
def test_pto_with_mask_and_crop(cli, tmp_path):
    """a PTO with an exclude mask on one image and a lens crop on a fisheye image: both facets gain an
    alpha channel, the masked regions are transparent in the output"""
    a, b = synth(200, 150, 3, 1), synth(160, 160, 3, 2)
    write_pfm(tmp_path / "a.pfm", a)
    write_pfm(tmp_path / "b.pfm", b)
    (tmp_path / "two.pto").write_text(
        'p f2 w300 h150 v360 n"TIFF"\n'
        'i w200 h150 f0 v70 y10 p5 r2 n"a.pfm"\n'
        'i w160 h160 f3 v170 y-100 p-20 r0 S10,150,10,150 n"b.pfm"\n'
        'k i0 t0 p"30 20 120 25 140 110 40 100"\n')
    r = cli(["--pto", "two.pto", "--output", "o.pfm", "--degree", "1", "--twine", "0"], tmp_path)
    assert r.returncode == 0, r.stderr
    out = read_pfm(tmp_path / "o.pfm")
    assert out.shape == (150, 300, 4)
    # the same job through the binding, the pixels prepared with the library's host function
    pa = np.concatenate([a, np.ones((150, 200, 1), np.float32)], 2)
    pb = np.concatenate([b, np.ones((160, 160, 1), np.float32)], 2)
    ea.facet_alpha(pa, [(np.array([30, 120, 140, 40], np.float32), np.array([20, 25, 110, 100], np.float32))])
    ea.facet_alpha(pb, crop=(10, 150, 10, 150), crop_kind=2)
    fa = ea.facet_spec(ea.RECTILINEAR, 200, 150, 70.0, nchannels=4, yaw=10, pitch=5, roll=2)
    fb = ea.facet_spec(ea.FISHEYE, 160, 160, 170.0, nchannels=4, yaw=-100, pitch=-20, roll=0)
    sa, sb = ea.Source.load(fa, pa, 1), ea.Source.load(fb, pb, 1)
    want = ea.render(ea.arguments(ea.SPHERICAL, 300, 150, 360.0, spline_degree=1), [sa, sb], 4)
    assert (bits(out) == bits(want)).all()
    assert (out[..., 3] == 0).any() and (out[..., 3] == 1).any()
    # --mask_for 1: where the fisheye image shows in that stitch (white * alpha), the other facet black
    r = cli(["--pto", "two.pto", "--output", "m1.pfm", "--degree", "1", "--twine", "0", "--mask_for", "1"], tmp_path)
    assert r.returncode == 0, r.stderr
    fa.masked, fb.masked = 0, 1
    sa.update_facet(fa)
    sb.update_facet(fb)
    wantm = ea.render(ea.arguments(ea.SPHERICAL, 300, 150, 360.0, spline_degree=1), [sa, sb], 4)
    got = read_pfm(tmp_path / "m1.pfm")
    assert (bits(got) == bits(wantm)).all()
    assert (got[..., 0] == got[..., 1]).all() and got[..., 0].max() == 1.0


@pytest.mark.gpu
def test_pipe_mode_keeps_assets_resident(cli, tmp_path):
    img = synth(128, 64, 3, 5)
    write_pfm(tmp_path / "pano.pfm", img)
    jobs = "--yaw 0 --output v0.pfm\n--yaw 45 --output 'v 45.pfm'\n"
    r = cli(["-v", "--facet", "pano.pfm", "spherical", "360", "0", "0", "0", "--projection", "rectilinear", "--hfov", "80",
             "--width", "96", "--height", "64", "--degree", "2", "--twine", "0", "-"], tmp_path, stdin=jobs)
    assert r.returncode == 0, r.stderr
    assert "already resident" in r.stdout
    src = ea.Source.load(ea.facet_spec(ea.SPHERICAL, 128, 64, 360.0), img, 2)
    for yaw, name in ((0, "v0.pfm"), (45, "v 45.pfm")):
        want = ea.render(ea.arguments(ea.RECTILINEAR, 96, 64, 80.0, yaw=yaw, spline_degree=2), src)
        assert (bits(read_pfm(tmp_path / name)) == bits(want)).all(), name


def srgb_to_linear(v):
    v = v.astype(np.float32)
    return np.where(v <= np.float32(0.04045), v / np.float32(12.92),
                    np.power((v + np.float32(0.055)) / np.float32(1.055), np.float32(2.4))).astype(np.float32)


def linear_to_srgb(v):
    v = v.astype(np.float32)
    return np.where(v <= np.float32(0.0031308), np.float32(12.92) * v,
                    np.float32(1.055) * np.power(np.maximum(v, 0), np.float32(1.0 / 2.4)) - np.float32(0.055)).astype(np.float32)


@pytest.mark.gpu
def test_eight_bit_input(cli, tmp_path):
    """integer formats are display-referred (OpenImageIO labels them; the reference converts every image whose
    colour space differs from the working one, envutil_basic.h:950-977): a P6 file is read as sRGB and taken to the
    working space (Linear); the output stays in the working space unless --output_colour_space says otherwise
    (:786-812). --input_colour_space overrides the file's label. Unknown names are an error, not silence."""
    rng = np.random.default_rng(9)
    q = rng.integers(0, 256, (40, 80, 3), dtype=np.uint8)
    (tmp_path / "in.ppm").write_bytes(b"P6\n# a comment\n80 40\n255\n" + q.tobytes())
    base = ["--facet", "in.ppm", "spherical", "360", "0", "0", "0", "--projection", "spherical", "--hfov", "360",
            "--width", "80", "--degree", "1", "--twine", "0"]

    def run(extra, name):
        r = cli(base + extra + ["--output", name], tmp_path)
        assert r.returncode == 0, r.stderr
        raw = (tmp_path / name).read_bytes()
        # the target's projection and hfov travel as comment lines, as envutil attaches them to its output
        assert raw.startswith(b"P6\n# Projection: spherical\n# Hfov: 360\n80 40\n255\n")
        return np.frombuffer(raw.split(b"80 40\n255\n", 1)[1], np.uint8).reshape(40, 80, 3).astype(np.int32)

    def expect(img, back=None):
        src = ea.Source.load(ea.facet_spec(ea.SPHERICAL, 80, 40, 360.0), img, 1)
        want = ea.render(ea.arguments(ea.SPHERICAL, 80, 40, 360.0, spline_degree=1), src)
        if back is not None:
            want = back(want)
        return (np.clip(want, 0, 1) * np.float32(255) + np.float32(0.5)).astype(np.int32)

    raw = (q.astype(np.float32) / np.float32(255)).astype(np.float32)
    # the file's own label: sRGB -> Linear on the way in (numpy's powf against the C library's: one count of slack)
    assert np.abs(run([], "out.ppm") - expect(srgb_to_linear(raw))).max() <= 1
    # told that the samples are linear: no conversion at all - exact
    assert (run(["--input_colour_space", "Linear"], "lin.ppm") == expect(raw)).all()
    # and back to sRGB on the way out
    assert np.abs(run(["--output_colour_space", "sRGB"], "srgb.ppm") - expect(srgb_to_linear(raw), linear_to_srgb)).max() <= 1
    # a colour space this build does not know is refused with a message
    r = cli(base + ["--input_colour_space", "ACEScg", "--output", "x.ppm"], tmp_path)
    assert r.returncode != 0 and "ACEScg" in r.stderr and "not known" in r.stderr


@pytest.mark.gpu
def test_split_recreates_every_facet(cli, tmp_path):
    """--split (core(), envutil_main.cc:1676-1722): one --single job per facet, each facet's geometry
    taken over as the target and the stitch of all facets rendered into it"""
    a, b = synth(120, 90, 3, 3), synth(100, 100, 3, 4)
    write_pfm(tmp_path / "a.pfm", a)
    write_pfm(tmp_path / "b.pfm", b)
    (tmp_path / "two.pto").write_text(
        'p f2 w200 h100 v360 n"TIFF"\n'
        'i w120 h90 f0 v60 y5 p2 r1 a0.01 b-0.02 c0.01 n"a.pfm"\n'
        'i w100 h100 f0 v75 y40 p-5 r0 n"b.pfm"\n')
    r = cli(["--pto", "two.pto", "--split", "facet_%02d.pfm", "--degree", "1", "--twine", "0"], tmp_path)
    assert r.returncode == 0, r.stderr
    lens = dict(a=0.01, b=-0.02, c=0.01)
    fa = ea.facet_spec(ea.RECTILINEAR, 120, 90, 60.0, yaw=5, pitch=2, roll=1, lens=lens)
    fb = ea.facet_spec(ea.RECTILINEAR, 100, 100, 75.0, yaw=40, pitch=-5, roll=0)
    srcs = [ea.Source.load(fa, a, 1), ea.Source.load(fb, b, 1)]
    for i, f in enumerate((fa, fb)):
        want = ea.render(ea.arguments.for_single(f, spline_degree=1), srcs, 3)
        got = read_pfm(tmp_path / f"facet_{i:02d}.pfm")
        assert got.shape == want.shape and (bits(got) == bits(want)).all(), i


@pytest.mark.gpu
def test_pto_crop_window_output(cli, tmp_path):
    """a p-line with a crop (S clause): the output file holds the crop window only (store_cropped)"""
    img = synth(160, 80, 3, 6)
    write_pfm(tmp_path / "pano.pfm", img)
    (tmp_path / "c.pto").write_text(
        'p f2 w240 h120 v360 S40,200,10,90 n"TIFF"\n'
        'i w160 h80 f4 v360 y20 p0 r0 n"pano.pfm"\n')
    r = cli(["--pto", "c.pto", "--output", "crop.pfm", "--degree", "3", "--twine", "0"], tmp_path)
    assert r.returncode == 0, r.stderr
    src = ea.Source.load(ea.facet_spec(ea.SPHERICAL, 160, 80, 360.0, yaw=20), img, 3)
    want = ea.render(ea.arguments(ea.SPHERICAL, 240, 120, 360.0, spline_degree=3, crop=(40, 200, 10, 90)), src)
    got = read_pfm(tmp_path / "crop.pfm")
    assert got.shape == (80, 160, 3) and (bits(got) == bits(want)).all()


# ---------------------------------------------------------------------------------- image files (CPU)

def rgbe_decode(q):
    """(h, w, 4) uint8 -> float32, mantissa * 2^(e - 136), zero exponent -> 0 (rgbe.c)"""
    e = q[..., 3].astype(np.int32)
    f = np.ldexp(np.float32(1.0), e - 136).astype(np.float32)
    out = (q[..., :3].astype(np.float32) * f[..., None]).astype(np.float32)
    out[e == 0] = 0
    return out


def rgbe_encode(img):
    v = img.max(axis=2)
    m, e = np.frexp(v.astype(np.float32))
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = (m.astype(np.float32) * np.float32(256.0) / v).astype(np.float32)
    q = np.zeros(img.shape[:2] + (4,), np.uint8)
    ok = v >= np.float32(1e-32)
    for c in range(3):
        q[..., c] = np.where(ok, np.maximum(img[..., c], 0) * scale, 0).astype(np.uint8)
    q[..., 3] = np.where(ok, e + 128, 0).astype(np.uint8)
    return q


def rle_scanline(row):
    """new-style Radiance run-length encoding of one (w, 4) uint8 scanline"""
    w = row.shape[0]
    out = bytearray([2, 2, w >> 8, w & 255])
    for c in range(4):
        comp = row[:, c].tolist()
        x = 0
        while x < w:
            run = 1
            while x + run < w and run < 127 and comp[x + run] == comp[x]:
                run += 1
            if run >= 4:
                out += bytes([128 + run, comp[x]])
                x += run
            else:
                n = 1
                while x + n < w and n < 128 and not (x + n + 3 < w and comp[x + n] == comp[x + n + 1] == comp[x + n + 2] == comp[x + n + 3]):
                    n += 1
                out += bytes([n]) + bytes(comp[x:x + n])
                x += n
    return bytes(out)


@pytest.fixture(scope="module")
def io_demo():
    exe = os.path.join(ROOT, "envutil_amd", "build", "io_demo")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "csrc", "io_demo.cc"), "-o", exe])

    def run(src, dst, cwd):
        return subprocess.run([exe, src, dst], capture_output=True, text=True, cwd=str(cwd), timeout=60)
    return run


def test_radiance_pictures(io_demo, tmp_path):
    rng = np.random.default_rng(4)
    w, h = 40, 12
    q = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    q[..., 3] = rng.integers(100, 150, (h, w))
    q[2, 5:30] = q[2, 5]                       # runs for the encoder
    q[3, :, 3] = 0                             # zero exponents: black
    head = b"#?RADIANCE\n# made by a test\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y %d +X %d\n" % (h, w)
    (tmp_path / "flat.hdr").write_bytes(head + q.tobytes())
    (tmp_path / "rle.hdr").write_bytes(head + b"".join(rle_scanline(q[y]) for y in range(h)))
    want = rgbe_decode(q)
    for name in ("flat.hdr", "rle.hdr"):
        r = io_demo(name, "out.pfm", tmp_path)
        assert r.returncode == 0 and r.stdout.split() == [str(w), str(h), "3"], r.stderr
        assert (bits(read_pfm(tmp_path / "out.pfm")) == bits(want)).all(), name
    # writing: float -> RGBE as rgbe.c does it, flat scanlines
    img = (rng.random((h, w, 3), dtype=np.float32) * np.float32(8.0)).astype(np.float32)
    img[0, 0] = 0
    img[0, 1] = (1e-35, 0, 0)
    write_pfm(tmp_path / "in.pfm", img)
    r = io_demo("in.pfm", "out.hdr", tmp_path)
    assert r.returncode == 0, r.stderr
    raw = (tmp_path / "out.hdr").read_bytes()
    body = raw.split(b"-Y %d +X %d\n" % (h, w), 1)[1]
    assert raw.startswith(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n")
    assert (np.frombuffer(body, np.uint8).reshape(h, w, 4) == rgbe_encode(img)).all()
    # a truncated picture and a foreign orientation are errors, not crashes
    (tmp_path / "cut.hdr").write_bytes(head + q.tobytes()[:100])
    assert io_demo("cut.hdr", "x.pfm", tmp_path).returncode == 1
    (tmp_path / "rot.hdr").write_bytes(head.replace(b"-Y", b"+Y") + q.tobytes())
    assert io_demo("rot.hdr", "x.pfm", tmp_path).returncode == 1


def test_integer_and_float_files_round_trip(io_demo, tmp_path):
    rng = np.random.default_rng(8)
    for n, name in ((1, "g.pfm"), (3, "c.pfm"), (4, "a.pfm")):
        img = rng.random((9, 14, n), dtype=np.float32)
        write_pfm(tmp_path / name, img if n > 1 else img[..., 0])
        r = io_demo(name, "o_" + name, tmp_path)
        assert r.returncode == 0 and r.stdout.split() == ["14", "9", str(n)], r.stderr
        assert (bits(read_pfm(tmp_path / ("o_" + name))) == bits(img)).all()
    # big-endian PFM in, 16-bit PAM out and back
    img = rng.random((5, 7, 4), dtype=np.float32)
    (tmp_path / "be.pfm").write_bytes(b"PF4\n7 5\n1.0\n" + np.ascontiguousarray(img[::-1], ">f4").tobytes())
    assert io_demo("be.pfm", "q.pam", tmp_path).returncode == 0
    assert io_demo("q.pam", "back.pfm", tmp_path).returncode == 0
    q16 = (np.clip(img, 0, 1) * np.float32(65535) + np.float32(0.5)).astype(np.uint32)
    assert (bits(read_pfm(tmp_path / "back.pfm")) == bits(q16.astype(np.float32) / np.float32(65535))).all()
    # 2-channel images need PAM; PFM refuses them
    write2 = tmp_path / "ga.pam"
    write2.write_bytes(b"P7\nWIDTH 3\nHEIGHT 2\nDEPTH 2\nMAXVAL 255\nTUPLTYPE GRAYSCALE_ALPHA\nENDHDR\n" + bytes(range(12)))
    assert io_demo("ga.pam", "ga2.pam", tmp_path).returncode == 0
    assert io_demo("ga.pam", "ga.pfm", tmp_path).returncode == 1


@pytest.mark.gpu
def test_radiance_environment_map(cli, tmp_path):
    """the usual case: a lat/lon .hdr environment map to a cubemap, written as .hdr again"""
    rng = np.random.default_rng(12)
    w, h = 128, 64
    q = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    q[..., 3] = rng.integers(120, 136, (h, w))
    head = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w)
    (tmp_path / "env.hdr").write_bytes(head + b"".join(rle_scanline(q[y]) for y in range(h)))
    r = cli(["--facet", "env.hdr", "spherical", "360", "0", "0", "0", "--projection", "cubemap", "--hfov", "90",
             "--width", "32", "--degree", "3", "--twine", "0", "--output", "cube.hdr"], tmp_path)
    assert r.returncode == 0, r.stderr
    img = rgbe_decode(q)
    src = ea.Source.load(ea.facet_spec(ea.SPHERICAL, w, h, 360.0), img, 3)
    want = ea.render(ea.arguments(ea.CUBEMAP, 32, 192, 90.0, spline_degree=3), src)
    raw = (tmp_path / "cube.hdr").read_bytes()
    body = raw.split(b"-Y 192 +X 32\n", 1)[1]
    assert (np.frombuffer(body, np.uint8).reshape(192, 32, 4) == rgbe_encode(want)).all()


@pytest.mark.gpu
def test_photo_reads_the_metadata_envutil_writes(cli, tmp_path):
    """--photo IMAGE = --facet IMAGE metadata -1 0 0 0 (envutil_main.cc:916-927): projection and hfov come from the
    image's "Projection" / "Hfov" metadata - which the program attaches to what it writes - or default to
    rectilinear, 65 degrees"""
    img = synth(256, 128, 3, 21)
    write_pfm(tmp_path / "pano.pfm", img)
    r = cli(["--facet", "pano.pfm", "spherical", "360", "0", "0", "0", "--projection", "fisheye", "--hfov", "140",
             "--width", "120", "--height", "120", "--degree", "1", "--twine", "0", "--output", "view.hdr"], tmp_path)
    assert r.returncode == 0, r.stderr
    raw = (tmp_path / "view.hdr").read_bytes()
    assert b"FORMAT=32-bit_rle_rgbe\nProjection=fisheye\nHfov=140\n\n-Y 120 +X 120\n" in raw[:200]
    r = cli(["--photo", "view.hdr", "--projection", "spherical", "--hfov", "360", "--width", "200", "--degree", "1",
             "--twine", "0", "--output", "back.pfm"], tmp_path)
    assert r.returncode == 0, r.stderr
    body = raw.split(b"-Y 120 +X 120\n", 1)[1]
    view = rgbe_decode(np.frombuffer(body, np.uint8).reshape(120, 120, 4))
    src = ea.Source.load(ea.facet_spec(ea.FISHEYE, 120, 120, 140.0), view, 1)
    want = ea.render(ea.arguments(ea.SPHERICAL, 200, 100, 360.0, spline_degree=1), src)
    assert (bits(read_pfm(tmp_path / "back.pfm")) == bits(want)).all()
    # a file without metadata: rectilinear, 65 degrees
    r = cli(["--photo", "pano.pfm", "--projection", "spherical", "--hfov", "360", "--width", "200", "--degree", "1",
             "--twine", "0", "--output", "p65.pfm"], tmp_path)
    assert r.returncode == 0, r.stderr
    src65 = ea.Source.load(ea.facet_spec(ea.RECTILINEAR, 256, 128, 65.0), img, 1)
    want65 = ea.render(ea.arguments(ea.SPHERICAL, 200, 100, 360.0, spline_degree=1), src65)
    assert (bits(read_pfm(tmp_path / "p65.pfm")) == bits(want65)).all()
