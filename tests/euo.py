"""ctypes bindings of the CPU oracle (oracle/eu_oracle.c). TEST INFRASTRUCTURE."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_build", "libeu_oracle.so")

SPHERICAL, CYLINDRICAL, RECTILINEAR, STEREOGRAPHIC, FISHEYE, CUBEMAP, BIATAN6 = range(7)
MIRROR, PERIODIC, REFLECT, NATURAL, CONSTANT, ZEROPAD, GUESS = range(7)
PRJ_NAMES = ["spherical", "cylindrical", "rectilinear", "stereographic",
             "fisheye", "cubemap", "biatan6"]


class Spline(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_float)),
                ("shape", C.c_long * 2), ("stride", C.c_long * 2),
                ("left", C.c_long * 2), ("right", C.c_long * 2),
                ("core", C.c_long * 2), ("bc", C.c_int * 2),
                ("degree", C.c_int), ("nch", C.c_int)]


class Source(C.Structure):
    _fields_ = [("projection", C.c_int), ("hfov", C.c_double),
                ("width", C.c_int), ("height", C.c_int),
                ("window_width", C.c_int), ("window_height", C.c_int),
                ("window_x_offset", C.c_int), ("window_y_offset", C.c_int),
                ("yaw", C.c_double), ("pitch", C.c_double), ("roll", C.c_double),
                ("brighten", C.c_double), ("step", C.c_double),
                ("has_lcp", C.c_int),
                ("a", C.c_double), ("b", C.c_double), ("c", C.c_double),
                ("h", C.c_double), ("v", C.c_double), ("s", C.c_double),
                ("shear_g", C.c_double), ("shear_t", C.c_double),
                ("tr_x", C.c_double), ("tr_y", C.c_double), ("tr_z", C.c_double),
                ("tp_y", C.c_double), ("tp_p", C.c_double), ("tp_r", C.c_double),
                ("spl", Spline),
                ("refc_md", C.c_float), ("model_to_px", C.c_float),
                ("section_px", C.c_int), ("mask_paint", C.c_int)]


class Job(C.Structure):
    _fields_ = [("projection", C.c_int), ("width", C.c_int), ("height", C.c_int),
                ("x0", C.c_double), ("x1", C.c_double),
                ("y0", C.c_double), ("y1", C.c_double),
                ("yaw", C.c_double), ("pitch", C.c_double), ("roll", C.c_double),
                ("nch", C.c_int), ("ntaps", C.c_int),
                ("taps", C.POINTER(C.c_float)),
                ("row_begin", C.c_int), ("row_end", C.c_int),
                ("stage", C.c_int), ("nthreads", C.c_int),
                ("crop_x0", C.c_int), ("crop_y0", C.c_int), ("crop_w", C.c_int), ("crop_h", C.c_int),
                ("screen", C.c_int), ("synopsis", C.c_int), ("single", C.POINTER(Source))]


class Metrics(C.Structure):
    _fields_ = [("face_px", C.c_long), ("section_px", C.c_long),
                ("left_frame_px", C.c_long), ("right_frame_px", C.c_long),
                ("n_tiles", C.c_long), ("inherent_support_px", C.c_long),
                ("model_to_px", C.c_double), ("px_to_model", C.c_double),
                ("section_md", C.c_double), ("refc_md", C.c_double),
                ("radius_md", C.c_double), ("discrete90", C.c_int)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"),
                           "_build/libeu_oracle.so"])


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ROOT, "oracle", "eu_oracle.c")
        if (not os.path.exists(LIB)
                or os.path.getmtime(LIB) < os.path.getmtime(src)):
            build()
        _lib = C.CDLL(LIB)
        _lib.euo_get_vfov.restype = C.c_double
        _lib.euo_get_step.restype = C.c_double
        _lib.euo_get_vfov.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double]
        _lib.euo_get_step.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double]
        _lib.euo_get_extent.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
        _lib.euo_make_r3.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p]
        _lib.euo_make_spread.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float,
                                         C.c_float, C.c_void_p, C.c_int]
        _lib.euo_basis_weights.argtypes = [C.c_int, C.c_float, C.c_void_p]
        _lib.euo_spline_init.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long,
                                         C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.euo_spline_geometry.argtypes = [C.c_int, C.c_int, C.c_int, C.c_long,
                                             C.c_long, C.c_void_p]
        _lib.euo_filter_lines.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long,
                                          C.c_long, C.c_int, C.c_int, C.c_double]
        _lib.euo_metrics_init.argtypes = [C.c_void_p, C.c_long, C.c_double, C.c_long, C.c_long]
        _lib.euo_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
        _lib.euo_eval_shifted.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_void_p]
        _lib.euo_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_long]
    return _lib


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def get_extent(prj, w, h, hfov):
    e = np.zeros(4, np.float64)
    lib().euo_get_extent(prj, w, h, hfov, ptr(e))
    return e


def make_r3(roll, pitch, yaw, inverse=False):
    m = np.zeros(9, np.float64)
    lib().euo_make_r3(roll, pitch, yaw, int(inverse), ptr(m))
    return m.reshape(3, 3)


def make_spread(w, h=0, d=1.0, sigma=0.0, threshold=0.0):
    out = np.zeros((max(w, 2) * max(h if h > 0 else w, 2), 3), np.float32)
    n = lib().euo_make_spread(w, h, d, sigma, threshold, ptr(out), out.shape[0])
    assert n >= 0
    return out[:n].copy()


def basis_weights(degree, delta):
    w = np.zeros(degree + 1, np.float32)
    lib().euo_basis_weights(degree, delta, ptr(w))
    return w


def weight_matrix(degree):
    m = np.zeros((degree + 1, degree + 1), np.float32)
    lib().euo_weight_matrix(degree, ptr(m))
    return m


def poles(degree):
    p = np.zeros(8, np.longdouble)
    n = lib().euo_poles(degree, ptr(p))
    return p[:n]


class BSpline:
    """A braced coefficient container owned by numpy, described by a Spline."""

    def __init__(self, core, degree, bc0, bc1):
        core = np.ascontiguousarray(core, np.float32)
        h, w, nch = core.shape
        g = (C.c_long * 6)()
        lib().euo_spline_geometry(degree, bc0, bc1, w, h, g)
        self.container = np.zeros((g[1], g[0], nch), np.float32)
        self.s = Spline()
        lib().euo_spline_init(C.byref(self.s), ptr(self.container), w, h, nch,
                              degree, bc0, bc1)
        lib().euo_spline_set_core(C.byref(self.s), ptr(core))
        self.w, self.h, self.nch, self.degree = w, h, nch, degree

    def brace(self, axis=-1):
        lib().euo_brace(C.byref(self.s), axis)

    def prefilter(self, degree):
        lib().euo_prefilter(C.byref(self.s), degree)

    def spherical_prefilter(self, degree):
        lib().euo_spherical_prefilter(C.byref(self.s), degree)

    def eval(self, crd):
        crd = np.ascontiguousarray(crd, np.float32)
        out = np.zeros((crd.shape[0], self.nch), np.float32)
        lib().euo_eval(C.byref(self.s), ptr(crd), crd.shape[0], ptr(out))
        return out


def metrics(face_px, face_fov=np.pi / 2, support_min=8, tile=64):
    m = Metrics()
    lib().euo_metrics_init(C.byref(m), face_px, face_fov, support_min, tile)
    return m


def cubemap_build(faces, spline_degree, prefilter_degree, face_fov=np.pi / 2,
                  support_min=8, tile=64):
    """faces: (6*F, F, nch) stacked cube faces -> (metrics, IR image)"""
    faces = np.ascontiguousarray(faces, np.float32)
    f = faces.shape[1]
    nch = faces.shape[2]
    m = metrics(f, face_fov, support_min, tile)
    ir = np.zeros((6 * m.section_px, m.section_px, nch), np.float32)
    lib().euo_cubemap_build(C.byref(m), ptr(faces), nch, spline_degree,
                            prefilter_degree, ptr(ir))
    return m, ir


def lens_factor(a, b, c, x):
    """the oracle's lcp<float>::eval (oracle/eu_oracle.c: lens_factor)"""
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros_like(x)
    f = lib().euo_lens_factor
    f.restype = None
    f.argtypes = [C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_long, C.c_void_p]
    f(a, b, c, x.ctypes.data, len(x), out.ctypes.data)
    return out


def inverse_lcp(a, b, c, r_max, sz, x):
    """the oracle's inverse_lcp<float, LANES>(a, b, c, r_max, sz).eval (oracle/eu_oracle.c: inv_lcp_*);
    returns (factors, prefiltered knots)"""
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros_like(x)
    knots = np.zeros(136, np.float32)
    f = lib().euo_inverse_lcp
    f.restype = C.c_int
    f.argtypes = [C.c_double] * 4 + [C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_int]
    nk = f(a, b, c, r_max, sz, x.ctypes.data, len(x), out.ctypes.data, knots.ctypes.data, len(knots))
    assert nk > 0
    return out, knots[:nk]


def source_coordinates(src, rays):
    """mount_t::get_coordinate / cubemap pickup for caller-supplied rays: (n, 3) -> (n, 3) of
    {source x, source y, cube face | 0}, {0, 0, -1} for a miss"""
    rays = np.ascontiguousarray(rays, np.float32)
    out = np.zeros_like(rays)
    f = lib().euo_source_coordinates
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
    f(C.byref(src), rays.ctypes.data, len(rays), out.ctypes.data)
    return out


def facet_alpha(w, h, polygons=(), crop=None, crop_kind=0, stage=1):
    """the oracle's alpha plane of a masked / cropped facet (oracle/eu_oracle.c: euo_facet_alpha);
    stage 0 stops before the binomial"""
    counts = np.array([len(x) for x, _ in polygons] or [0], np.int32)
    xs = np.ascontiguousarray(np.concatenate([np.asarray(x, np.float32) for x, _ in polygons] or [np.zeros(1, np.float32)]))
    ys = np.ascontiguousarray(np.concatenate([np.asarray(y, np.float32) for _, y in polygons] or [np.zeros(1, np.float32)]))
    alpha = np.zeros((h, w), np.float32)
    c = crop if crop is not None else (0, 0, 0, 0)
    f = lib().euo_facet_alpha
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6
    f(alpha.ctypes.data, w, h, len(polygons), counts.ctypes.data, xs.ctypes.data, ys.ctypes.data,
      crop_kind if crop is not None else 0, c[0], c[1], c[2], c[3], stage)
    return alpha


def binomial_plane(plane):
    """zimt::convolve with 1 4 6 4 1 / 16, REFLECT, both axes, as the oracle restates it"""
    out = np.ascontiguousarray(plane, np.float32).copy()
    f = lib().euo_binomial_plane
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_int]
    f(out.ctypes.data, out.shape[1], out.shape[0])
    return out
