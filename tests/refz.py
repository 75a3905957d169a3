"""ctypes bindings of oracle/_ref/libref_zimt.so - the reference's own zimt
headers compiled in place from /root/reference. Only present in the build
container; tests that need it are marked 'ref' and skip elsewhere."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libref_zimt.so")


def available():
    return os.path.exists(LIB)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB)
        _lib.ref_bspline_new.restype = C.c_void_p
        _lib.ref_bspline_new.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_int,
                                         C.c_int, C.c_int, C.c_int]
        for n in ("free", "prefilter", "spherical", "brace", "geometry",
                  "container", "eval", "process_affine"):
            getattr(_lib, "ref_bspline_" + n).restype = None
        _lib.ref_bspline_free.argtypes = [C.c_void_p]
        _lib.ref_bspline_prefilter.argtypes = [C.c_void_p, C.c_int]
        _lib.ref_bspline_spherical.argtypes = [C.c_void_p, C.c_int]
        _lib.ref_bspline_brace.argtypes = [C.c_void_p, C.c_int]
        _lib.ref_bspline_geometry.argtypes = [C.c_void_p, C.c_void_p]
        _lib.ref_bspline_container.argtypes = [C.c_void_p, C.c_void_p]
        _lib.ref_bspline_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
        _lib.ref_bspline_process_affine.argtypes = [C.c_void_p, C.c_long, C.c_long,
                                                    C.c_void_p, C.c_void_p]
        _lib.ref_basis_weights.argtypes = [C.c_int, C.c_float, C.c_void_p]
        _lib.ref_filter_2d.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_int,
                                       C.c_int, C.c_int, C.c_int]
    return _lib


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class RefSpline:
    def __init__(self, core, degree, bc0, bc1):
        core = np.ascontiguousarray(core, np.float32)
        h, w, nch = core.shape
        self.h_ = lib().ref_bspline_new(ptr(core), w, h, nch, degree, bc0, bc1)
        self.nch = nch

    def __del__(self):
        if getattr(self, "h_", None):
            lib().ref_bspline_free(self.h_)
            self.h_ = None

    def geometry(self):
        g = (C.c_long * 10)()
        lib().ref_bspline_geometry(self.h_, g)
        return list(g)

    def prefilter(self, degree):
        lib().ref_bspline_prefilter(self.h_, degree)

    def spherical(self, degree):
        lib().ref_bspline_spherical(self.h_, degree)

    def brace(self, axis=-1):
        lib().ref_bspline_brace(self.h_, axis)

    def container(self):
        g = self.geometry()
        out = np.zeros((g[1], g[0], self.nch), np.float32)
        lib().ref_bspline_container(self.h_, ptr(out))
        return out

    def eval(self, crd):
        crd = np.ascontiguousarray(crd, np.float32)
        out = np.zeros((crd.shape[0], self.nch), np.float32)
        lib().ref_bspline_eval(self.h_, ptr(crd), crd.shape[0], ptr(out))
        return out

    def process_affine(self, w, h, aff):
        aff = np.asarray(aff, np.float32)
        out = np.zeros((h, w, self.nch), np.float32)
        lib().ref_bspline_process_affine(self.h_, w, h, ptr(aff), ptr(out))
        return out


def basis_weights(degree, delta):
    w = np.zeros(degree + 1, np.float32)
    lib().ref_basis_weights(degree, delta, ptr(w))
    return w


def poles(degree):
    p = np.zeros(max(degree // 2, 1), np.longdouble)
    lib().ref_poles(degree, ptr(p))
    return p[:degree // 2]


def filter_2d(img, degree, bc0, bc1):
    """zimt::prefilter of a plain 2-D array (no frame), both axes, in place copy"""
    a = np.ascontiguousarray(img, np.float32).copy()
    h, w, nch = a.shape
    lib().ref_filter_2d(ptr(a), w, h, nch, degree, bc0, bc1)
    return a


def lut_eval(knots, vin):
    """lut_based_tf's zimt calls (envutil_payload.cc:251-287) on `knots`"""
    knots = np.ascontiguousarray(knots, np.float32)
    vin = np.ascontiguousarray(vin, np.float32)
    out = np.zeros_like(vin)
    f = lib().ref_lut_eval
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p]
    f(knots.ctypes.data, len(knots), vin.ctypes.data, len(vin), out.ctypes.data)
    return out


def lcp_factor(a, b, c, x):
    """project::lcp<float, 16>(a, b, c).eval on 16-lane vectors (lens_correction.h:224-235)"""
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros_like(x)
    f = lib().ref_lcp_factor
    f.restype = None
    f.argtypes = [C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_long, C.c_void_p]
    f(a, b, c, x.ctypes.data, len(x), out.ctypes.data)
    return out


def inverse_lcp(a, b, c, r_max, sz, x):
    """project::inverse_lcp<float, 16>(a, b, c, r_max, sz).eval (lens_correction.h:236-301) and the
    prefiltered core of its spline model. NB the reference asserts when Newton's iteration does not
    find an inverse (strong coefficients at a large r_max): keep the sets mild"""
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros_like(x)
    knots = np.zeros(sz + 4, np.float32)
    f = lib().ref_inverse_lcp
    f.restype = None
    f.argtypes = [C.c_double] * 4 + [C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_int]
    f(a, b, c, r_max, sz, x.ctypes.data, len(x), out.ctypes.data, knots.ctypes.data, len(knots))
    return out, knots


def binomial_alpha(plane):
    """zimt::convolve(alpha, alpha, {REFLECT, REFLECT}, {1, 4, 6, 4, 1} / 16, 2): the call of
    environment.h:833-843 on a (h, w) float plane"""
    out = np.ascontiguousarray(plane, np.float32).copy()
    f = lib().ref_binomial_alpha
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_long, C.c_long]
    f(out.ctypes.data, out.shape[1], out.shape[0])
    return out


def masking(nch, paint, n):
    """masking_t<2, nch, 16> (masking.h:74-93) on n pixels"""
    out = np.full((n, nch), np.nan, np.float32)
    f = lib().ref_masking
    f.restype = None
    f.argtypes = [C.c_int, C.c_float, C.c_long, C.c_void_p]
    f(nch, paint, n, ptr(out))
    return out


def alpha_masking(spline, paint, crd):
    """alpha_masking_t<nch, 16> (masking.h:95-135) over a RefSpline of 2 or 4 channels at spline coordinates crd"""
    crd = np.ascontiguousarray(crd, np.float32)
    out = np.zeros((crd.shape[0], spline.nch), np.float32)
    f = lib().ref_alpha_masking
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_long, C.c_void_p]
    f(spline.h_, paint, ptr(crd), crd.shape[0], ptr(out))
    return out
