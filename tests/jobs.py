"""Shared job construction for the parity tests: one description drives both
the CPU oracle (tests/euo.py) and the HIP library (envutil_amd). TEST CODE."""
import ctypes as C
import math

import numpy as np

import euo


def synth_image(w, h, nch, seed=12345):
    """smooth field + mild LCG noise, deterministic."""
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    out = np.zeros((h, w, nch), np.float32)
    for c in range(nch):
        if nch in (2, 4) and c == nch - 1:
            out[:, :, c] = 1.0
            continue
        smooth = 0.5 + 0.25 * np.sin(2 * np.pi * (3 + c) * x / w) * np.cos(2 * np.pi * (2 + c) * y / h)
        n = w * h
        # lcg(seed + c; a=1664525, c=1013904223), one step per pixel index
        idx = (np.arange(n, dtype=np.uint64) + np.uint64(1))
        state = (np.uint64(seed + c) + idx * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
        state = (state * np.uint64(1664525) + np.uint64(1013904223)) & np.uint64(0xFFFFFFFF)
        noise = ((state >> np.uint64(8)).astype(np.float64) / float(1 << 24)).reshape(h, w)
        out[:, :, c] = (smooth + 0.05 * noise - 0.025).astype(np.float32)
    return out


def synth_cubefaces(face, nch, seed=777):
    return synth_image(face, 6 * face, nch, seed)


def source_bcs(prj, hfov_rad):
    bc0 = euo.REFLECT
    if prj in (euo.SPHERICAL, euo.CYLINDRICAL) and abs(hfov_rad - 2 * math.pi) < 1e-6:
        bc0 = euo.PERIODIC
    return bc0, euo.REFLECT


class OracleSource:
    """source_t / cubemap_t set-up on the CPU oracle (environment.h:594-962,
    cubemap.h:1147-1233) -> euo.Source"""

    def __init__(self, prj, width, height, hfov_deg, pixels, spline_degree,
                 prefilter_degree=None, yaw=0.0, pitch=0.0, roll=0.0, brighten=1.0,
                 support_min=8, tile=64, lens=None, window=None, translation=None, masked=-1):
        """window = (window_width, window_height, x_offset, y_offset): `pixels` is that
        window of a width x height frame (a cropped PTO image, envutil_basic.h:447-470)"""
        if prefilter_degree is None:
            prefilter_degree = spline_degree
        hf = math.radians(hfov_deg)
        pixels = np.ascontiguousarray(pixels, np.float32)
        nch = pixels.shape[2]
        s = euo.Source()
        s.projection = prj
        s.hfov = hf
        s.width, s.height = width, height
        s.window_width, s.window_height = width, height
        if window is not None:
            s.window_width, s.window_height, s.window_x_offset, s.window_y_offset = window
        s.yaw, s.pitch, s.roll = (math.radians(v) for v in (yaw, pitch, roll))
        s.brighten = brighten
        s.mask_paint = masked + 1
        s.step = euo.lib().euo_get_step(prj, width, height, hf)
        if lens:
            # PTO lens parameters a, b, c (radial), h, v (shift), g, t (shear)
            for k, v in lens.items():
                setattr(s, {"g": "shear_g", "t": "shear_t"}.get(k, k), v)
            s.has_lcp = int(any(lens.get(k, 0.0) != 0.0 for k in "abc"))
        if translation:
            # PTO TrX, TrY, TrZ (model space units) and the translation plane's Tpy, Tpp (+ roll), degrees
            s.tr_x, s.tr_y, s.tr_z = (translation.get(k, 0.0) for k in ("x", "y", "z"))
            s.tp_y, s.tp_p, s.tp_r = (math.radians(translation.get(k, 0.0)) for k in ("tp_y", "tp_p", "tp_r"))
        if prj in (euo.CUBEMAP, euo.BIATAN6):
            m, ir = euo.cubemap_build(pixels, spline_degree, prefilter_degree, hf,
                                      support_min, tile)
            self.container = ir
            sp = euo.Spline()
            sp.data = ir.ctypes.data_as(C.POINTER(C.c_float))
            sp.shape[0], sp.shape[1] = m.section_px, 6 * m.section_px
            sp.stride[0], sp.stride[1] = 1, m.section_px
            sp.core[0], sp.core[1] = m.section_px, 6 * m.section_px
            sp.bc[0] = sp.bc[1] = euo.REFLECT
            sp.degree = spline_degree
            sp.nch = nch
            s.spl = sp
            s.refc_md = np.float32(m.refc_md)
            s.model_to_px = np.float32(m.model_to_px)
            s.section_px = m.section_px
            self.bc = (euo.REFLECT, euo.REFLECT)
        else:
            bc0, bc1 = source_bcs(prj, hf)
            b = euo.BSpline(pixels, spline_degree, bc0, bc1)
            if prj == euo.SPHERICAL and abs(hf - 2 * math.pi) < 1e-6 and width == 2 * height:
                b.spherical_prefilter(prefilter_degree)
            else:
                b.prefilter(prefilter_degree)
            self.bspline = b
            self.container = b.container
            s.spl = b.s
            self.bc = (bc0, bc1)
        self.s = s
        self.nch = nch
        self.degree = spline_degree


def oracle_source_from_container(prj, width, height, hfov_deg, container, geom, degree, nch,
                                 cubemap_metrics=None, yaw=0.0, pitch=0.0, roll=0.0, lens=None):
    """an OracleSource over a finished (braced, prefiltered) coefficient array, e.g. the
    one a GPU source downloads: the oracle then evaluates exactly what the kernels read.
    geom: envutil_amd Container (shape/left/right/core); cubemap_metrics: dict of
    envutil_amd.cubemap_metrics for cubemap / biatan6 sources"""
    o = OracleSource.__new__(OracleSource)
    s = euo.Source()
    s.projection = prj
    s.hfov = math.radians(hfov_deg)
    s.width, s.height = width, height
    s.window_width, s.window_height = width, height
    s.brighten = 1.0
    s.yaw, s.pitch, s.roll = (math.radians(v) for v in (yaw, pitch, roll))
    s.step = euo.lib().euo_get_step(prj, width, height, s.hfov)
    if lens:
        for k, v in lens.items():
            setattr(s, {"g": "shear_g", "t": "shear_t"}.get(k, k), v)
        s.has_lcp = int(any(lens.get(k, 0.0) != 0.0 for k in "abc"))
    sp = euo.Spline()
    sp.data = container.ctypes.data_as(C.POINTER(C.c_float))
    sp.shape[0], sp.shape[1] = geom.shape[0], geom.shape[1]
    sp.stride[0], sp.stride[1] = 1, geom.shape[0]
    sp.left[0], sp.left[1] = geom.left[0], geom.left[1]
    sp.right[0], sp.right[1] = geom.right[0], geom.right[1]
    sp.core[0], sp.core[1] = geom.core[0], geom.core[1]
    if prj in (euo.CUBEMAP, euo.BIATAN6):
        sp.bc[0] = sp.bc[1] = euo.REFLECT
        s.refc_md = np.float32(cubemap_metrics["refc_md"])
        s.model_to_px = np.float32(cubemap_metrics["model_to_px"])
        s.section_px = cubemap_metrics["section_px"]
    else:
        sp.bc[0], sp.bc[1] = source_bcs(prj, s.hfov)
    sp.degree = degree
    sp.nch = nch
    s.spl = sp
    o.s, o.nch, o.degree, o.container = s, nch, degree, container
    o.bc = (sp.bc[0], sp.bc[1])
    return o


def oracle_render(args, osrc, stage=0, row_begin=0, row_end=None, nthreads=8, nch=None):
    """args: envutil_amd.arguments (only its plain fields are read); osrc: one
    OracleSource or a list of them (multi-facet job)"""
    srcs = osrc if isinstance(osrc, (list, tuple)) else [osrc]
    osrc = srcs[0]
    arr = (euo.Source * len(srcs))(*[o.s for o in srcs])
    j = euo.Job()
    j.projection = args.projection
    j.width, j.height = args.width, args.height
    j.x0, j.x1, j.y0, j.y1 = (float(v) for v in args.extent)
    j.yaw, j.pitch, j.roll = (math.radians(v) for v in (args.yaw, args.pitch, args.roll))
    j.nch = nch or osrc.nch
    taps = None
    if args.twine_spread is not None:
        taps = np.ascontiguousarray(args.twine_spread, np.float32)
        j.ntaps = len(taps)
        j.taps = taps.ctypes.data_as(C.POINTER(C.c_float))
    j.row_begin = row_begin
    j.row_end = args.height if row_end is None else row_end
    j.stage = stage
    j.nthreads = nthreads
    j.synopsis = 1 if getattr(args, "synopsis", "panorama") == "hdr_merge" else 0
    if getattr(args, "single", None) is not None:
        # --single: the target recreates this facet (an OracleSource built with the same parameters)
        j.single = C.pointer(args.single_oracle.s)
    w = args.width
    if getattr(args, "store_cropped", False):
        x0, x1, y0, y1 = args.p_crop
        j.crop_x0, j.crop_y0, j.crop_w, j.crop_h = x0, y0, x1 - x0, y1 - y0
        w = x1 - x0
        if row_end is None:
            j.row_end = y1 - y0
    if getattr(args, "tethered", False):
        j.screen = 1
        out = np.zeros((j.row_end - j.row_begin, w), np.uint32)
        rc = euo.lib().euo_render(C.byref(j), arr, len(srcs), out.ctypes.data_as(C.c_void_p), w)
        assert rc == 0, rc
        return out
    och = 3 if stage else (nch or osrc.nch)
    out = np.zeros((j.row_end - j.row_begin, w, och), np.float32)
    rc = euo.lib().euo_render(C.byref(j), arr, len(srcs), euo.ptr(out), w * och)
    assert rc == 0, rc
    return out


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def ulp_diff(a, b):
    """distance in float32 ULPs (monotone integer mapping of the bit patterns)"""
    def key(v):
        u = bits(v).astype(np.int64)
        return np.where(u & 0x80000000, 0x80000000 - u, u)
    return np.abs(key(a) - key(b))
