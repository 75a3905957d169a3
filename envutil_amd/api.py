"""ctypes binding of include/eu_hip.h and a host mirror of envutil's job surface."""
import ctypes as C
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

SPHERICAL, CYLINDRICAL, RECTILINEAR, STEREOGRAPHIC, FISHEYE, CUBEMAP, BIATAN6 = range(7)
BC_MIRROR, BC_PERIODIC, BC_REFLECT, BC_NATURAL, BC_CONSTANT, BC_ZEROPAD, BC_GUESS = range(7)
PROJECTION_NAMES = ["spherical", "cylindrical", "rectilinear", "stereographic",
                    "fisheye", "cubemap", "biatan6"]


class EuError(RuntimeError):
    pass


class Facet(C.Structure):
    """struct eu_facet"""
    _fields_ = [("projection", C.c_int32), ("nchannels", C.c_int32),
                ("hfov", C.c_double),
                ("width", C.c_int32), ("height", C.c_int32),
                ("window_width", C.c_int32), ("window_height", C.c_int32),
                ("window_x_offset", C.c_int32), ("window_y_offset", C.c_int32),
                ("yaw", C.c_double), ("pitch", C.c_double), ("roll", C.c_double),
                ("brighten", C.c_double), ("step", C.c_double),
                ("has_lcp", C.c_int32),
                ("a", C.c_double), ("b", C.c_double), ("c", C.c_double),
                ("h", C.c_double), ("v", C.c_double), ("s", C.c_double),
                ("shear_g", C.c_double), ("shear_t", C.c_double),
                ("tr_x", C.c_double), ("tr_y", C.c_double), ("tr_z", C.c_double),
                ("tp_y", C.c_double), ("tp_p", C.c_double), ("tp_r", C.c_double),
                ("mask_paint", C.c_int32)]


class Container(C.Structure):
    """struct eu_container"""
    _fields_ = [("shape", C.c_int64 * 2), ("left", C.c_int64 * 2),
                ("right", C.c_int64 * 2), ("core", C.c_int64 * 2)]


class Target(C.Structure):
    """struct eu_target"""
    _fields_ = [("projection", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("x0", C.c_double), ("x1", C.c_double), ("y0", C.c_double), ("y1", C.c_double),
                ("yaw", C.c_double), ("pitch", C.c_double), ("roll", C.c_double),
                ("nchannels", C.c_int32), ("ntaps", C.c_int32),
                ("taps", C.POINTER(C.c_float)),
                ("row_begin", C.c_int32), ("row_end", C.c_int32), ("stage", C.c_int32),
                ("crop_x0", C.c_int32), ("crop_y0", C.c_int32),
                ("crop_w", C.c_int32), ("crop_h", C.c_int32),
                ("out_format", C.c_int32),
                ("band_rows", C.c_int32), ("band_count", C.c_int32), ("band_index", C.c_int32),
                ("synopsis", C.c_int32), ("single", C.POINTER(Facet))]


OUT_FLOAT, OUT_SRGBA8 = 0, 1
SYN_PANORAMA, SYN_HDR_MERGE = 0, 1


def lib_path():
    # EU_HIP_LIB: another build of the library (build-time A/B experiments)
    return os.environ.get("EU_HIP_LIB") or os.path.join(HERE, "lib", "libeu_hip.so")


def build(force=False):
    """compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)"""
    if force:
        subprocess.check_call(["make", "-s", "-C", HERE, "clean"])
    subprocess.check_call(["make", "-s", "-j8", "-C", HERE])
    return lib_path()


_lib = None


def _share_hip_runtime():
    """A PyTorch-ROCm wheel carries its own libamdhip64 / libhsa-runtime64, and a
    process that ends up with two HIP runtimes sees no GPU in the second one.
    When torch is installed, load its runtime first (without importing torch):
    libeu_hip.so's NEEDED libamdhip64.so.7 then binds to the same copy, whichever
    of the two is imported first."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise EuError(f"{p} is missing: run envutil_amd.build() (hipcc, gfx950). "
                      "There is no fallback implementation.")
    _share_hip_runtime()
    L = C.CDLL(p)
    vp, i32, f64 = C.c_void_p, C.c_int, C.c_double
    L.eu_hip_last_error.restype = C.c_char_p
    L.eu_hip_get_step.restype = f64
    L.eu_hip_get_step.argtypes = [i32, i32, i32, f64]
    L.eu_hip_get_extent.argtypes = [i32, i32, i32, f64, vp]
    L.eu_hip_make_spread.argtypes = [i32, i32, C.c_float, C.c_float, C.c_float, vp, i32]
    L.eu_hip_cubemap_metrics.argtypes = [i32, f64, i32, i32, vp, vp, vp, vp]
    L.eu_hip_container_geometry.argtypes = [i32, i32, i32, C.c_int64, C.c_int64, vp]
    L.eu_hip_source_load.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    L.eu_hip_source_adopt.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    L.eu_hip_source_alloc.argtypes = [vp, i32, i32, i32, vp]
    L.eu_hip_source_device_ptr.argtypes = [vp, vp, vp]
    L.eu_hip_source_download.argtypes = [vp, vp, C.c_size_t]
    L.eu_hip_source_info.argtypes = [vp, vp, vp]
    L.eu_hip_source_update_facet.argtypes = [vp, vp]
    L.eu_hip_source_release.argtypes = [vp]
    L.eu_hip_render.argtypes = [vp, vp, i32, vp, C.c_size_t, i32, vp]
    L.eu_hip_init_devices.argtypes = [vp, i32]
    L.eu_hip_render_devices.argtypes = [vp, vp, i32, vp, C.c_size_t, i32]
    L.eu_hip_device_strips.argtypes = [vp, vp, i32, vp, vp]
    L.eu_hip_render_timed.argtypes = [vp, vp, i32, vp, C.c_size_t, i32, vp]
    L.eu_hip_layout_segments.argtypes = [vp, vp, i32, vp, i32, vp]
    L.eu_hip_band_rows.argtypes = [i32, i32, i32, i32]
    L.eu_hip_band_rows.restype = i32
    L.eu_hip_malloc.argtypes = [vp, C.c_size_t]
    L.eu_hip_free.argtypes = [vp]
    L.eu_hip_memcpy_d2h.argtypes = [vp, vp, C.c_size_t]
    L.eu_hip_memcpy_h2d.argtypes = [vp, vp, C.c_size_t]
    _lib = L
    return L


def _check(rc):
    if rc < 0:
        raise EuError(f"eu_hip error {rc}: {lib().eu_hip_last_error().decode()}")
    return rc


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def device_count():
    return lib().eu_hip_device_count()


def get_extent(projection, width, height, hfov):
    e = np.zeros(4, np.float64)
    _check(lib().eu_hip_get_extent(projection, width, height, hfov, _ptr(e)))
    return e


def get_step(projection, width, height, hfov):
    return lib().eu_hip_get_step(projection, width, height, hfov)


def make_spread(w, h=0, d=1.0, sigma=0.0, threshold=0.0):
    n = max(w, 2) * max(h if h > 0 else max(w, 2), 1)
    out = np.zeros((n, 3), np.float32)
    k = _check(lib().eu_hip_make_spread(w, h, d, sigma, threshold, _ptr(out), n))
    return out[:k].copy()


class MaskPolygon(C.Structure):
    _fields_ = [("n", C.c_int), ("x", C.c_void_p), ("y", C.c_void_p)]


def facet_alpha(pixels, polygons=(), crop=None, crop_kind=0):
    """eu_hip_facet_alpha: PTO exclude masks (list of (xs, ys) vertex arrays) and the lens crop
    (x0, x1, y0, y1; kind 1 rectangular, 2 elliptic) of a facet, multiplied into `pixels`
    ((h, w, 2|4) float32, in place). Host function: works without a device. Returns the alpha plane."""
    assert pixels.dtype == np.float32 and pixels.ndim == 3 and pixels.flags.c_contiguous
    h, w, nch = pixels.shape
    keep = [(np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32)) for x, y in polygons]
    arr = (MaskPolygon * max(len(keep), 1))()
    for i, (x, y) in enumerate(keep):
        arr[i].n, arr[i].x, arr[i].y = len(x), x.ctypes.data, y.ctypes.data
    alpha = np.zeros((h, w), np.float32)
    c = crop if crop is not None else (0, 0, 0, 0)
    f = lib().eu_hip_facet_alpha
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    _check(f(_ptr(pixels), w, h, nch, C.cast(arr, C.c_void_p), len(keep), crop_kind if crop is not None else 0,
             c[0], c[1], c[2], c[3], _ptr(alpha)))
    return alpha


def cubemap_metrics(face_px, face_fov=math.pi / 2, support_min=8, tile_px=64):
    sec, lf = C.c_int64(), C.c_int64()
    refc, m2p = C.c_double(), C.c_double()
    _check(lib().eu_hip_cubemap_metrics(face_px, face_fov, support_min, tile_px,
                                        C.byref(sec), C.byref(lf), C.byref(refc), C.byref(m2p)))
    return dict(section_px=sec.value, left_frame_px=lf.value, refc_md=refc.value,
                model_to_px=m2p.value)


def container_geometry(degree, bc0, bc1, w, h):
    g = Container()
    _check(lib().eu_hip_container_geometry(degree, bc0, bc1, w, h, C.byref(g)))
    return g


class facet_spec:
    """Host mirror of envutil's facet_spec (envutil_basic.h:432-520): the fields
    the render path reads. Angles in DEGREES here, as on envutil's command
    line; converted to radians when the C struct is built
    (envutil_main.cc:957-960)."""

    def __init__(self, projection, width, height, hfov, nchannels=3, yaw=0.0,
                 pitch=0.0, roll=0.0, brighten=1.0, window=None, lens=None, translation=None, masked=-1):
        self.projection = projection
        self.masked = masked            # --mask_for: -1 ordinary, 0 painted black, 1 painted white
        self.width, self.height = width, height
        self.hfov = hfov
        self.nchannels = nchannels
        self.yaw, self.pitch, self.roll = yaw, pitch, roll
        self.brighten = brighten
        self.window = window or (width, height, 0, 0)
        self.lens = lens or {}          # PTO a, b, c, h, v, g (shear_g), t (shear_t)
        # PTO TrX, TrY, TrZ as x, y, z (model space units), Tpy, Tpp as tp_y, tp_p (+ tp_r), degrees
        self.translation = translation or {}

    def c_struct(self):
        f = Facet()
        f.projection = self.projection
        f.nchannels = self.nchannels
        f.hfov = math.radians(self.hfov)
        f.width, f.height = self.width, self.height
        (f.window_width, f.window_height, f.window_x_offset, f.window_y_offset) = self.window
        f.yaw, f.pitch, f.roll = (math.radians(v) for v in (self.yaw, self.pitch, self.roll))
        f.brighten = self.brighten
        f.step = get_step(self.projection, self.width, self.height, f.hfov)
        for k, v in self.lens.items():
            setattr(f, {"g": "shear_g", "t": "shear_t"}.get(k, k), v)
        f.has_lcp = int(any(self.lens.get(k, 0.0) != 0.0 for k in "abc"))
        f.tr_x, f.tr_y, f.tr_z = (self.translation.get(k, 0.0) for k in ("x", "y", "z"))
        f.tp_y, f.tp_p, f.tp_r = (math.radians(self.translation.get(k, 0.0)) for k in ("tp_y", "tp_p", "tp_r"))
        f.mask_paint = self.masked + 1
        return f


class Source:
    """A source image resident in HBM (the asset_handler entry,
    environment.h:84-227)."""

    def __init__(self, handle, fct):
        self.handle = handle
        self.fct = fct

    @classmethod
    def load(cls, fct, pixels, spline_degree, prefilter_degree=None, support_min=8,
             tile_size=64):
        """pixels -> braced + prefiltered coefficients, on the device"""
        if prefilter_degree is None:
            prefilter_degree = spline_degree
        pixels = np.ascontiguousarray(pixels, np.float32)
        cf = fct.c_struct()
        h = C.c_void_p()
        _check(lib().eu_hip_source_load(C.byref(cf), _ptr(pixels), spline_degree,
                                        prefilter_degree, support_min, tile_size, C.byref(h)))
        return cls(h, fct)

    @classmethod
    def adopt(cls, fct, container, spline_degree, bc0=BC_REFLECT, bc1=BC_REFLECT,
              support_min=8, tile_size=64):
        """upload an already braced + prefiltered container"""
        container = np.ascontiguousarray(container, np.float32)
        cf = fct.c_struct()
        h = C.c_void_p()
        _check(lib().eu_hip_source_adopt(C.byref(cf), _ptr(container), spline_degree, bc0,
                                         bc1, support_min, tile_size, C.byref(h)))
        return cls(h, fct)

    @classmethod
    def alloc(cls, fct, spline_degree, support_min=8, tile_size=64):
        """an unfilled container in HBM (to be filled by a broadcast)"""
        cf = fct.c_struct()
        h = C.c_void_p()
        _check(lib().eu_hip_source_alloc(C.byref(cf), spline_degree, support_min, tile_size,
                                         C.byref(h)))
        return cls(h, fct)

    def update_facet(self, fct):
        """the facet's geometry changed (orientation, hfov, lens, brighten): rebuild the
        evaluator / mount parameters, keep the resident coefficients"""
        _check(lib().eu_hip_source_update_facet(self.handle, C.byref(fct.c_struct())))
        self.fct = fct

    def device_ptr(self):
        p = C.c_void_p()
        n = C.c_size_t()
        _check(lib().eu_hip_source_device_ptr(self.handle, C.byref(p), C.byref(n)))
        return p.value, n.value

    def info(self):
        g = Container()
        n = C.c_int()
        _check(lib().eu_hip_source_info(self.handle, C.byref(g), C.byref(n)))
        return g, n.value

    def download(self):
        g, n = self.info()
        out = np.zeros((g.shape[1], g.shape[0], n), np.float32)
        _check(lib().eu_hip_source_download(self.handle, _ptr(out), out.size))
        return out

    def release(self):
        if self.handle:
            lib().eu_hip_source_release(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class arguments:
    """Host mirror of the target half of envutil's global `args`
    (envutil_basic.h:633-705): projection, size, hfov -> extent, camera
    orientation (degrees), spline degree, twining."""

    def __init__(self, projection, width, height, hfov, yaw=0.0, pitch=0.0, roll=0.0,
                 spline_degree=1, prefilter_degree=None, twine=0, twine_width=1.0,
                 twine_sigma=0.0, twine_threshold=0.0, support_min=8, tile_size=64,
                 crop=None, tethered=False, synopsis="panorama", single=None):
        # store_cropped + p_crop_x0/x1/y0/y1 (envutil_basic.h:684-687) as
        # (x0, x1, y0, y1); tethered: the job writes packed sRGBA8 words
        # (args.p_screen_data, envutil_payload.cc:524-530)
        self.store_cropped = crop is not None
        self.p_crop = tuple(crop) if crop is not None else None
        self.tethered = tethered
        # args.synopsis (envutil_main.cc:232): how several facets are composed, "panorama" or "hdr_merge"
        if synopsis not in ("panorama", "hdr_merge"):
            raise ValueError("synopsis must be panorama or hdr_merge")
        self.synopsis = synopsis
        # args.single: the facet_spec this target recreates ((facet_base&) args = fspec, envutil_main.cc:1161-1180);
        # projection, size, hfov and orientation of the target must be the facet's own (see for_single)
        self.single = single
        self.projection = projection
        self.width, self.height = width, height
        self.hfov = hfov
        self.yaw, self.pitch, self.roll = yaw, pitch, roll
        self.spline_degree = spline_degree
        self.prefilter_degree = spline_degree if prefilter_degree is None else prefilter_degree
        self.twine = twine
        self.twine_width, self.twine_sigma = twine_width, twine_sigma
        self.twine_threshold = twine_threshold
        self.support_min, self.tile_size = support_min, tile_size
        # envutil_main.cc:1203-1232
        self.extent = get_extent(projection, width, height, math.radians(hfov))
        self.step = (self.extent[1] - self.extent[0]) / width
        self.twine_spread = None
        if twine:
            # arguments::twine_setup, envutil_main.cc:1405-1616 (explicit twine)
            self.twine_spread = make_spread(twine, twine, twine_width, twine_sigma,
                                            twine_threshold)

    @classmethod
    def for_single(cls, fct, **kw):
        """the target of a --single job: the facet's own geometry taken over as target geometry"""
        return cls(fct.projection, fct.width, fct.height, fct.hfov, yaw=fct.yaw, pitch=fct.pitch, roll=fct.roll,
                   single=fct, **kw)

    def target(self, nchannels, row_begin=0, row_end=None, stage=0, band=None):
        """band = (band_rows, band_count, band_index): this call renders the
        interleaved row bands of one part (eu_target.band_*); rows are local"""
        t = Target()
        t.projection = self.projection
        t.width, t.height = self.width, self.height
        t.x0, t.x1, t.y0, t.y1 = (float(v) for v in self.extent)
        t.yaw, t.pitch, t.roll = (math.radians(v) for v in (self.yaw, self.pitch, self.roll))
        t.nchannels = nchannels
        if self.twine_spread is not None:
            t.ntaps = len(self.twine_spread)
            t.taps = self.twine_spread.ctypes.data_as(C.POINTER(C.c_float))
        if self.store_cropped:
            x0, x1, y0, y1 = self.p_crop
            t.crop_x0, t.crop_y0, t.crop_w, t.crop_h = x0, y0, x1 - x0, y1 - y0
        t.out_format = OUT_SRGBA8 if self.tethered else OUT_FLOAT
        t.synopsis = SYN_HDR_MERGE if self.synopsis == "hdr_merge" else SYN_PANORAMA
        if self.single is not None:
            self._single_c = self.single.c_struct()          # kept alive with the arguments object
            t.single = C.pointer(self._single_c)
        nrows = self.out_height
        if band is not None and band[1] > 1:
            t.band_rows, t.band_count, t.band_index = band
            nrows = band_rows(self.out_height, *band)
        t.row_begin = row_begin
        t.row_end = nrows if row_end is None else row_end
        t.stage = stage
        return t

    @property
    def out_width(self):
        return self.p_crop[1] - self.p_crop[0] if self.store_cropped else self.width

    @property
    def out_height(self):
        return self.p_crop[3] - self.p_crop[2] if self.store_cropped else self.height


def layout_segments(args, sources, nchannels=None):
    """(seg_rows, flags): flags[k] = 1 where rows [k * seg_rows, (k + 1) * seg_rows) of the
    job's frame are rendered with the tile layout and cost about 1.55-2x the others
    (eu_hip_layout_segments); flags is empty when the job has no such structure"""
    if not isinstance(sources, (list, tuple)):
        sources = [sources]
    if len(sources) != 1:
        return 512, np.zeros(0, np.uint8)
    nch = nchannels or sources[0].fct.nchannels
    t = args.target(nch)
    arr = (C.c_void_p * 1)(sources[0].handle)
    flags = np.zeros(65536, np.uint8)
    seg = C.c_int(0)
    n = lib().eu_hip_layout_segments(C.byref(t), arr, 1, flags.ctypes.data_as(C.c_void_p), flags.size,
                                     C.byref(seg))
    _check(n if n < 0 else 0)
    return seg.value, flags[:n].copy()


def band_rows(height, rows, count, index):
    """local rows of part `index` when `height` rows are dealt out in bands of
    `rows` rows to `count` parts (eu_hip_band_rows)"""
    return lib().eu_hip_band_rows(height, rows, count, index)


def band_frame_rows(height, rows, count, index):
    """frame row of every local row of that part, in order (numpy int64)"""
    y = np.arange(height)
    return y[(y // rows) % count == index] if count > 1 else y


def render(args, sources, nchannels=None, row_begin=0, row_end=None, stage=0, out=None, band=None):
    """zimt::process(shape, get, act, put, bill) for rows [row_begin, row_end):
    returns (rows, width, nch) float32 on the host - (rows, width) uint32
    sRGBA8 words for a tethered job; width/rows are those of the crop window
    when args.store_cropped."""
    if not isinstance(sources, (list, tuple)):
        sources = [sources]
    nch = nchannels or sources[0].fct.nchannels
    t = args.target(nch, row_begin, row_end, stage, band)
    rows = t.row_end - t.row_begin
    w = args.out_width
    if args.tethered:
        och = 1
        if out is None:
            out = np.zeros((rows, w), np.uint32)
    else:
        och = 3 if stage else nch
        if out is None:
            out = np.zeros((rows, w, och), np.float32)
    arr = (C.c_void_p * len(sources))(*[s.handle for s in sources])
    _check(lib().eu_hip_render(C.byref(t), arr, len(sources), out.ctypes.data_as(C.c_void_p),
                               w * och * 4, 0, None))
    return out


def init_devices(devices):
    """one process, several devices: one slot per entry (the same device may be listed twice)"""
    arr = (C.c_int * len(devices))(*devices)
    _check(lib().eu_hip_init_devices(arr, len(devices)))


def device_slots():
    return lib().eu_hip_device_slots()


def device_strips(args, sources, nchannels=None):
    """the rows eu_hip_render_devices gives every slot for this job: [(begin, end), ...]"""
    if not isinstance(sources, (list, tuple)):
        sources = [sources]
    nch = nchannels or sources[0].fct.nchannels
    t = args.target(nch, 0, None, 0, None)
    n = device_slots()
    b, e = (C.c_int * n)(), (C.c_int * n)()
    arr = (C.c_void_p * len(sources))(*[s.handle for s in sources])
    _check(lib().eu_hip_device_strips(C.byref(t), arr, len(sources), b, e))
    return list(zip(list(b), list(e)))


def render_devices(args, sources, nchannels=None, out=None, out_dev_ptr=None):
    """the whole frame, its rows tiled over the device slots (eu_hip_render_devices): into host memory
    (returned) or, with out_dev_ptr, into device memory of slot 0"""
    if not isinstance(sources, (list, tuple)):
        sources = [sources]
    nch = nchannels or sources[0].fct.nchannels
    t = args.target(nch, 0, None, 0, None)
    rows = t.row_end - t.row_begin
    w = args.out_width
    och = 1 if args.tethered else nch
    arr = (C.c_void_p * len(sources))(*[s.handle for s in sources])
    if out_dev_ptr is not None:
        _check(lib().eu_hip_render_devices(C.byref(t), arr, len(sources), C.c_void_p(out_dev_ptr), w * och * 4, 1))
        return None
    if out is None:
        out = np.zeros((rows, w), np.uint32) if args.tethered else np.zeros((rows, w, och), np.float32)
    _check(lib().eu_hip_render_devices(C.byref(t), arr, len(sources), out.ctypes.data_as(C.c_void_p), w * och * 4, 0))
    return out


def render_timed(args, sources, out_dev_ptr, iters, nchannels=None, row_begin=0,
                 row_end=None, band=None):
    """kernel-only timing with HIP events on the library's stream; the output
    stays in HBM at out_dev_ptr. Returns mean milliseconds per launch."""
    if not isinstance(sources, (list, tuple)):
        sources = [sources]
    nch = nchannels or sources[0].fct.nchannels
    t = args.target(nch, row_begin, row_end, 0, band)
    arr = (C.c_void_p * len(sources))(*[s.handle for s in sources])
    ms = C.c_float()
    och = 1 if args.tethered else nch
    _check(lib().eu_hip_render_timed(C.byref(t), arr, len(sources), C.c_void_p(out_dev_ptr),
                                     args.out_width * och * 4, iters, C.byref(ms)))
    return ms.value


class _hip_dispatch:
    """dispatch_base (envutil_dispatch.h:49-65) for the HIP back-end:
    payload(nchannels, ninputs, projection) runs one render job described by
    the `args` / sources it was bound to and returns 0, like the reference."""

    hwy_target_name = "gfx950"
    hwy_target_str = "HIP/CDNA4"

    def __init__(self):
        self.args = None
        self.sources = None
        self.result = None

    def bind(self, args, sources):
        self.args, self.sources = args, sources
        return self

    def payload(self, nchannels, ninputs, projection):
        a = self.args
        if projection != a.projection:
            raise EuError("payload projection differs from args.projection")
        if (ninputs == 9) != (a.twine_spread is not None):
            raise EuError("ninputs must be 9 with twining and 3 without")
        self.result = render(a, self.sources, nchannels)
        return 0


def get_dispatch():
    return _hip_dispatch()
