"""envutil_amd - MI355X-native reprojection path of envutil.

The product is the shared library ``envutil_amd/lib/libeu_hip.so`` (hand-written
HIP kernels behind the C ABI of ``include/eu_hip.h``). This package is a thin
ctypes binding of that ABI plus a host-side mirror of the part of envutil's
``arguments`` / ``facet_spec`` / ``dispatch_base::payload`` surface that the
render path reads (envutil_basic.h:432-705, envutil_dispatch.h:49-73).

There is no CPU rendering path: if the library is missing, or no HIP device is
present, calls raise.
"""
from .api import (  # noqa: F401
    EuError, Facet, Source, Target, arguments, facet_spec, get_dispatch,
    container_geometry, cubemap_metrics, device_count, get_extent, get_step,
    lib, lib_path, make_spread, render, render_timed, build, band_rows, band_frame_rows,
    layout_segments, facet_alpha, init_devices, device_slots, device_strips, render_devices,
    SPHERICAL, CYLINDRICAL, RECTILINEAR, STEREOGRAPHIC, FISHEYE, CUBEMAP, BIATAN6,
    BC_MIRROR, BC_PERIODIC, BC_REFLECT, BC_NATURAL, BC_CONSTANT,
)
