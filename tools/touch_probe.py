#!/usr/bin/env python3
"""EXPERIMENT: does touching a cube face's source region one face ahead, from a second stream,
shorten the headline frame? The frame is rendered as six launches sets (one per face, rows f*4096..(f+1)*4096) on
stream A; while face f renders, stream B reads one float per 128-byte line of the bounding box of face f+1's
source coordinates (from a stage-2 render of a small target of the same geometry). Prints ms per frame with and
without the touches. Not product code."""
import ctypes as C
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import envutil_amd as ea
from bench import synth_on_device, _DevBuf

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
ea.lib().eu_hip_init(0)
sw, sh, tw, th, nch, deg = 16384, 8192, 4096, 24576, 3, 3
img = synth_on_device(torch, dev, sw, sh, nch)
host = img.cpu().numpy(); del img; torch.cuda.empty_cache()
fct = ea.facet_spec(ea.SPHERICAL, sw, sh, 360.0, nchannels=nch)
src = ea.Source.load(fct, host, deg); del host
args = ea.arguments(ea.CUBEMAP, tw, th, 90.0, spline_degree=deg)
# source bounding boxes per face from the source coordinates of a 6 x 64 target of the same geometry
small = ea.arguments(ea.CUBEMAP, 64, 384, 90.0, spline_degree=deg)
crd = ea.render(small, src, stage=2)
geom, _ = src.info()
ptr, n = src.device_ptr()
buf = torch.as_tensor(_DevBuf(ptr, n), device=dev)
cw, chh = int(geom.shape[0]), int(geom.shape[1])
cont = buf.view(chh, cw * nch)
boxes = []
for f in range(6):
    c = crd[64 * f:64 * (f + 1)]
    x0, x1 = int(c[..., 0].min()) - 2, int(c[..., 0].max()) + 4
    y0, y1 = int(c[..., 1].min()) - 2, int(c[..., 1].max()) + 4
    wrap = (x1 - x0) > sw * 0.75          # the face across the +-180 degree seam: two column ranges, take all columns
    if wrap: x0, x1 = 0, sw
    x0, y0 = max(x0, 0), max(y0, 0); x1, y1 = min(x1, sw), min(y1, sh)
    boxes.append((x0 + int(geom.left[0]), x1 + int(geom.left[0]), y0 + int(geom.left[1]), y1 + int(geom.left[1])))
    print("face", f, "source box x", x0, x1, "y", y0, y1, "MB", (x1 - x0) * (y1 - y0) * 12 / 1e6, file=sys.stderr)
out = torch.empty((th, tw, nch), device=dev, dtype=torch.float32)
srcs = (C.c_void_p * 1)(src.handle)
A, B = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
tg = [args.target(nch, 4096 * f, 4096 * (f + 1), 0, None) for f in range(6)]
sink = torch.zeros(1, device=dev)

def touch(f):
    x0, x1, y0, y1 = boxes[f]
    v = cont[y0:y1, x0 * nch:x1 * nch:32]      # one float per 128 bytes
    sink.add_(v.sum())

def frame(with_touch):
    for f in range(6):
        if with_touch and f + 1 < 6:
            ev = torch.cuda.Event(); ev.record(A)          # face f is about to start on A
            B.wait_event(ev)
            with torch.cuda.stream(B):
                touch(f + 1)
        rc = ea.lib().eu_hip_render(C.byref(tg[f]), srcs, 1, C.c_void_p(out[4096 * f:].data_ptr()), tw * nch * 4, 1,
                                    C.c_void_p(A.cuda_stream))
        assert rc == 0, ea.lib().eu_hip_last_error()

for mode in (False, True, False, True):
    for _ in range(5): frame(mode)
    torch.cuda.synchronize(); ea.lib().eu_hip_sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(A)
    for _ in range(20): frame(mode)
    e1.record(A)
    torch.cuda.synchronize(); ea.lib().eu_hip_sync()
    print("touch" if mode else "plain", round(e0.elapsed_time(e1) / 20, 4), "ms per frame")
