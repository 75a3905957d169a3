#!/usr/bin/env python3
"""Do the first steps on ANOTHER stream run slower than the kernel-only loop that preceded them on the
library's stream? 200 launches on the library's stream, then 25 steps (eu_hip_render) on (a) the library's
stream, (b) a torch stream, wall clock around them. usage (GPU box): python tools/queue_probe.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import envutil_amd as ea
from envutil_amd.api import PROJECTION_NAMES

(sname, sw, sh, shfov), (tname, tw, th, thfov), nch, degree, twine, ypr = bench.WORKLOADS["headline"]
dev = torch.device("cuda", 0)
img = bench.synth_on_device(torch, dev, sw, sh, nch).cpu().numpy()
src = ea.Source.load(ea.facet_spec(PROJECTION_NAMES.index(sname), sw, sh, shfov, nchannels=nch), img, degree)
args = ea.arguments(PROJECTION_NAMES.index(tname), tw, th, thfov, spline_degree=degree)
out = torch.zeros((th, tw, nch), device=dev, dtype=torch.float32)
tgt = args.target(nch, 0, th, 0, None)
srcs = (C.c_void_p * 1)(src.handle)
st = torch.cuda.Stream(device=dev)
t = lambda n: ea.render_timed(args, [src], out.data_ptr(), n, nch, 0, th, None)


def steps(n, stream):
    for _ in range(n):
        rc = ea.lib().eu_hip_render(C.byref(tgt), srcs, 1, C.c_void_p(out.data_ptr()), tw * nch * 4, 1,
                                    C.c_void_p(stream))
        assert rc == 0


def sync():
    ea.lib().eu_hip_sync(); torch.cuda.synchronize()


print("kernel-only loop: first 20 %.4f, next 200 %.4f" % (t(20), t(200)))
for name, s in (("library stream", None), ("torch stream", st.cuda_stream), ("library stream", None), ("torch stream", st.cuda_stream)):
    t(200)
    steps(5, s); sync()
    t0 = time.perf_counter(); steps(20, s); sync(); dt = time.perf_counter() - t0
    t1 = time.perf_counter(); steps(200, s); sync(); dt2 = time.perf_counter() - t1
    print("%-15s 5 warm-up steps, then 20 steps: %.4f ms per step (wall); the 200 after: %.4f" % (name, dt / 20 * 1e3, dt2 / 200 * 1e3), flush=True)
