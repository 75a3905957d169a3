"""DIAGNOSTIC: where does a wave of the headline path spend its cycles?
usage (GPU box): python tools/diag_stamps.py [workload]"""
import ctypes as C
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import envutil_amd as ea
import bench
from envutil_amd.api import PROJECTION_NAMES

wl = sys.argv[1] if len(sys.argv) > 1 else "headline"
(sname, sw, sh, shfov), (tname, tw, th, thfov), nch, degree, twine, ypr = bench.WORKLOADS[wl]
dev = torch.device("cuda:0")
img = bench.synth_on_device(torch, dev, sw, sh, nch).cpu().numpy()
src = ea.Source.load(ea.facet_spec(PROJECTION_NAMES.index(sname), sw, sh, shfov), img, degree)
args = ea.arguments(PROJECTION_NAMES.index(tname), tw, th, thfov, spline_degree=degree)
out = torch.empty((th, tw, nch), device=dev)
t = args.target(nch)
nw = ((tw + 63) // 64) * ((th + 3) // 4) * 4
st = np.zeros((nw, 8), np.uint64)
L = ea.lib()
L.eu_hip_diag_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
srcs = (C.c_void_p * 1)(src.handle)
rc = L.eu_hip_diag_stamps(C.byref(t), srcs, 1, C.c_void_p(out.data_ptr()), tw * nch * 4,
                          st.ctypes.data_as(C.c_void_p), nw)
assert rc == 0, L.eu_hip_last_error()
st = st[st[:, 0] > 0]
d = np.diff(st[:, :6].astype(np.int64), axis=1)
names = ["tables+ray", "coord math", "issue 16 loads", "wait loads", "weights+sum+store"]
print(wl, "waves", len(st), "kernel span cycles", int(st[:, 5].max() - st[:, 0].min()))
life = (st[:, 5] - st[:, 0]).astype(np.int64)
print("wave lifetime: mean %.0f median %.0f p90 %.0f" % (life.mean(), np.median(life), np.percentile(life, 90)))
for i, n in enumerate(names):
    print("  %-20s mean %8.0f  median %8.0f  p90 %8.0f  share %.1f%%" % (
        n, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90), 100 * d[:, i].sum() / life.sum()))
# by face (tile index -> output row -> face)
tiles_x = (tw + 63) // 64
face = (st[:, 7].astype(np.int64) // tiles_x * 4) // (th // 6) if tname == "cubemap" else None
if face is not None:
    for f in range(6):
        m = face == f
        print("  face %d: lifetime mean %.0f, wait loads mean %.0f" % (f, life[m].mean(), d[m, 3].mean()))
