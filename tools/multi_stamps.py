#!/usr/bin/env python3
"""Where a wave of the multi-facet kernel spends its cycles (config 5): the -DEU_MULTI_STAMPS build
(tools/mkvariant_multi.sh mst "-DEU_MULTI_STAMPS") sums s_memtime differences per phase of eu_synopsis's alpha
path. usage (GPU box): EU_HIP_LIB=$PWD/envutil_amd/build/libeu_hip_mst.so python tools/multi_stamps.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import envutil_amd as ea
from envutil_amd.api import PROJECTION_NAMES

(sname, sw, sh, shfov), (tname, tw, th, thfov), nch, degree, twine, ypr = bench.WORKLOADS["config5"]
dev = torch.device("cuda", 0)
views = [(0, 0, 0), (90, 0, 0), (180, 0, 0), (270, 0, 0), (0, 90, 0), (0, -90, 0)]
srcs = []
for v in views:
    img = bench.synth_on_device(torch, dev, sw, sh, nch)
    img[:, :, nch - 1] = 1.0
    host = img.cpu().numpy(); del img
    srcs.append(ea.Source.load(ea.facet_spec(PROJECTION_NAMES.index(sname), sw, sh, shfov, nchannels=nch, yaw=v[0], pitch=v[1],
                                             roll=v[2], lens=dict(a=0.01, b=-0.03, c=0.02)), host, degree))
    del host
args = ea.arguments(PROJECTION_NAMES.index(tname), tw, th, thfov, spline_degree=degree)
out = torch.zeros((th, tw, nch), device=dev, dtype=torch.float32)
ms = ea.render_timed(args, srcs, out.data_ptr(), 5, nch, 0, th, None)
acc = (C.c_ulonglong * 4)()
ea.lib().eu_multi_stamps_read(acc)            # drop what the launches so far have summed
ms = ea.render_timed(args, srcs, out.data_ptr(), 4, nch, 0, th, None)
assert ea.lib().eu_multi_stamps_read(acc) == 0
w = acc[3] & 0xffffffff
print("exact hit tests per wave (of %d facets): %.2f" % (len(srcs), (acc[3] >> 32) / w))
print("kernel %.3f ms (stamped build); waves %d; cycles per wave: mask pass %.0f, top / all-top evaluation %.0f, compositing %.0f"
      % (ms, w, acc[0] / w, acc[1] / w, acc[2] / w))
