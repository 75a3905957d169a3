#!/usr/bin/env python3
"""How long may the GPU sit idle before the headline kernel's next launches run at lower clocks?
200 launches (kernel-only loop), an idle gap on the host, then 20 launches timed with HIP events.
usage (GPU box): python tools/clock_gap_probe.py"""
import os, sys, time
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
t = lambda n: ea.render_timed(args, [src], out.data_ptr(), n, nch, 0, th, None)
print("first 20: %.4f  next 200: %.4f" % (t(20), t(200)))
for gap in (0.0, 0.0005, 0.002, 0.01, 0.05, 0.2, 1.0):
    t(200)
    time.sleep(gap)
    a = t(5); b = t(20)
    print("gap %.4f s: next 5 launches %.4f ms, the 20 after %.4f ms" % (gap, a, b), flush=True)
