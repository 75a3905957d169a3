#!/usr/bin/env python3
"""LABELLED EXPERIMENT (never the bench line): what the 0-ULP contract costs. The b-spline
weights and the weighted sum with fused multiply-adds (envutil_amd/build/libeu_hip_fma.so, built with
-DEU_FMA_EXPERIMENT; offsets, deltas and the whole coordinate chain untouched) against the shipped library,
which is bit-identical to the oracle: per workload the kernel time of both builds on the same box and the ULP
distance of every float of the frame (config 4: a band of rows - its frame is 6.4 GB).
    python tools/fma_experiment.py render LIBTAG WORKLOAD   -> /tmp/fma_<tag>_<workload>.npy + one line with the time
    python tools/fma_experiment.py compare WORKLOAD         -> ULP histogram of the two frames"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def render(tag, wl):
    import torch
    import bench
    import envutil_amd as ea
    from envutil_amd.api import PROJECTION_NAMES
    (sname, sw, sh, shfov), (tname, tw, th, thfov), nch, degree, twine, ypr = bench.WORKLOADS[wl]
    sprj, tprj = PROJECTION_NAMES.index(sname), PROJECTION_NAMES.index(tname)
    dev = torch.device("cuda", 0)
    views = [(0, 0, 0), (90, 0, 0), (180, 0, 0), (270, 0, 0), (0, 90, 0), (0, -90, 0)] if wl == "config5" else [(0, 0, 0)]
    srcs = []
    for v in views:
        img = bench.synth_on_device(torch, dev, sw, sh, nch)
        if nch in (2, 4):
            img[:, :, nch - 1] = 1.0
        host = img.cpu().numpy()
        del img
        kw = dict(yaw=v[0], pitch=v[1], roll=v[2], lens=dict(a=0.01, b=-0.03, c=0.02)) if wl == "config5" else {}
        srcs.append(ea.Source.load(ea.facet_spec(sprj, sw, sh, shfov, nchannels=nch, **kw), host, degree))
        del host
    args = ea.arguments(tprj, tw, th, thfov, yaw=ypr[0], pitch=ypr[1], roll=ypr[2], spline_degree=degree, twine=twine)
    r0, r1 = (0, th) if wl != "config4" else (th // 2 - 256, th // 2 + 256)
    out = torch.zeros((r1 - r0, tw, nch), device=dev, dtype=torch.float32)
    ea.render_timed(args, srcs, out.data_ptr(), 5, nch, r0, r1, None)
    ms = ea.render_timed(args, srcs, out.data_ptr(), 10, nch, r0, r1, None)
    np.save(f"/tmp/fma_{tag}_{wl}.npy", out.cpu().numpy())
    print(json.dumps({"lib": tag, "workload": wl, "rows": [r0, r1], "kernel_ms": round(ms, 4)}), flush=True)


def compare(wl):
    a = np.load(f"/tmp/fma_default_{wl}.npy").view(np.int32).astype(np.int64)
    b = np.load(f"/tmp/fma_fma_{wl}.npy").view(np.int32).astype(np.int64)
    # ULP distance on the ordered-integer line of IEEE floats
    a = np.where(a < 0, -(a & 0x7fffffff), a)
    b = np.where(b < 0, -(b & 0x7fffffff), b)
    d = np.abs(a - b)
    hist = {str(k): int((d == k).sum()) for k in range(0, 5)}
    hist[">4"] = int((d > 4).sum())
    print(json.dumps({"workload": wl, "floats": int(d.size), "max_ulp": int(d.max()), "mean_ulp": float(d.mean()),
                      "share_differing": float((d > 0).mean()), "ulp_histogram": hist}), flush=True)
    os.remove(f"/tmp/fma_default_{wl}.npy"); os.remove(f"/tmp/fma_fma_{wl}.npy")


if __name__ == "__main__":
    if sys.argv[1] == "render":
        render(sys.argv[2], sys.argv[3])
    else:
        compare(sys.argv[2])
