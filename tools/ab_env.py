#!/usr/bin/env python3
"""Render one bench workload under several environment settings (read by the library on every
call), compare every frame with the first one bit for bit on the GPU and time each setting with
eu_hip_render_timed. usage:
    python tools/ab_env.py WORKLOAD "EU_HIP_R4=0" "EU_HIP_R4=1" "EU_HIP_R4=1 EU_HIP_R5=0" ...
Prints one line per setting: kernel ms (HIP events, mean of N launches), differing floats."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import envutil_amd as ea  # noqa: E402
from envutil_amd.api import PROJECTION_NAMES  # noqa: E402


def main():
    wl = sys.argv[1]
    settings = sys.argv[2:] or [""]
    reps = int(os.environ.get("AB_REPS", "20"))
    (sname, sw, sh, shfov), (tname, tw, th, thfov), nch, degree, twine, ypr = bench.WORKLOADS[wl]
    sprj, tprj = PROJECTION_NAMES.index(sname), PROJECTION_NAMES.index(tname)
    dev = torch.device("cuda", 0)
    img = bench.synth_on_device(torch, dev, sw, sh, nch)
    if nch in (2, 4):
        img[:, :, nch - 1] = 1.0
    host = img.cpu().numpy()
    del img
    src = ea.Source.load(ea.facet_spec(sprj, sw, sh, shfov, nchannels=nch), host, degree)
    del host
    args = ea.arguments(tprj, tw, th, thfov, yaw=ypr[0], pitch=ypr[1], roll=ypr[2],
                        spline_degree=degree, twine=twine)
    ref = None
    keys = set()
    for s in settings:
        for kv in s.split():
            keys.add(kv.split("=")[0])
    for s in settings:
        for k in keys:
            os.environ.pop(k, None)
        for kv in s.split():
            k, v = kv.split("=", 1)
            os.environ[k] = v
        out = torch.zeros((th, tw, nch), device=dev, dtype=torch.float32)
        ms = ea.render_timed(args, [src], out.data_ptr(), reps, nch, 0, th, None)
        ms2 = ea.render_timed(args, [src], out.data_ptr(), reps, nch, 0, th, None)
        torch.cuda.synchronize()
        if ref is None:
            ref = out
            diff = 0
        else:
            diff = int((out.view(torch.int32) != ref.view(torch.int32)).sum().item())
        print(f"{wl:10s} [{s:40s}] kernel_ms {ms:.4f} {ms2:.4f}  differing floats vs first: {diff}", flush=True)
        if out is not ref:
            del out


if __name__ == "__main__":
    main()
