"""Kernel time of every rank's strip of the headline frame, measured on ONE GPU:
predicts the max-over-ranks time of an N-GPU run (no 8-GPU node needed)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import envutil_amd as ea
import bench
from envutil_amd.distributed import row_partition

dev = torch.device("cuda:0")
ea.lib().eu_hip_init(0)
sw, sh, tw, th, nch, deg = 16384, 8192, 4096, 24576, 3, 3
img = bench.synth_on_device(torch, dev, sw, sh, nch).cpu().numpy()
src = ea.Source.load(ea.facet_spec(ea.SPHERICAL, sw, sh, 360.0), img, deg)
del img
args = ea.arguments(ea.CUBEMAP, tw, th, 90.0, spline_degree=deg)
out = torch.empty((th, tw, nch), device=dev, dtype=torch.float32)
full = ea.render_timed(args, src, out.data_ptr(), 10, nch)
res = {"full_ms": round(full, 4)}
for world in (2, 4, 8):
    ts = []
    for r in range(world):
        r0, r1 = row_partition(th, world, r, align=4)
        ts.append(round(ea.render_timed(args, src, out.data_ptr(), 10, nch, r0, r1), 4))
    res[f"n{world}"] = {"strip_ms": ts, "predicted_speedup": round(full / max(ts), 2)}
# the same with interleaved bands (what bench.py does for N > 1)
for world in (2, 4, 8):
    ts = []
    for r in range(world):
        band = (bench.BAND_ROWS, world, r)
        ts.append(round(ea.render_timed(args, src, out.data_ptr(), 10, nch, band=band), 4))
    res[f"n{world}_bands"] = {"part_ms": ts, "predicted_speedup": round(full / max(ts), 2)}
# contiguous strips of equal estimated cost (what bench.py uses when the library reports
# layout segments): the launch-level layout choice applies inside every strip
from envutil_amd.distributed import cost_partition
seg_rows, flags = ea.layout_segments(args, src, nch)
if flags.size and flags.any():
    for fc in [float(v) for v in os.environ.get("EU_FLAG_COSTS", "2.0").split(",")]:
        for world in (2, 4, 8):
            ts = [round(ea.render_timed(args, src, out.data_ptr(), 10, nch, r0, r1), 4)
                  for r0, r1 in cost_partition(th, world, seg_rows, flags, flag_cost=fc)]
            res[f"n{world}_cost_{fc}"] = {"strip_ms": ts, "predicted_speedup": round(full / max(ts), 2)}
faces = [round(ea.render_timed(args, src, out.data_ptr(), 10, nch, f * tw, (f + 1) * tw), 4) for f in range(6)]
res["face_ms"] = faces
print(json.dumps(res))
