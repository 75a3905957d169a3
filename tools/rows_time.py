"""kernel time of row ranges of the headline frame (env switches apply): python tools/rows_time.py r0:r1 ..."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import envutil_amd as ea
import bench
dev = torch.device("cuda:0")
ea.lib().eu_hip_init(0)
if os.environ.get("EU_WORKLOAD") == "config3":      # 6x2048 cubemap -> 16384x8192 spherical, cubic
    sw, sh, tw, th, nch, deg = 2048, 12288, 16384, 8192, 3, 3
    img = bench.synth_on_device(torch, dev, sw, sh, nch).cpu().numpy()
    src = ea.Source.load(ea.facet_spec(ea.CUBEMAP, sw, sh, 90.0), img, deg)
    args = ea.arguments(ea.SPHERICAL, tw, th, 360.0, spline_degree=deg)
else:
    sw, sh, tw, th, nch, deg = 16384, 8192, 4096, 24576, 3, 3
    img = bench.synth_on_device(torch, dev, sw, sh, nch).cpu().numpy()
    src = ea.Source.load(ea.facet_spec(ea.SPHERICAL, sw, sh, 360.0), img, deg)
    args = ea.arguments(ea.CUBEMAP, tw, th, 90.0, spline_degree=deg)
del img
out = torch.empty((th, tw, nch), device=dev, dtype=torch.float32)
res = {}
for spec in sys.argv[1:]:
    r0, r1 = (int(v) for v in spec.split(":"))
    ea.render_timed(args, src, out.data_ptr(), 3, nch, r0, r1)
    res[spec] = round(ea.render_timed(args, src, out.data_ptr(), 20, nch, r0, r1), 4)
print(json.dumps(res))
