import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import envutil_amd as ea
sw, sh = 16384, 8192
rng = np.random.default_rng(1)
img = rng.random((sh, sw, 3), dtype=np.float32)
fct = ea.facet_spec(ea.SPHERICAL, sw, sh, 360.0)
for cfg in sys.argv[1:]:
    r, l, exp, wpg = cfg.split(",")
    os.environ["EU_HIP_IIR_ROWS"] = r; os.environ["EU_HIP_IIR_COLS"] = l; os.environ["EU_HIP_IIR_EXP"] = exp
    os.environ["EU_HIP_IIR_WPG"] = wpg
    g = ea.Source.load(fct, img, 3); g.release()
    print("cfg", cfg, "done", flush=True)
