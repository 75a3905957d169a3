import os, sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np
import envutil_amd as ea, euo, jobs
def load(fct, img, deg, stream, dbg=0):
    os.environ["EU_HIP_IIR_STREAM"] = str(stream); os.environ["EU_HIP_IIR_DEBUG"] = str(dbg)
    g = ea.Source.load(fct, img, deg); a = g.download(); g.release(); return a
for (sw, sh, nch, deg) in [(128, 64, 1, 3), (640, 320, 3, 3), (1000, 500, 3, 5)]:
    img = jobs.synth_image(sw, sh, nch, seed=21 + nch)
    fct = ea.facet_spec(ea.SPHERICAL, sw, sh, 360.0, nchannels=nch)
    b = load(fct, img, deg, 0)
    for mode in (2, 3):
        for dbg in (0, 1, 2, 3):
            a = load(fct, img, deg, mode, dbg)
            d = a.view(np.uint32) != b.view(np.uint32)
            print(sw, sh, nch, deg, "mode", mode, "dbg", dbg, "diff", int(d.sum()), flush=True)
