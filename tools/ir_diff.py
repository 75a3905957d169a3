"""where do the device-built and the oracle-built cubemap IR differ? (debug aid)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import envutil_amd as ea
import euo, jobs
face, nch, degree, pdeg = [int(v) for v in sys.argv[1:5]]
img = jobs.synth_image(face, 6 * face, nch, seed=2)
o = jobs.OracleSource(euo.CUBEMAP, face, 6 * face, 90.0, img, degree, pdeg)
g = ea.Source.load(ea.facet_spec(ea.CUBEMAP, face, 6 * face, 90.0, nchannels=nch), img, degree, pdeg)
m = ea.cubemap_metrics(face)
S = m["section_px"]
got = g.download().reshape(6 * S, S, nch)
ref = np.asarray(o.container, np.float32).reshape(6 * S, S, nch)
bad = np.argwhere((got.view(np.uint32) != ref.view(np.uint32)).any(axis=2))
print("metrics", m, "bad texels", len(bad))
if len(bad):
    ys, xs = bad[:, 0] % S, bad[:, 1]
    print("rows in section:", sorted(set(ys.tolist()))[:40])
    print("cols:", sorted(set(xs.tolist()))[:40])
    i = tuple(bad[0]); print("first", i, got[i], ref[i])
