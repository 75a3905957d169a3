import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import envutil_amd as ea, jobs, euo
import test_gpu_fuzz as F
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1033; want = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rng = np.random.default_rng(9000 + seed)
for k in range(5):
    sprj, sw, sh, shfov, nch, degree, *_ = F.draw_job(rng)
    pdeg = int(rng.choice([degree, degree, 0, 1, 3, 5]))
    img = jobs.synth_image(sw, sh, nch, seed=seed * 77 + k)
    smin, tile = int(rng.choice([8, 8, 4, 12, 1])), int(rng.choice([64, 64, 16, 32]))
    yaw = float(rng.uniform(-180, 180)); pitch = float(rng.uniform(-60, 60))
    if k != want:
        continue
    print("job", k, "prj", sprj, sw, sh, "fov", shfov, "nch", nch, "degree", degree, "pdeg", pdeg, "smin", smin, "tile", tile, "yaw", yaw, "pitch", pitch)
    o = jobs.OracleSource(sprj, sw, sh, shfov, img, degree, pdeg, support_min=smin, tile=tile)
    g = ea.Source.load(ea.facet_spec(sprj, sw, sh, shfov, nchannels=nch), img, degree, pdeg, support_min=smin, tile_size=tile)
    same = (g.download().reshape(-1).view(np.uint32) == np.ascontiguousarray(o.container, np.float32).reshape(-1).view(np.uint32))
    print("container identical:", same.all())
    a = ea.arguments(ea.SPHERICAL, 96, 48, 360.0, yaw=yaw, pitch=pitch, spline_degree=degree)
    ref = jobs.oracle_render(a, o)
    for st in (1, 2):
        gs_, os_ = ea.render(a, g, stage=st), jobs.oracle_render(a, o, stage=st)
        badst = np.argwhere(gs_.view(np.uint32) != os_.view(np.uint32))
        print("stage", st, "differing floats:", len(badst), badst[:6].tolist())
    got0 = ea.render(a, g)
    bad0 = np.argwhere(got0.view(np.uint32) != ref.view(np.uint32))
    c2g, c2o = ea.render(a, g, stage=2), jobs.oracle_render(a, o, stage=2)
    print("metrics: container shape", o.container.shape)
    for i in sorted({(int(b[0]), int(b[1])) for b in bad0})[:16]:
        print("  pixel", i, "source coordinate gpu", c2g[i].tolist(), "oracle", c2o[i].tolist(), "pixel gpu", got0[i].tolist(), "oracle", ref[i].tolist())
    for envs in ({},):
        for kk in ("EU_HIP_COLMAJOR", "EU_HIP_R4", "EU_HIP_KERNEL"):
            os.environ.pop(kk, None)
        os.environ.update(envs)
        got = ea.render(a, g)
        bad = np.argwhere(got.view(np.uint32) != ref.view(np.uint32))
        print(envs, "differing floats:", len(bad), "first:", bad[:4].tolist(), "max ulp", int(jobs.ulp_diff(got, ref).max()) if len(bad) else 0)
        if len(bad):
            i = tuple(bad[0]); print("   gpu", got[i], "oracle", ref[i])
