"""Finds the first (pixel, SAMPLE) whose radiance is not finite in the bench workload and prints it."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpuraytracer_amd as g

scene = g.read_rtscene(os.path.join(ROOT, "tests", "golden", "scenes", "cubes.rtscene"))
env = np.zeros((4, 8, 4), np.float32); env[..., :3] = (0.5, 0.7, 1.0)
ctx = g.Context(0)
ctx.upload_scene(*scene); ctx.upload_env(env)
ctx.set_frame(g.camera_matrix(), g.DEFAULT_CLOCK, 6, 1.0)
W, H = 1920, 1080
found = []
for s0 in range(0, 1024, 64):
    ctx.resize(W, H); ctx.trace(s0, 64, 1)
    img, _ = ctx.read_sum()
    bad = ~np.isfinite(img).all(axis=2)
    if bad.any():
        ys, xs = np.nonzero(bad)
        print("block", s0, "bad pixels", len(ys), list(zip(xs[:5], ys[:5])))
        # bisect the sample
        for s in range(s0, s0 + 64):
            ctx.resize(W, H); ctx.trace(s, 1, 1)
            one, _ = ctx.read_sum()
            b = ~np.isfinite(one).all(axis=2)
            if b.any():
                yy, xx = np.nonzero(b)
                for x, y in zip(xx, yy):
                    found.append((int(x), int(y), s))
        break
print("found", found[:10])
if found:
    from oracle import oracle
    sc = oracle.OracleScene(*scene); sc.upload_env(env)
    x, y, s = found[0]
    print("oracle:", sc.pixel_samples(g.camera_matrix(), g.DEFAULT_CLOCK, 6, 1.0, W, H, [x], [y], [s]))
    sl = oracle.OracleScene(*scene, libm=True); sl.upload_env(env)
    print("oracle libm:", sl.pixel_samples(g.camera_matrix(), g.DEFAULT_CLOCK, 6, 1.0, W, H, [x], [y], [s]))
