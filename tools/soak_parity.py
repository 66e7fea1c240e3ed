"""Soak run of the parity check on a GPU: random triangle soups of random size (the generator of the GPU tests), random
cameras, clocks, depths and target sizes; the library's default forms against the loops over all triangles (the
reference's own algorithm) and two other staged forms, bit for bit with equal counters; every fourth scene also against
the CPU checker.  Prints one line per scene and a summary; exit status 1 if anything differed.
    python tools/soak_parity.py [first seed] [number of scenes] [seed step]
Seeds from 10000 on add further families by seed % 5: 0 a bundled scene, 1 a camera matrix with exact zeros, 2 a soup of
30 000 to 160 000 triangles."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpuraytracer_amd as g                     # noqa: E402
from gpuraytracer_amd import host                # noqa: E402
from gpuraytracer_amd.camera import camera_matrix   # noqa: E402
from conftest import gradient_env                # noqa: E402


def case_for(seed, tmp, write_soup=None):
    """Everything that defines the scene and the render of one seed.  `write_soup`: the soup generator of the GPU
    tests (the tests pass their own, so that this file does not import them a second time)."""
    if write_soup is None:
        import test_gpu_parity
        write_soup = test_gpu_parity._write_soup
    rng = np.random.default_rng(seed)
    n_tris = int(10 ** rng.uniform(1.7, 4.5))
    big = seed >= 10000 and seed % 5 == 2      # sizes at which the hierarchy is built by several threads, has leaves
    if big:                                    # of two triangles, and the closest-hit kernels refill their lanes
        n_tris = int(10 ** np.random.default_rng([seed, 2]).uniform(4.5, 5.2))
    base = os.path.join(tmp, "soup%d" % seed)
    write_soup(base, seed, n_tris)
    scene = host.load_wavefront(base)
    cam = camera_matrix(tuple(rng.uniform([-1, -0.5, -6], [1, 2, -3])), tuple(rng.uniform([-0.4, -0.3], [0.2, 0.3])))
    w, h = int(rng.integers(17, 400)), int(rng.integers(9, 300))
    spp, rec, clock = int(rng.integers(1, 7)), int(rng.integers(1, 20)), int(rng.integers(1, 2 ** 31))
    batches = [int(rng.integers(1, 5)) for _ in range(4)]
    if seed >= 10000:
        # further families (seeds below 10000 keep the cases they always had): the bundled scenes, which are small
        # enough for the hierarchy to be resident in LDS, and cameras with exact zeros in the matrix
        extra = np.random.default_rng([seed, 1])
        family = seed % 5
        if family == 0:
            from conftest import load_scene
            scene = load_scene(["cubes", "box", "sphere", "_ObiWan"][int(extra.integers(0, 4))])
            cam = camera_matrix(tuple(np.array([0.0, 1.5, -3.0]) + extra.uniform(-1, 1, 3)),
                                tuple(np.array([-np.pi / 4, 0.0]) + extra.uniform(-0.4, 0.4, 2)))
        elif family == 2:
            w, h, spp = min(w, 120), min(h, 90), min(spp, 2)
        elif family == 1:
            cam = camera_matrix(tuple(float(v) for v in extra.integers(-2, 3, 3) + np.array([0, 0, -4])),
                                (float(extra.choice([0.0, 0.0, -0.25])), 0.0))
    return scene, cam, w, h, spp, rec, clock, batches


def check_seed(ctx, seed, tmp, env, T, oracle=None, oracle_budget=4e9):
    """Renders the case of one seed in the loop form and in five hierarchy forms (and with `oracle`, the CPU checker's
    module, every fourth seed by the checker too, while triangles x pixel-samples stays under `oracle_budget`).
    Returns (description, list of differences)."""
    scene, cam, w, h, spp, rec, clock, batches = case_for(seed, tmp, T._write_soup)
    T._upload(ctx, scene, env)
    want, wst = T._render(ctx, w, h, 0, spp, recursion=rec, clock=clock, pipeline=3, batch=batches[0], cam=cam)
    line = "seed %d: %d tris %dx%d spp %d rec %d rays %d" % (seed, len(scene[0]), w, h, spp, rec, wst["rays"])
    diffs = []
    for k, form in enumerate((1, 5, 7, 6, 9)):
        # queues for the whole pass, or for a fraction of it (passes that do not fit take further rounds)
        rounds = (1, 3, 8, 2)[(seed + k) % 4]
        img, st = T._render(ctx, w, h, 0, spp, recursion=rec, clock=clock, pipeline=form, batch=batches[k % 3 + 1], cam=cam,
                            rounds=rounds)
        same = (img.view(np.uint32) == want.view(np.uint32)) | (np.isnan(img) & np.isnan(want))
        ok = same.all() and all(st[key] == wst[key] for key in ("rays", "shadow_probes", "glossy_probes"))
        if not ok:
            diffs.append("FORM %d DIFFERS (%d values)" % (form, int((~same).sum())))
    if oracle is not None and seed % 4 == 0 and len(scene[0]) * w * h * spp < oracle_budget:
        sc = oracle.OracleScene(*scene)
        sc.upload_env(env)
        ref, rst = sc.render(cam, clock, rec, 1.0, w, h, 0, spp)
        sc.close()
        same = (ref.view(np.uint32) == want.view(np.uint32)) | (np.isnan(ref) & np.isnan(want))
        if not (same.all() and rst["rays"] == wst["rays"]):
            diffs.append("CHECKER DIFFERS (%d values)" % int((~same).sum()))
        else:
            line += "  checker ok"
    return line, diffs


def main():
    import test_gpu_parity as T
    from oracle import oracle
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    step = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    ctx = g.Context(0)
    bad = 0
    tmp = tempfile.mkdtemp()
    env = gradient_env()
    for seed in range(first, first + count * step, step):
        line, diffs = check_seed(ctx, seed, tmp, env, T, oracle)
        bad += len(diffs)
        print(line + "".join("  " + d for d in diffs), flush=True)
    print("%d scenes from seed %d (step %d), %d differences" % (count, first, step, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
