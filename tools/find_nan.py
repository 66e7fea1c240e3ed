"""Accounts for the non-finite pixels of a bench workload: finds (pixel, SAMPLE) pairs whose radiance is not finite on the
GPU, replays each in the CPU checker (both back ends) with its per-bounce trace switched on, and names the expression
in which the first NaN appears.
    python tools/find_nan.py --workload cubes|obiwan|c3 [--spp 2048] [--pairs 6]"""
import argparse
import os
import re
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpuraytracer_amd as g                     # noqa: E402
import bench                                     # noqa: E402

W, H = 1920, 1080
BLOCK = 64


def upload(target, scene, layers, atlas):
    if layers is not None:
        target.set_atlas_size(atlas)
        flags = np.zeros(len(scene[3]), dtype=np.int32)
        for m in (0, 1):
            for k in range(3):
                target.upload_atlas_layer(m, k, layers[k])
            flags[m] = 1
        target.set_has_texture(flags)
    target.upload_env(bench.constant_env())


def traced_replay(sc, cam, clock, recursion, x, y, s):
    """One pixel-sample in the checker with its trace (the checker prints one block per bounce to standard error)."""
    sys.stderr.flush()
    saved = os.dup(2)
    with tempfile.TemporaryFile() as f:
        os.dup2(f.fileno(), 2)
        try:
            rgb = sc.pixel_samples(cam, clock, recursion, 1.0, W, H, [x], [y], [s])[0]
        finally:
            os.dup2(saved, 2)
            os.close(saved)
        f.seek(0)
        text = f.read().decode(errors="replace")
    return rgb, text


def first_nan(text):
    """The first line of a trace that holds a NaN, and the expression of the shader it belongs to."""
    for line in text.splitlines():
        if not re.search(r"\bnan\b|-nan", line):
            continue
        if line.lstrip().startswith("fresnel:"):
            m = re.search(r"1 - cos\^2 = (\S+)\)", line)
            neg = m and float(m.group(1)) < 0
            return line.strip(), ("getTransmission, raytracer.glsl:204: sqrt(1 - cosThetaI * cosThetaI) of a rounded cosine above 1"
                                  if neg else "Fresnel term, raytracer.glsl:496-497")
        if line.lstrip().startswith("sky:"):
            return line.strip(), ("skyColor, raytracer.glsl:274: asin(direction.y) of a direction whose y is beyond 1 in magnitude (the probe "
                                  "direction of sampleLightGlossy, :401, is a reflect() of unit vectors and is not normalised again)")
        if line.lstrip().startswith("textured:"):
            return line.strip(), "tangent frame / normal map, raytracer.glsl:261-270, :471-479"
        return line.strip(), "first NaN on this line of the trace"
    return None, "no NaN in the trace"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=("cubes", "obiwan", "c3"), default="cubes")
    ap.add_argument("--spp", type=int, default=2048)
    ap.add_argument("--pairs", type=int, default=6)
    ap.add_argument("--grid", type=int, default=724)
    ap.add_argument("--atlas", type=int, default=4096)
    args = ap.parse_args()
    os.environ["ORC_DEBUG"] = "1"                # read once by the checker library, before its first bounce
    scene, cam, recursion, name, layers = bench.load_workload(g, args)
    clock = g.DEFAULT_CLOCK
    print("workload: %s, %dx%d, RECURSION %d, SAMPLE 0..%d" % (name, W, H, recursion, args.spp - 1))
    ctx = g.Context(0)
    ctx.upload_scene(*scene)
    upload(ctx, scene, layers, args.atlas)
    ctx.set_option("batch_samples", BLOCK)
    ctx.set_frame(cam, clock, recursion, 1.0)
    found, bad_pixels, blocks = [], 0, 0
    for s0 in range(0, args.spp, BLOCK):
        ctx.resize(W, H)
        ctx.trace(s0, BLOCK, 1)
        img, _ = ctx.read_sum()
        bad = ~np.isfinite(img).all(axis=2)
        bad_pixels += int(bad.sum())
        blocks += 1
        if not bad.any() or len(found) >= args.pairs:
            continue
        ys, xs = np.nonzero(bad)
        for s in range(s0, s0 + BLOCK):          # which SAMPLE of the block it was
            ctx.resize(W, H)
            ctx.trace(s, 1, 1)
            one, _ = ctx.read_sum()
            for x, y in zip(xs, ys):
                if not np.isfinite(one[y, x]).all():
                    found.append((int(x), int(y), s, one[y, x].copy()))
    print("non-finite (pixel, block of %d samples) pairs on the GPU: %d in %.3g pixel-samples" % (BLOCK, bad_pixels, float(W) * H * blocks * BLOCK))
    ctx.close()
    from oracle import oracle
    backends = []
    for libm in (False, True):
        sc = oracle.OracleScene(*scene, libm=libm)
        upload(sc, scene, layers, args.atlas)
        backends.append(sc)
    same = 0
    for x, y, s, gpu_rgb in found[:args.pairs]:
        rgb, text = traced_replay(backends[0], cam, clock, recursion, x, y, s)
        rgb_libm, _ = traced_replay(backends[1], cam, clock, recursion, x, y, s)
        line, where = first_nan(text)
        agrees = bool(((rgb.view(np.uint32) == gpu_rgb.view(np.uint32)) | (np.isnan(rgb) & np.isnan(gpu_rgb))).all())
        same += agrees
        print("\npixel (%d, %d) SAMPLE %d" % (x, y, s))
        print("  GPU              %s" % gpu_rgb)
        print("  checker          %s   %s" % (rgb, "same bits as the GPU" if agrees else "DIFFERS FROM THE GPU"))
        print("  checker, C libm  %s" % rgb_libm)
        print("  first NaN: %s" % where)
        print("    %s" % line)
        bounces = [l for l in text.splitlines() if l.startswith("depth ")]
        print("  bounces traced: %d" % len(bounces))
    print("\n%d of %d replayed pixel-samples are non-finite in the checker with the GPU's bits" % (same, len(found[:args.pairs])))


if __name__ == "__main__":
    main()
