"""Golden vectors produced by the REFERENCE ITSELF: the unmodified compute shader
res/shader/raytracer.glsl, executed in the build container by Mesa llvmpipe through
oracle/glsl_ref/run_reference_glsl.c (a headless GL 4.5 context on Mesa's software DRI driver).

Run in the build container only (needs /root/reference and Mesa's swrast_dri.so):
    python tests/golden/make_glsl_reference_vectors.py
Writes tests/golden/ref_*.npz; nothing of the reference's text is stored, only inputs and outputs:

  ref_builtins.npz   the GL implementation's elementary functions on 20000 arguments (the probe shaders under
                     oracle/glsl_ref/probes/ are this repository's own) and its bilinear REPEAT lookup
                     in two RGBA32F textures
  ref_hits_<scene>.npz   4096 rays, with the triangle id and (u, v, t) the reference's findIntersection
                     returns (harness = the reference's text up to findIntersection + a 10-line main())
  ref_image_<case>.npz   the image the reference's main() writes for single SAMPLE values on a cleared target
                     (for SAMPLE = S > 0 that is colour * 1/(S+1), raytracer.glsl:609-613), and for the
                     cubes case also the running mean over SAMPLE 0..15
  ref_irradiance.npz the (w/4) x (h/4) image the reference's res/shader/irradiance.glsl leaves after one dispatch
                     per radiance row (Scene.cpp:150-186), for three small random environments
                     (`python tests/golden/make_glsl_reference_vectors.py irradiance` makes this file alone)
  `... extra` makes the cases of make_extra() (EXPOSURE 0.5, texels at the edges of the packed-float upload, a scene of
                     denormal numbers)
  ref_c1_cubes512.json   the reference's own CPU-runnable configuration (cubes, 512 x 512, SAMPLE 0..15): SHA-256
                     of the RGB float32 bytes of each single-SAMPLE image and of the 16-sample running mean
                     (results are compared bit for bit, so a digest is enough), and llvmpipe's wall time
"""
import hashlib
import json
import time
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, ".."))
from conftest import constant_env, gradient_env, load_scene, textured_cubes, tiny_cubes  # noqa: E402
from gpuraytracer_amd.scene_io import write_rtscene  # noqa: E402
from gpuraytracer_amd.camera import camera_matrix, DEFAULT_CLOCK  # noqa: E402
from make_golden_vectors import ray_batch  # noqa: E402

RUNNER = os.path.join(ROOT, "oracle", "_ref", "run_reference_glsl")
PROBES = os.path.join(ROOT, "oracle", "glsl_ref", "probes")
TMP = os.path.join(ROOT, "oracle", "_ref", "tmp")


def run(*args):
    r = subprocess.run([RUNNER] + [str(a) for a in args], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("run_reference_glsl %s failed:\n%s" % (args[0], r.stderr))
    return r.stderr


def tmp(name):
    return os.path.join(TMP, name)


def scene_path(name):
    return os.path.join(HERE, "scenes", name + ".rtscene")


def ref_func(shader, x, n_items, per_item, with_reference_text=False, env=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    x.tofile(tmp("in.f32"))
    args = ["func", os.path.join(PROBES, shader), tmp("in.f32"), x.size, n_items, tmp("out.f32"), per_item,
            "ref" if with_reference_text else "own"]
    if env is not None:
        np.ascontiguousarray(env, dtype=np.float32).tofile(tmp("env.f32"))
        args += [tmp("env.f32"), env.shape[1], env.shape[0]]
    run(*args)
    return np.fromfile(tmp("out.f32"), dtype=np.float32).reshape(n_items, per_item)


def ref_hits(scene, o, d):
    n = len(o)
    np.concatenate([o, d], axis=1).astype(np.float32).tofile(tmp("rays.f32"))
    run("hits", scene if os.path.sep in scene else scene_path(scene), tmp("rays.f32"), n, tmp("hits.bin"))
    raw = np.fromfile(tmp("hits.bin"), dtype=np.uint8)
    return raw[:4 * n].view(np.int32).copy(), raw[4 * n:].view(np.float32).reshape(n, 3).copy()


def ref_image(scene, env, width, height, clock, recursion, first, count, cam, atlas=None, exposure=None):
    """scene: name of a bundled scene or path of an rtscene file; atlas: path of an atlas description."""
    if exposure is None:
        os.environ.pop("REF_EXPOSURE", None)
    else:
        os.environ["REF_EXPOSURE"] = repr(float(exposure))
    np.ascontiguousarray(env, dtype=np.float32).tofile(tmp("env.f32"))
    np.ascontiguousarray(cam, dtype=np.float32).tofile(tmp("cam.f32"))
    args = ["image", scene if os.path.sep in scene else scene_path(scene), tmp("env.f32"), env.shape[1], env.shape[0],
            width, height, clock, recursion, count, tmp("img.f32"), tmp("cam.f32"), first]
    if atlas:
        args.append(atlas)
    run(*args)
    return np.fromfile(tmp("img.f32"), dtype=np.float32).reshape(height, width, 4)[..., :3].copy()


def ref_irradiance(env):
    h, w = env.shape[:2]
    np.ascontiguousarray(env, dtype=np.float32).tofile(tmp("irr_env.f32"))
    run("irradiance", tmp("irr_env.f32"), w, h, tmp("irr_out.f32"))
    return np.fromfile(tmp("irr_out.f32"), dtype=np.float32).reshape(h // 4, w // 4, 3)


def make_irradiance():
    os.makedirs(TMP, exist_ok=True)
    out = {}
    for k, (w, h, seed) in enumerate(((64, 32, 5), (52, 28, 6), (100, 50, 7))):
        rng = np.random.default_rng(seed)
        env = rng.random((h, w, 4), dtype=np.float32) * 3.0
        env[..., 3] = 1.0
        out["env%d" % k] = env
        out["irr%d" % k] = ref_irradiance(env)
        print("irradiance %dx%d -> %dx%d" % (w, h, w // 4, h // 4))
    np.savez_compressed(os.path.join(HERE, "ref_irradiance.npz"), **out)


def write_atlas(path, size, uploads, flags):
    """uploads: list of (material, slot, rgba[h, w, 4])."""
    with open(path, "wb") as f:
        f.write(np.array([size, len(uploads)], dtype=np.uint32).tobytes())
        for material, slot, rgba in uploads:
            f.write(np.array([material, slot, rgba.shape[1], rgba.shape[0]], dtype=np.uint32).tobytes())
            f.write(np.ascontiguousarray(rgba, dtype=np.float32).tobytes())
        f.write(np.ascontiguousarray(flags, dtype=np.int32).tobytes())


# (case name, scene, environment, env name, width, height, RECURSION, camera translation/rotation, SAMPLE values)
IMAGE_CASES = [
    ("cubes", "cubes", constant_env(), 64, 64, 6, None, (0, 1, 2, 3, 7, 100)),
    ("cubes_wide", "cubes", constant_env(), 96, 40, 6, None, (0, 5)),          # ragged 8x8 tiles
    ("cubes_gradient", "cubes", gradient_env(), 64, 64, 6, None, (0, 5)),
    ("sphere_gradient_npot", "sphere", gradient_env(50, 30), 64, 64, 6, None, (0, 1)),
    ("box", "box", constant_env(), 64, 64, 6, None, (0, 1, 2)),
    ("obiwan", "_ObiWan", constant_env(), 64, 64, 16, None, (0, 1)),
    ("obiwan_close", "_ObiWan", gradient_env(), 48, 48, 16, ((0.0, 1.0, -1.5), (-0.5, 0.0)), (0, 3)),
]


def ref_write_tiny(path, scale):
    write_rtscene(path, *tiny_cubes(scale))


def make_extra():
    """Cases added after the first set (kept apart so that the first set's files need not be rewritten):
      ref_image_cubes_exposure.npz       EXPOSURE = 0.5 (Scene.cpp:331 always sets 1.0; the uniform scales the environment)
      ref_image_cubes_textured_edge.npz  a 128^2 atlas whose albedo layer carries values at the edges of the
                                         R11F_G11F_B10F upload conversion
      ref_hits_tiny.npz                  findIntersection on the cubes scene scaled by 2^-66, where the triangle normals
                                         (products of two edges) are denormal numbers: llvmpipe runs with denormals
                                         flushed to zero, so every plane test sees a zero normal"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    os.makedirs(TMP, exist_ok=True)
    cam = camera_matrix()
    env = gradient_env()
    out = {"env": env, "camera": cam, "clock": np.uint32(DEFAULT_CLOCK), "recursion": np.int32(6), "exposure": np.float32(0.5),
           "samples": np.array([0, 5], dtype=np.uint32), "scene": np.array("cubes")}
    for s in (0, 5):
        out["sample_%d" % s] = ref_image("cubes", env, 64, 64, DEFAULT_CLOCK, 6, s, 1, cam, exposure=0.5)
    np.savez_compressed(os.path.join(HERE, "ref_image_cubes_exposure.npz"), **out)
    print("image cubes_exposure          64x64 RECURSION 6 SAMPLE [0, 5] EXPOSURE 0.5")

    scene, layers, flags = textured_cubes(128, edge_texels=True)
    write_rtscene(tmp("textured.rtscene"), *scene)
    write_atlas(tmp("atlas.bin"), 128, [(2, k, layers[k]) for k in (0, 1, 2)], flags)
    out = {"env": env, "camera": cam, "clock": np.uint32(777), "recursion": np.int32(6),
           "samples": np.array([0, 1], dtype=np.uint32), "atlas_size": np.uint32(128)}
    for s in (0, 1):
        out["sample_%d" % s] = ref_image(tmp("textured.rtscene"), env, 48, 32, 777, 6, s, 1, cam, atlas=tmp("atlas.bin"))
    np.savez_compressed(os.path.join(HERE, "ref_image_cubes_textured_edge.npz"), **out)
    print("image cubes_textured_edge     48x32 RECURSION 6 SAMPLE [0, 1] atlas 128")

    models, shadings, lights, materials = load_scene("cubes")
    o0, d = ray_batch(models)
    out = {"d": d, "scales": np.array([2.0 ** -42, 2.0 ** -44], dtype=np.float32)}
    for k, scale in enumerate(out["scales"]):
        # the plane distance dot(normal, origin - corner) is of the order scale^3: around 2^-126, the smallest normal number,
        # for the first scale (some flushed, some not) and below it for the second
        o = (o0 * scale).astype(np.float32)
        ref_write_tiny(tmp("tiny.rtscene"), scale)
        ids, uvt = ref_hits(tmp("tiny.rtscene"), o, d)
        out["o_%d" % k], out["ids_%d" % k], out["uvt_%d" % k] = o, ids, uvt
        print("hits tiny (scale 2^%d)  %d of %d rays hit" % (int(np.log2(scale)), int((ids >= 0).sum()), len(ids)))
    np.savez_compressed(os.path.join(HERE, "ref_hits_tiny.npz"), **out)
    for f in os.listdir(TMP):
        os.remove(os.path.join(TMP, f))


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    os.makedirs(TMP, exist_ok=True)
    print(run("probe").strip())

    # --- elementary functions and the texture filter of the GL implementation
    rng = np.random.default_rng(20261111)
    x = np.concatenate([rng.uniform(0, 1, 8000), rng.uniform(0, 6.2832, 8000), rng.uniform(-1, 1, 3990),
                        [0.0, 1.0, -1.0, 0.5, 6.2831855, 1e-8, 0.99999994, -0.99999994, 2.0, -2.0]]).astype(np.float32)
    b1 = ref_func("builtins.glsl", x, len(x), 12)
    b2 = ref_func("builtins2.glsl", x, len(x), 4)
    ay = rng.uniform(-2, 2, 20000).astype(np.float32)
    ax = rng.uniform(-2, 2, 20000).astype(np.float32)
    ay[:8] = [0, 0, 1, -1, 1, -1, 0.5, -0.5]
    ax[:8] = [1, -1, 0, 0, 1, -1, 0.5, 0.5]
    at = ref_func("atan2.glsl", np.stack([ay, ax], axis=1), len(ay), 1)[:, 0]
    tex = {}
    for key, (h, w) in (("pot", (32, 64)), ("npot", (30, 50))):
        env = rng.uniform(0, 4, (h, w, 4)).astype(np.float32)
        st = rng.uniform(-1.5, 2.5, (20000, 2)).astype(np.float32)
        st[:64, 0] = np.linspace(0, 1, 64)
        st[:64, 1] = 0.0
        tex["tex_%s_env" % key] = env
        tex["tex_%s_st" % key] = st
        tex["tex_%s_rgb" % key] = ref_func("env_lookup.glsl", st, len(st), 3, env=env)
    # the conversion of an RGBA float upload to GL_R11F_G11F_B10F
    px = np.concatenate([rng.uniform(0, 2, 5000), 10 ** rng.uniform(-7, 5.2, 4988),
                         [0, -1, 1e-8, 6.1e-5, 6.0e-5, 3e-5, 65024, 65025, 70000, 1e9, np.inf, np.nan]]).astype(np.float32)
    rgba = np.stack([px, px, px, np.ones_like(px)], axis=1)
    rgba.tofile(tmp("pack.f32"))
    run("pack", tmp("pack.f32"), len(px), tmp("pack_out.f32"))
    tex["pack_x"] = px
    tex["pack_rgb"] = np.fromfile(tmp("pack_out.f32"), dtype=np.float32).reshape(len(px), 3)
    np.savez_compressed(os.path.join(HERE, "ref_builtins.npz"), x=x, sin=b1[:, 0], cos=b1[:, 1], pow5=b1[:, 2],
                        tan=b1[:, 3], acos=b1[:, 4], asin=b1[:, 5], atan2_x037=b1[:, 6], sqrt=b1[:, 7],
                        inversesqrt=b1[:, 8], rcp=b1[:, 9], log2=b2[:, 0], exp2=b2[:, 1], pow24=b2[:, 2],
                        pow5_1mx=b2[:, 3], atan2_y=ay, atan2_x=ax, atan2=at, **tex)
    print("builtins: %d arguments" % len(x))

    # --- findIntersection of the reference on fixed ray batches
    for name in ("cubes", "_ObiWan", "box", "sphere"):
        models, _, _, _ = load_scene(name)
        o, d = ray_batch(models)
        ids, uvt = ref_hits(name, o, d)
        np.savez_compressed(os.path.join(HERE, "ref_hits_%s.npz" % name), o=o, d=d, ids=ids, uvt=uvt)
        print("hits %-8s %d of %d rays hit" % (name, int((ids >= 0).sum()), len(ids)))

    # --- images of the reference's main()
    for case, scene, env, width, height, recursion, camera, samples in IMAGE_CASES:
        cam = camera_matrix() if camera is None else camera_matrix(*camera)
        out = {"env": env, "camera": cam, "clock": np.uint32(DEFAULT_CLOCK), "recursion": np.int32(recursion),
               "samples": np.array(samples, dtype=np.uint32), "scene": np.array(scene)}
        for s in samples:
            out["sample_%d" % s] = ref_image(scene, env, width, height, DEFAULT_CLOCK, recursion, s, 1, cam)
        if case == "cubes":
            out["mean_16"] = ref_image(scene, env, width, height, DEFAULT_CLOCK, recursion, 0, 16, cam)
        np.savez_compressed(os.path.join(HERE, "ref_image_%s.npz" % case), **out)
        print("image %-22s %dx%d RECURSION %d SAMPLE %s" % (case, width, height, recursion, list(samples)))

    # --- textured materials: the atlas path (three layers of material 2, LINEAR + REPEAT on R11F_G11F_B10F)
    scene, layers, flags = textured_cubes(32)
    write_rtscene(tmp("textured.rtscene"), *scene)
    write_atlas(tmp("atlas.bin"), 32, [(2, k, layers[k]) for k in (0, 1, 2)], flags)
    cam = camera_matrix()
    env = gradient_env()
    out = {"env": env, "camera": cam, "clock": np.uint32(4242), "recursion": np.int32(6),
           "samples": np.array([0, 1, 2, 3], dtype=np.uint32), "atlas_size": np.uint32(32)}
    for s in (0, 1, 2, 3):
        out["sample_%d" % s] = ref_image(tmp("textured.rtscene"), env, 48, 32, 4242, 6, s, 1, cam, atlas=tmp("atlas.bin"))
    np.savez_compressed(os.path.join(HERE, "ref_image_cubes_textured.npz"), **out)
    print("image cubes_textured         48x32 RECURSION 6 SAMPLE [0, 1, 2, 3]")

    make_irradiance()

    # --- the reference's CPU-runnable configuration, as digests
    cam = camera_matrix()
    env = constant_env()
    c1 = {"scene": "cubes", "width": 512, "height": 512, "clock": DEFAULT_CLOCK, "recursion": 6,
          "environment": "constant_env()", "samples": list(range(16)), "sha256_sample": []}
    for s in range(16):
        img = ref_image("cubes", env, 512, 512, DEFAULT_CLOCK, 6, s, 1, cam)
        c1["sha256_sample"].append(hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest())
    t0 = time.time()
    mean = ref_image("cubes", env, 512, 512, DEFAULT_CLOCK, 6, 0, 16, cam)
    c1["llvmpipe_seconds_16spp"] = round(time.time() - t0, 3)
    c1["llvmpipe_threads"] = os.cpu_count()
    c1["sha256_mean_16"] = hashlib.sha256(np.ascontiguousarray(mean).tobytes()).hexdigest()
    c1["mean_radiance_16"] = float(mean.mean())
    with open(os.path.join(HERE, "ref_c1_cubes512.json"), "w") as f:
        json.dump(c1, f, indent=1)
    print("C1 digests written; llvmpipe: %.2f s for 16 spp on %d threads" % (c1["llvmpipe_seconds_16spp"], os.cpu_count()))
    for f in os.listdir(TMP):
        os.remove(os.path.join(TMP, f))


if __name__ == "__main__":
    if sys.argv[1:] == ["irradiance"]:
        make_irradiance()
    elif sys.argv[1:] == ["extra"]:
        make_extra()
    else:
        main()
