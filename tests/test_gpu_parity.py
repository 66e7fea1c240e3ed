"""GPU parity tests: the HIP path, called through the C ABI, against
  * vectors made by the reference's own shader run under Mesa llvmpipe (tests/golden/ref_*.npz):
    hit ids, (u, v, t) and whole images, bit for bit;
  * the CPU checker on seeded inputs: bit-exact ids, (u, v, t), ray counters and radiance sums (the
    device and the checker evaluate the same IEEE operation sequence), and a relative-RMSE bound
    against the checker's independent C-library back end."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, constant_env, gradient_env, load_scene, textured_cubes
from gpuraytracer_amd.camera import camera_matrix, DEFAULT_CLOCK

pytestmark = pytest.mark.gpu

CAM = camera_matrix()


def _upload(ctx, scene, env):
    models, shadings, lights, materials = scene
    ctx.upload_scene(models, shadings, lights, materials)
    ctx.upload_env(env)


# Forms of the same computation (`pipeline` argument of _render); every one must give the same bits.
#   0 single kernel, loops over all triangles         1 staged, library defaults (hierarchy, resident in LDS when it fits,
#   2 staged, every bounce staged (no tail kernel)      shared any-hit searches, two staged bounces then the tail kernel)
#   3 staged, loops over all triangles                4 staged, closest-hit kernels with lane refill (1 turns it on by
#                                                       itself from 65536 triangles on, the other forms keep it off)
#   5 staged, tail kernel right after depth 0         6 staged, hierarchy never resident in LDS
#   7 staged, every lane searches alone (no sharing)  8 staged, loops over all triangles, every bounce staged
#   9 staged, library defaults but camera rays through the hierarchy (no candidate lists per tile)
FORMS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]


def _render(ctx, w, h, begin, count, stride=1, recursion=6, clock=DEFAULT_CLOCK, pipeline=1, batch=4, cam=CAM, exposure=1.0,
            rounds=3):
    ctx.set_option("queue_rounds", rounds)
    ctx.set_option("accel", 0 if pipeline in (3, 8) else 1)
    ctx.set_option("refill", 1 if pipeline == 4 else (-1 if pipeline in (1, 9) else 0))
    ctx.set_option("tile_lists", 0 if pipeline == 9 else -1)
    ctx.set_option("staged_depths", -1 if pipeline in (2, 8) else (0 if pipeline == 5 else 2))
    ctx.set_option("resident", 0 if pipeline == 6 else -1)
    ctx.set_option("share", 0 if pipeline == 7 else 1)
    ctx.set_option("pipeline", min(pipeline, 1))
    ctx.set_option("batch_samples", batch)
    ctx.resize(w, h)
    ctx.set_frame(cam, clock, recursion, exposure)
    ctx.trace(begin, count, stride)
    img, n = ctx.read_sum()
    assert n == count
    return img, ctx.stats()


@pytest.mark.parametrize("accel", [0, 1])
@pytest.mark.parametrize("name", ["cubes", "_ObiWan", "box", "sphere"])
def test_closest_hit_ids_bit_exact(gpu_ctx, oracle_mod, name, accel):
    # ids and (u, v, t) returned by the reference's findIntersection
    g = np.load(os.path.join(GOLDEN, "ref_hits_%s.npz" % name))
    scene = load_scene(name)
    _upload(gpu_ctx, scene, constant_env())
    gpu_ctx.set_option("accel", accel)
    ids, uvt = gpu_ctx.closest_hit_batch(g["o"], g["d"])
    assert np.array_equal(ids, g["ids"])
    hit = ids >= 0
    assert np.array_equal(uvt[hit].view(np.uint32), g["uvt"][hit].view(np.uint32))
    # and against the checker run now, on a second seeded batch
    rng = np.random.default_rng(11)
    o = rng.uniform(-4, 4, size=(3000, 3)).astype(np.float32)
    d = rng.normal(size=(3000, 3)).astype(np.float32)
    sc = oracle_mod.OracleScene(*scene)
    want_ids, want_uvt = sc.closest_hit_batch(o, d)
    ids, uvt = gpu_ctx.closest_hit_batch(o, d)
    assert np.array_equal(ids, want_ids) and np.array_equal(uvt.view(np.uint32), want_uvt.view(np.uint32))


@pytest.mark.parametrize("accel", [0, 1])
def test_closest_hit_edge_cases(gpu_ctx, oracle_mod, cubes, accel):
    _upload(gpu_ctx, cubes, constant_env())
    gpu_ctx.set_option("accel", accel)
    sc = oracle_mod.OracleScene(*cubes)
    # empty batch, one ray, a batch that is not a multiple of the block size, NaN / zero directions
    ids, uvt = gpu_ctx.closest_hit_batch(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert len(ids) == 0
    o = np.array([[0, 1.5, -3], [np.nan] * 3, [0, 0, 0], [0, 1.5, -3], [np.nan, 0, 0]], np.float32)
    d = np.array([[0, -0.7, 0.7], [np.nan] * 3, [0, 0, 0], [np.inf, 0, 0], [0, 0, 1]], np.float32)
    ids, uvt = gpu_ctx.closest_hit_batch(o, d)
    want_ids, want_uvt = sc.closest_hit_batch(o, d)
    assert np.array_equal(ids, want_ids)
    assert np.array_equal(uvt, want_uvt, equal_nan=True)
    assert ids[1] == len(cubes[0]) - 1          # an all-NaN ray passes every rejection: last triangle wins


@pytest.mark.parametrize("name", ["cubes", "_ObiWan", "sphere"])
def test_hierarchy_equals_index_order_loop(gpu_ctx, name):
    # 400k rays: camera-like, random, and aimed exactly at vertices / edge midpoints / coincident
    # points, where three or more triangles tie in depth and the later index has to win
    models, shadings, lights, materials = load_scene(name)
    _upload(gpu_ctx, (models, shadings, lights, materials), constant_env())
    rng = np.random.default_rng(99)
    n = 400000
    p0 = models["position"]
    verts = np.concatenate([p0, p0 + models["u"], p0 + models["v"]]).astype(np.float64)
    lo, hi = verts.min(0), verts.max(0)
    o = rng.uniform(lo - 1, hi + 1, size=(n, 3))
    d = rng.normal(size=(n, 3))
    k = n // 2
    tri = rng.integers(0, len(models), size=k)
    a, b, c = p0[tri].astype(np.float64), (p0 + models["u"])[tri].astype(np.float64), (p0 + models["v"])[tri].astype(np.float64)
    w = rng.integers(0, 5, size=k)[:, None]
    target = np.where(w == 0, a, np.where(w == 1, b, np.where(w == 2, c, np.where(w == 3, 0.5 * (a + b), (a + b + c) / 3))))
    d[:k] = target - o[:k]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o32, d32 = o.astype(np.float32), d.astype(np.float32)
    gpu_ctx.set_option("accel", 0)
    ids0, uvt0 = gpu_ctx.closest_hit_batch(o32, d32)
    gpu_ctx.set_option("accel", 1)
    ids1, uvt1 = gpu_ctx.closest_hit_batch(o32, d32)
    assert np.array_equal(ids0, ids1), int((ids0 != ids1).sum())
    assert np.array_equal(uvt0.view(np.uint32), uvt1.view(np.uint32))
    assert (ids0 >= 0).mean() > 0.3


REF_IMAGE_CASES = ["cubes", "cubes_wide", "cubes_gradient", "sphere_gradient_npot", "box", "obiwan", "obiwan_close", "cubes_exposure"]


@pytest.mark.parametrize("pipeline", FORMS)
@pytest.mark.parametrize("case", REF_IMAGE_CASES)
def test_images_equal_reference_shader(gpu_ctx, case, pipeline):
    """Images the reference's main() wrote for single SAMPLE values, bit for bit.  The reference stores
    colour * 1/(SAMPLE+1) on a cleared target (raytracer.glsl:609-613); this library returns the colour."""
    if pipeline in (0, 3, 8) and case.startswith("obiwan"):
        pytest.skip("loops over all 12030 triangles are covered by the smaller scenes")
    g = np.load(os.path.join(GOLDEN, "ref_image_%s.npz" % case))
    _upload(gpu_ctx, load_scene(str(g["scene"])), g["env"])
    for s in g["samples"].tolist():
        want = g["sample_%d" % s]
        h, w = want.shape[:2]
        colour, _ = _render(gpu_ctx, w, h, s, 1, recursion=int(g["recursion"]), clock=int(g["clock"]),
                            pipeline=pipeline, cam=g["camera"], exposure=float(g["exposure"]) if "exposure" in g else 1.0)
        got = colour
        if s > 0:
            inv = np.float32(1.0) / np.float32(s + 1)
            got = (np.float32(0.0) * (np.float32(s) * inv) + inv * colour).astype(np.float32)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), "%s SAMPLE %d pipeline %d: %d pixels differ" % (case, s, pipeline, int((~same.all(axis=2)).sum()))


def test_c1_digests_equal_reference_shader(gpu_ctx):
    """cubes, 512 x 512, SAMPLE 0..15 (the reference's own CPU-runnable configuration) at full size against the
    SHA-256 digests of the images the reference's shader wrote for each SAMPLE."""
    import hashlib
    import json
    with open(os.path.join(GOLDEN, "ref_c1_cubes512.json")) as f:
        c1 = json.load(f)
    _upload(gpu_ctx, load_scene(c1["scene"]), constant_env())
    for s in c1["samples"]:
        colour, _ = _render(gpu_ctx, c1["width"], c1["height"], s, 1, recursion=c1["recursion"], clock=c1["clock"])
        got = colour
        if s > 0:
            inv = np.float32(1.0) / np.float32(s + 1)
            got = (np.float32(0.0) * (np.float32(s) * inv) + inv * colour).astype(np.float32)
        assert hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest() == c1["sha256_sample"][s], s


def test_mean_equals_reference_running_mean(gpu_ctx):
    # 16 samples: sum / 16 against the reference's running mean; the two accumulate in different orders
    g = np.load(os.path.join(GOLDEN, "ref_image_cubes.npz"))
    _upload(gpu_ctx, load_scene("cubes"), g["env"])
    want = g["mean_16"]
    h, w = want.shape[:2]
    total, _ = _render(gpu_ctx, w, h, 0, 16, recursion=int(g["recursion"]), clock=int(g["clock"]), cam=g["camera"])
    assert np.allclose(total / 16.0, want, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("pipeline", FORMS)
def test_golden_images_bit_exact(gpu_ctx, cubes, pipeline):
    _upload(gpu_ctx, cubes, constant_env())
    for spp in (1, 4, 16):
        g = np.load(os.path.join(GOLDEN, "image_cubes_%dspp.npz" % spp))
        img, st = _render(gpu_ctx, 64, 64, 0, spp, pipeline=pipeline)
        assert np.array_equal(img, g["rgb_sum"]), (pipeline, spp, float(np.abs(img - g["rgb_sum"]).max()))
        assert [st["pixel_samples"], st["rays"], st["shadow_probes"], st["glossy_probes"]] == g["stats"].tolist()


@pytest.mark.parametrize("pipeline", FORMS)
def test_image_vs_checker_gradient_env_ragged(gpu_ctx, oracle_mod, cubes, pipeline):
    # 52 x 37 is not a multiple of the 8 x 8 tile; the gradient environment exercises the bilinear lookup
    _upload(gpu_ctx, cubes, gradient_env())
    sc = oracle_mod.OracleScene(*cubes)
    sc.upload_env(gradient_env())
    cam = camera_matrix((0.3, 1.2, -2.5), (-0.5, 0.4))
    want, wst = sc.render(cam, 12345, 6, 1.0, 52, 37, 0, 6)
    img, st = _render(gpu_ctx, 52, 37, 0, 6, clock=12345, pipeline=pipeline, batch=4, cam=cam)
    assert np.array_equal(img, want), float(np.abs(img - want).max())
    for k in ("pixel_samples", "rays", "shadow_probes", "glossy_probes"):
        assert st[k] == wst[k], k
    # tolerance against the independent C-library back end: relative RMSE of the 6-spp image
    sl = oracle_mod.OracleScene(*cubes, libm=True)
    sl.upload_env(gradient_env())
    ref, _ = sl.render(cam, 12345, 6, 1.0, 52, 37, 0, 6)
    assert np.sqrt(np.mean((img - ref) ** 2)) / np.mean(ref) < 0.02


def test_full_size_forms_agree(gpu_ctx, oracle_mod, cubes):
    # At the size of the headline measurement (1920 x 1080): the staged pipeline (hierarchy, batches of 3 and of 8
    # samples) and the single kernel that loops over all triangles give the same sums bit for bit and the same
    # counters over 8 samples, and one whole sample agrees with the checker bit for bit.
    _upload(gpu_ctx, cubes, constant_env())
    a, sa = _render(gpu_ctx, 1920, 1080, 0, 8, pipeline=1, batch=3)
    b, sb = _render(gpu_ctx, 1920, 1080, 0, 8, pipeline=0)
    c, sc_ = _render(gpu_ctx, 1920, 1080, 0, 8, pipeline=1, batch=8)
    assert np.array_equal(a, b) and np.array_equal(a, c)
    for k in ("pixel_samples", "rays", "shadow_probes", "glossy_probes"):
        assert sa[k] == sb[k] == sc_[k], k
    assert sa["pixel_samples"] == 1920 * 1080 * 8
    # the staged form leaves the probes of sparse depth-0 chunks to the probe kernels, the single kernel has no queues
    assert sa["deferred_probes"] > 0 and sb["deferred_probes"] == 0
    sc = oracle_mod.OracleScene(*cubes)
    sc.upload_env(constant_env())
    want, _ = sc.render(CAM, DEFAULT_CLOCK, 6, 1.0, 1920, 1080, 0, 1)
    one, _ = _render(gpu_ctx, 1920, 1080, 0, 1)
    assert np.array_equal(one, want)


@pytest.mark.parametrize("name", ["cubes", "box"])
def test_queue_rounds_agree(gpu_ctx, name):
    # The queue regions hold 1/queue_rounds of what a pass may have to hand on, and a pass whose survivors do not fit
    # takes further rounds of the launch sequence.  At 1920 x 1080 with 8 samples a wave's share is 64 chunks; with 8
    # rounds a region holds 9 chunks' worth of paths, fewer than survive (checked below), so the later rounds have
    # work to do: same sums and counters as with queues for the whole pass, bit for bit.  `box` is a closed room.
    scene = load_scene(name)
    _upload(gpu_ctx, scene, constant_env())
    for pipeline in (1, 4, 5):
        want, wst = _render(gpu_ctx, 1920, 1080, 0, 8, pipeline=pipeline, batch=8, rounds=1)
        for rounds in (2, 8):
            img, st = _render(gpu_ctx, 1920, 1080, 0, 8, pipeline=pipeline, batch=8, rounds=rounds)
            assert np.array_equal(img, want, equal_nan=True), (pipeline, rounds)
            for k in ("pixel_samples", "rays", "shadow_probes", "glossy_probes", "depth1_paths"):
                assert st[k] == wst[k], (k, pipeline, rounds)
            assert st["queue_bytes"] < wst["queue_bytes"]
        chunks_per_wave = -(-(240 * 135 * 8) // 4096)
        assert st["depth1_paths"] > 4096 * (-(-chunks_per_wave // 8) + 1) * 64      # one round of 8 could not hold them


def test_obiwan_depth16_bit_exact(gpu_ctx, oracle_mod, obiwan):
    # glass (Ni 1.474) and rough metals, RECURSION 16: Fresnel, metallic and roulette branches
    _upload(gpu_ctx, obiwan, gradient_env())
    sc = oracle_mod.OracleScene(*obiwan)
    sc.upload_env(gradient_env())
    cam = camera_matrix((0.0, 1.0, -2.0), (-0.3, 0.0))
    want, wst = sc.render(cam, DEFAULT_CLOCK, 16, 1.0, 32, 24, 0, 3)
    for pipeline in FORMS:
        img, st = _render(gpu_ctx, 32, 24, 0, 3, recursion=16, pipeline=pipeline, batch=2, cam=cam)
        assert np.array_equal(img, want, equal_nan=True), pipeline
        assert st["rays"] == wst["rays"] and st["glossy_probes"] == wst["glossy_probes"]


def test_call_sequences(oracle_mod, cubes, obiwan):
    # State carried between calls: many small passes back to back without synchronising (the per-pass tables are
    # staged through two pinned slots), a change of resolution, of scene and of frame parameters in between, and
    # two contexts used in turn.
    import gpuraytracer_amd as g
    env = constant_env()
    oc = oracle_mod.OracleScene(*cubes)
    oc.upload_env(env)
    oo = oracle_mod.OracleScene(*obiwan)
    oo.upload_env(env)
    a, b = g.Context(0), g.Context(0)
    try:
        for ctx in (a, b):
            ctx.set_option("batch_samples", 1)
        a.upload_scene(*cubes); a.upload_env(env)
        b.upload_scene(*obiwan); b.upload_env(env)
        a.resize(40, 24); b.resize(24, 40)
        a.set_frame(CAM, DEFAULT_CLOCK, 6, 1.0)
        b.set_frame(CAM, 4242, 9, 1.0)
        for k in range(12):                      # 12 passes of one sample each, alternating contexts, no sync
            a.trace(k, 1, 1)
            b.trace(2 * k, 1, 2)
        img_a, n_a = a.read_sum()
        img_b, n_b = b.read_sum()
        want_a, _ = oc.render(CAM, DEFAULT_CLOCK, 6, 1.0, 40, 24, 0, 12)
        want_b, _ = oo.render(CAM, 4242, 9, 1.0, 24, 40, 0, 12, 2)
        assert n_a == 12 and n_b == 12
        assert np.array_equal(img_a, want_a) and np.array_equal(img_b, want_b, equal_nan=True)
        # new size, new frame, larger passes on the same context; then another scene on it
        a.set_option("batch_samples", 5)
        a.resize(56, 16)
        a.set_frame(CAM, 99, 4, 1.0)
        a.trace(3, 7, 1)
        img, n = a.read_sum()
        want, _ = oc.render(CAM, 99, 4, 1.0, 56, 16, 3, 7)
        assert n == 7 and np.array_equal(img, want)
        a.upload_scene(*obiwan)
        a.resize(56, 16)
        a.trace(0, 3, 1)
        img, n = a.read_sum()
        want, _ = oo.render(CAM, 99, 4, 1.0, 56, 16, 0, 3)
        assert n == 3 and np.array_equal(img, want, equal_nan=True)
    finally:
        a.close()
        b.close()


def test_trace_only_enqueues_and_can_be_captured(oracle_mod, cubes):
    # rt_trace reads nothing back and copies nothing: (1) it returns while its kernels are still running, (2) once the
    # queues have been sized by a first call, a call can be captured in a HIP graph and replayed
    import ctypes
    import gpuraytracer_amd as g
    from conftest import HipRuntime
    hip = HipRuntime()
    ctx = g.Context(0)
    stream = hip.stream()
    try:
        ctx.set_stream(stream.value)
        ctx.upload_scene(*cubes)
        ctx.upload_env(constant_env())
        ctx.set_option("batch_samples", 32)
        ctx.resize(1920, 1080)
        ctx.set_frame(CAM, DEFAULT_CLOCK, 6, 1.0)
        ctx.trace(0, 32, 1)                      # sizes the queues
        ctx.synchronize()
        done = hip.event()
        ctx.trace(32, 96, 1)                     # three passes of 32 samples at 1080p: tens of milliseconds of device work
        hip.check(hip.lib.hipEventRecord(done, stream))
        assert hip.lib.hipEventQuery(done) == hip.NOT_READY, "rt_trace waited for the device"
        ctx.synchronize()
        assert hip.lib.hipEventQuery(done) == 0
        hip.lib.hipEventDestroy(done)
        # capture: 6 samples in passes of 4 + 2 on a small target, replayed twice
        ctx.set_option("batch_samples", 4)
        ctx.resize(64, 40)
        ctx.trace(0, 6, 1)
        want, _ = ctx.read_sum()
        ctx.trace(0, 6, 1)                       # the same samples once more on top (what a second replay does)
        want_twice, _ = ctx.read_sum()
        sc = oracle_mod.OracleScene(*cubes)
        sc.upload_env(constant_env())
        ref, _ = sc.render(CAM, DEFAULT_CLOCK, 6, 1.0, 64, 40, 0, 6)
        assert np.array_equal(want, ref)
        ctx.resize(64, 40)                       # clears the sums
        graph, graph_exec = ctypes.c_void_p(), ctypes.c_void_p()
        hip.check(hip.lib.hipStreamBeginCapture(stream, 0))
        try:
            ctx.trace(0, 6, 1)
        finally:
            rc = hip.lib.hipStreamEndCapture(stream, ctypes.byref(graph))
        hip.check(rc)
        hip.check(hip.lib.hipGraphInstantiate(ctypes.byref(graph_exec), graph, None, None, 0))
        # capturing executes nothing, so one replay renders the 6 samples once
        hip.check(hip.lib.hipGraphLaunch(graph_exec, stream))
        hip.check(hip.lib.hipStreamSynchronize(stream))
        once = ctx.read_sum()[0]
        assert np.array_equal(once, want)
        hip.check(hip.lib.hipGraphLaunch(graph_exec, stream))
        hip.check(hip.lib.hipStreamSynchronize(stream))
        twice = ctx.read_sum()[0]
        assert np.array_equal(twice, want_twice)
        hip.lib.hipGraphExecDestroy(graph_exec)
        hip.lib.hipGraphDestroy(graph)
    finally:
        ctx.close()
        hip.lib.hipStreamDestroy(stream)


@pytest.mark.parametrize("w,h,recursion,count", [(1, 1, 6, 3), (9, 3, 1, 2), (8, 8, 2, 1), (65, 7, 6, 70), (3, 130, 40, 2)])
def test_edge_sizes(gpu_ctx, oracle_mod, cubes, w, h, recursion, count):
    # a single pixel, targets smaller than a tile, RECURSION 1, more samples than fit one pass, RECURSION 40
    env = gradient_env()
    _upload(gpu_ctx, cubes, env)
    sc = oracle_mod.OracleScene(*cubes)
    sc.upload_env(env)
    want, wst = sc.render(CAM, 5, recursion, 1.0, w, h, 0, count)
    img, st = _render(gpu_ctx, w, h, 0, count, recursion=recursion, clock=5, batch=64)
    assert np.array_equal(img, want, equal_nan=True)
    assert st["rays"] == wst["rays"]


def test_box_scene_without_lights(gpu_ctx, oracle_mod):
    scene = load_scene("box")
    _upload(gpu_ctx, scene, constant_env())
    sc = oracle_mod.OracleScene(*scene)
    sc.upload_env(constant_env())
    want, wst = sc.render(CAM, 777, 6, 1.0, 48, 48, 0, 4)
    img, st = _render(gpu_ctx, 48, 48, 0, 4, clock=777)
    assert np.array_equal(img, want)
    assert st["shadow_probes"] == 0 and st["rays"] == wst["rays"]


def test_launch_chaining_and_sample_split(gpu_ctx, cubes):
    _upload(gpu_ctx, cubes, constant_env())
    full, _ = _render(gpu_ctx, 64, 40, 0, 12, batch=5)
    # the same 12 samples in three launches and another batch size: same bits
    gpu_ctx.resize(64, 40)
    gpu_ctx.set_option("batch_samples", 3)
    gpu_ctx.set_frame(CAM, DEFAULT_CLOCK, 6, 1.0)
    gpu_ctx.trace(0, 5, 1)
    gpu_ctx.trace(5, 4, 1)
    gpu_ctx.trace(9, 3, 1)
    part, n = gpu_ctx.read_sum()
    assert n == 12 and np.array_equal(part, full)
    # the multi-GPU split (SAMPLE = rank mod N) emulated on one device: per-rank sums add up to the full image
    acc = np.zeros_like(full, dtype=np.float64)
    for rank in range(3):
        part, _ = _render(gpu_ctx, 64, 40, rank, 4, stride=3)
        acc += part
    assert np.allclose(acc, full, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("case", ["cubes_textured", "cubes_textured_edge"])
def test_textured_materials_bit_exact(gpu_ctx, oracle_mod, case):
    # synthetic albedo / normal / ARM layers on one material: against the images of the reference's shader (single
    # SAMPLE values) and against the checker (4-sample sums and counters).  First case: 32^2 atlas; second: 128^2 atlas
    # whose albedo texels include the edge values of the packed-float upload conversion (k_pack_atlas on the device)
    g = np.load(os.path.join(GOLDEN, "ref_image_%s.npz" % case))
    S = int(g["atlas_size"])
    (models, shadings, lights, materials), layers, flags = textured_cubes(S, edge_texels=case.endswith("edge"))
    gpu_ctx.upload_scene(models, shadings, lights, materials)
    gpu_ctx.upload_env(g["env"])
    gpu_ctx.set_atlas_size(S)
    sc = oracle_mod.OracleScene(models, shadings, lights, materials)
    sc.upload_env(g["env"])
    sc.set_atlas_size(S)
    for target in (gpu_ctx, sc):
        for k in (0, 1, 2):
            target.upload_atlas_layer(2, k, layers[k])
        target.set_has_texture(flags)
    clock, recursion = int(g["clock"]), int(g["recursion"])
    want, wst = sc.render(g["camera"], clock, recursion, 1.0, 48, 32, 0, 4)
    for pipeline in (0, 1, 3, 4, 5, 6, 9):
        img, st = _render(gpu_ctx, 48, 32, 0, 4, clock=clock, recursion=recursion, pipeline=pipeline, cam=g["camera"])
        assert np.array_equal(img, want, equal_nan=True), (pipeline, float(np.nanmax(np.abs(img - want))))
        assert st["rays"] == wst["rays"]
        for s in g["samples"].tolist():
            ref = g["sample_%d" % s]
            colour, _ = _render(gpu_ctx, 48, 32, s, 1, clock=clock, recursion=recursion, pipeline=pipeline, cam=g["camera"])
            got = colour
            if s > 0:
                inv = np.float32(1.0) / np.float32(s + 1)
                got = (np.float32(0.0) * (np.float32(s) * inv) + inv * colour).astype(np.float32)
            same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
            assert same.all(), (pipeline, s)
    gpu_ctx.set_atlas_size(4096)


def test_nonfinite_and_edge_texels_bit_exact(gpu_ctx, oracle_mod):
    # the device's upload conversion (k_pack_atlas) on Inf / NaN / negative / tiny / huge texels, seen through whole
    # images: the checker's conversion is pinned to the GL implementation's by its own test (ref_builtins.npz: pack_x)
    S = 64
    (models, shadings, lights, materials), layers, flags = textured_cubes(S, edge_texels=True, nonfinite_texels=True)
    env = gradient_env()
    gpu_ctx.upload_scene(models, shadings, lights, materials)
    gpu_ctx.upload_env(env)
    gpu_ctx.set_atlas_size(S)
    sc = oracle_mod.OracleScene(models, shadings, lights, materials)
    sc.upload_env(env)
    sc.set_atlas_size(S)
    for target in (gpu_ctx, sc):
        for k in (0, 1, 2):
            target.upload_atlas_layer(2, k, layers[k])
        target.set_has_texture(flags)
    want, wst = sc.render(CAM, 31337, 6, 1.0, 64, 48, 0, 6)
    assert (~np.isfinite(want)).any() and np.isfinite(want).any()
    for pipeline in (1, 3):
        img, st = _render(gpu_ctx, 64, 48, 0, 6, clock=31337, pipeline=pipeline)
        same = (img.view(np.uint32) == want.view(np.uint32)) | (np.isnan(img) & np.isnan(want))
        assert same.all(), (pipeline, int((~same).sum()))
        assert st["rays"] == wst["rays"]
    gpu_ctx.set_atlas_size(4096)


@pytest.mark.parametrize("accel", [0, -1])
def test_denormal_numbers_are_kept_like_llvmpipe(gpu_ctx, accel):
    # the reference's findIntersection under llvmpipe on the cubes scene scaled by 2^-42 and 2^-44: the plane distances
    # are denormal numbers there, and the hits it reports are those of arithmetic that keeps them.  The products of the
    # reference's depth comparison (raytracer.glsl:150) underflow to zero at this scale, so that the later index wins
    # whatever the depths: the library's automatic choice (accel -1) keeps to the loop over all triangles for scenes
    # with normals this small; the hierarchy forced on them (accel 1) prunes by depth and is not expected to agree.
    from conftest import tiny_cubes
    g = np.load(os.path.join(GOLDEN, "ref_hits_tiny.npz"))
    for k, scale in enumerate(g["scales"]):
        _upload(gpu_ctx, tiny_cubes(scale), constant_env())
        gpu_ctx.set_option("accel", accel)
        ids, uvt = gpu_ctx.closest_hit_batch(g["o_%d" % k], g["d"])
        assert np.array_equal(ids, g["ids_%d" % k]), int((ids != g["ids_%d" % k]).sum())
        hit = ids >= 0
        assert np.array_equal(uvt[hit].view(np.uint32), g["uvt_%d" % k][hit].view(np.uint32))


def _write_soup(base, seed, n_tris):
    """Random triangle soup as OBJ/MTL: sizes over four orders of magnitude, exact duplicates (ties between
    indices), zero-area triangles, a far-away cluster, emissive triangles, glass, metal and rough materials."""
    rng = np.random.default_rng(seed)
    with open(base + ".mtl", "w") as f:
        f.write("newmtl A\nNs 700\nKd 0.8 0.7 0.6\nKs 0.2 0.2 0.2\nKe 0 0 0\nNi 1.0\n\n")
        f.write("newmtl B\nNs 40\nKd 0.3 0.5 0.9\nKs 0.9 0.9 0.9\nKe 0 0 0\nNi 1.5\n\n")
        f.write("newmtl L\nNs 900\nKd 1 1 1\nKs 0 0 0\nKe 9 8 7\nNi 1.0\n\n")
        f.write("newmtl M\nNs 100\nKd 0.9 0.6 0.2\nKs 0.9 0.6 0.2\nKe 0 0 0\nNi 0.0\n\n")
    verts, faces = [], {"A": [], "B": [], "L": [], "M": []}
    names = ["A", "B", "L", "M"]
    for k in range(n_tris):
        scale = 10.0 ** rng.uniform(-2.5, 0.7)
        centre = rng.uniform(-2.0, 2.0, 3) + (np.array([40.0, 0.0, 40.0]) if k % 17 == 0 else 0.0)
        tri = centre + rng.normal(0.0, 1.0, (3, 3)) * scale
        if k % 23 == 0:
            tri[2] = tri[1]                                   # zero area
        base_index = len(verts)
        verts.extend(tri.tolist())
        name = names[int(rng.integers(0, 4))] if k % 9 else "L"
        faces[name].append(base_index)
        if k % 11 == 0:                                       # the same triangle again, later in the file
            faces[names[int(rng.integers(0, 4))]].append(base_index)
    with open(base + ".obj", "w") as f:
        f.write("mtllib %s.mtl\n" % os.path.basename(base))
        for v in verts:
            f.write("v %.9g %.9g %.9g\n" % tuple(v))
        f.write("vn 0 1 0\nvn 0.6 0.8 0\nvn 0 0.6 -0.8\nvt 0 0\nvt 1 0\nvt 0 1\n")
        for name in names:
            f.write("o %s\nusemtl %s\n" % (name, name))
            for b in faces[name]:
                f.write("f %d/1/1 %d/2/2 %d/3/3\n" % (b + 1, b + 2, b + 3))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_soups(gpu_ctx, oracle_mod, tmp_path, seed):
    # geometry nobody tuned for: every pipeline form against the checker, bit for bit, on random triangle soups
    from gpuraytracer_amd import host
    base = str(tmp_path / ("soup%d" % seed))
    _write_soup(base, seed, 300 + 700 * seed)
    scene = host.load_wavefront(base)
    env = gradient_env()
    _upload(gpu_ctx, scene, env)
    sc = oracle_mod.OracleScene(*scene)
    sc.upload_env(env)
    cam = camera_matrix((0.2, 0.4, -4.5), (-0.1, 0.05))
    want, wst = sc.render(cam, 1000 + seed, 8, 1.0, 72, 56, 0, 5)
    assert wst["rays"] > 1.05 * wst["pixel_samples"] and wst["shadow_probes"] > 500 and wst["glossy_probes"] > 500
    for pipeline in (0, 1, 2, 3, 4, 5, 7, 9):
        img, st = _render(gpu_ctx, 72, 56, 0, 5, recursion=8, clock=1000 + seed, pipeline=pipeline, batch=2, cam=cam)
        same = (img.view(np.uint32) == want.view(np.uint32)) | (np.isnan(img) & np.isnan(want))
        assert same.all(), (seed, pipeline, int((~same).sum()))
        for k in ("rays", "shadow_probes", "glossy_probes"):
            assert st[k] == wst[k], (seed, pipeline, k)


def _flat_rays(rng, n, lo, hi):
    """Rays with direction components of exactly zero, of both signs of zero, denormal, and tiny but normal."""
    o = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    tiny = np.array([0.0, -0.0, 1e-39, -1e-39, 1e-30, -1e-30, 2.0 ** -70, -(2.0 ** -60)], np.float32)
    which = rng.integers(0, 7, size=n)                     # bit k: component k is replaced; 0 leaves the ray as it is
    for k in range(3):
        sel = ((which >> k) & 1) == 1
        d[sel, k] = tiny[rng.integers(0, len(tiny), size=int(sel.sum()))]
    return o, d


@pytest.mark.parametrize("name", ["cubes", "soup"])
def test_axis_parallel_rays(gpu_ctx, oracle_mod, tmp_path, name):
    # a zero direction component makes the reciprocal infinite: the box tests must still visit every candidate
    if name == "soup":
        from gpuraytracer_amd import host
        base = str(tmp_path / "soup_flat")
        _write_soup(base, 21, 1500)
        scene = host.load_wavefront(base)
    else:
        scene = load_scene(name)
    _upload(gpu_ctx, scene, constant_env())
    o, d = _flat_rays(np.random.default_rng(5), 60000, -2.5, 2.5)
    # origins on the coordinates of triangle corners as well: the ray then lies in the plane of a box side
    p0 = scene[0]["position"]
    o[:20000] = p0[np.random.default_rng(6).integers(0, len(p0), size=20000)]
    want_ids, want_uvt = oracle_mod.OracleScene(*scene).closest_hit_batch(o, d)
    assert (want_ids >= 0).mean() > 0.1
    for accel in (0, 1):
        gpu_ctx.set_option("accel", accel)
        ids, uvt = gpu_ctx.closest_hit_batch(o, d)
        assert np.array_equal(ids, want_ids), (accel, int((ids != want_ids).sum()))
        assert np.array_equal(uvt.view(np.uint32), want_uvt.view(np.uint32)), accel


@pytest.mark.parametrize("zero_row", [0, 1, 2])
@pytest.mark.parametrize("name", ["cubes", "soup"])
def test_axis_parallel_render(gpu_ctx, oracle_mod, tmp_path, name, zero_row):
    # a camera matrix with a zero row: every primary ray has that direction component exactly zero, in all
    # forms of the pipeline (resident hierarchy for the cubes, hierarchy in memory for the soup)
    if name == "soup":
        from gpuraytracer_amd import host
        base = str(tmp_path / "soup_flat")
        _write_soup(base, 22, 900)
        scene = host.load_wavefront(base)
        cam = camera_matrix((0.2, 0.4, -4.5), (-0.1, 0.05))
    else:
        scene = load_scene(name)
        cam = camera_matrix()
    cam = cam.copy()
    cam[[zero_row, 4 + zero_row, 8 + zero_row]] = 0.0
    env = gradient_env()
    _upload(gpu_ctx, scene, env)
    sc = oracle_mod.OracleScene(*scene)
    sc.upload_env(env)
    want, wst = sc.render(cam, 4242, 6, 1.0, 64, 48, 0, 4)
    assert wst["rays"] > wst["pixel_samples"]
    for pipeline in (1, 3, 5, 6, 7, 9):
        img, st = _render(gpu_ctx, 64, 48, 0, 4, recursion=6, clock=4242, pipeline=pipeline, batch=2, cam=cam)
        same = (img.view(np.uint32) == want.view(np.uint32)) | (np.isnan(img) & np.isnan(want))
        assert same.all(), (pipeline, int((~same).sum()))
        for k in ("rays", "shadow_probes", "glossy_probes"):
            assert st[k] == wst[k], (pipeline, k)


@pytest.mark.parametrize("seed", [1111, 1260])
def test_soak_findings(gpu_ctx, tmp_path, seed):
    # scenes of tools/soak_parity.py in which one ray had a direction component of exactly zero (a primary ray in
    # the first, a glossy probe in the second) and the hierarchy forms lost its hit
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import soak_parity
    scene, cam, w, h, spp, rec, clock, batches = soak_parity.case_for(seed, str(tmp_path), _write_soup)
    _upload(gpu_ctx, scene, gradient_env())
    want, wst = _render(gpu_ctx, w, h, 0, spp, recursion=rec, clock=clock, pipeline=3, cam=cam)
    for pipeline in (1, 5, 7, 9):
        img, st = _render(gpu_ctx, w, h, 0, spp, recursion=rec, clock=clock, pipeline=pipeline, cam=cam)
        same = (img.view(np.uint32) == want.view(np.uint32)) | (np.isnan(img) & np.isnan(want))
        assert same.all(), (pipeline, int((~same).sum()))
        for k in ("rays", "shadow_probes", "glossy_probes"):
            assert st[k] == wst[k], (pipeline, k)


def test_irradiance_prefilter_bit_exact(gpu_ctx, oracle_mod):
    # Scene.cpp:150-186 + irradiance.glsl against the reference shader's own output, and a larger case against the oracle
    g = np.load(os.path.join(GOLDEN, "ref_irradiance.npz"))
    for k in range(3):
        gpu_ctx.upload_env(g["env%d" % k])
        got = gpu_ctx.prefilter_irradiance()
        assert got.shape == g["irr%d" % k].shape
        assert (got.view(np.uint32) == g["irr%d" % k].view(np.uint32)).all(), k
    rng = np.random.default_rng(11)
    env = rng.random((96, 200, 4), dtype=np.float32) * 5.0
    gpu_ctx.upload_env(env)
    got = gpu_ctx.prefilter_irradiance()
    want = oracle_mod.irradiance(env)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    # an environment narrower than 4 texels has an empty irradiance image
    gpu_ctx.upload_env(np.ones((2, 3, 4), dtype=np.float32))
    assert gpu_ctx.prefilter_irradiance().shape == (0, 0, 3)


def test_srgb8_readback(gpu_ctx, oracle_mod, cubes):
    _upload(gpu_ctx, cubes, constant_env())
    img, _ = _render(gpu_ctx, 32, 32, 0, 4)
    rgb8 = gpu_ctx.read_srgb8()
    mean = img / 4.0
    want = np.vectorize(oracle_mod.linear_to_srgb)(mean.astype(np.float64)).astype(np.float64)
    want8 = np.clip(np.floor(want * 255.0 + 0.5), 0, 255)
    assert np.abs(rgb8.astype(np.int32) - want8.astype(np.int32)).max() <= 1     # displayQuad.fs:19-25, +-1 code
    # the file mapping of Scene::exportBMP (Scene.cpp:66-73, :564-567): ACES film curve, sRGB transfer, * 256 truncated
    gpu_ctx.set_option("tonemap", 1)
    aces8 = gpu_ctx.read_srgb8()
    gpu_ctx.set_option("tonemap", 0)
    x = mean.astype(np.float64)
    film = np.clip((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14), 0.0, 1.0)
    want = np.vectorize(oracle_mod.linear_to_srgb)(film)
    want8 = np.clip(np.floor(want * 256.0), 0, 255)
    assert np.abs(aces8.astype(np.int32) - want8.astype(np.int32)).max() <= 1
    assert (aces8 != rgb8).any()


def test_error_behaviour(oracle_mod, cubes):
    import gpuraytracer_amd as g
    ctx = g.Context(0)
    try:
        # a whole MI355X (256 compute units) is what the work distribution is laid out for: no note about the device
        assert ctx.lib.rt_last_error(ctx.h).decode() == ""
        with pytest.raises(g.GpuRtError):       # no scene yet
            ctx.trace(0, 1, 1)
        models, shadings, lights, materials = cubes
        with pytest.raises(g.GpuRtError):       # lights > count
            ctx.upload_scene(models[:4], shadings[:4], 5, materials)
        bad = shadings[:4].copy()
        bad["material_id"] = 9
        with pytest.raises(g.GpuRtError):       # material id out of range
            ctx.upload_scene(models[:4], bad, 0, materials)
        # 2^28 triangles and more are refused before any record is read (the leaf links of the hierarchy hold 28 bits
        # of slot number): the call goes straight through the C ABI with the small arrays' pointers
        import ctypes
        rc = ctx.lib.rt_upload_scene(ctx.h, models.ctypes.data_as(ctypes.c_void_p), shadings.ctypes.data_as(ctypes.c_void_p),
                                     1 << 28, 0, materials.ctypes.data_as(ctypes.c_void_p), len(materials))
        assert rc == -5, rc                     # RT_ERR_LIMIT
        ctx.upload_scene(models, shadings, lights, materials)
        with pytest.raises(g.GpuRtError):       # no frame / no target
            ctx.trace(0, 1, 1)
        ctx.resize(8, 8)
        ctx.set_frame(CAM, 1, 6, 1.0)
        with pytest.raises(g.GpuRtError):
            ctx.trace(0, 1, 0)                  # stride 0
        with pytest.raises(g.GpuRtError):
            ctx.set_option("no_such_option", 1)
        ctx.trace(0, 0, 1)                      # zero samples is a no-op
        ctx.upload_env(constant_env())
        ctx.trace(0, 1, 1)
        img, n = ctx.read_sum()
        assert n == 1 and np.isfinite(img).all()
    finally:
        ctx.close()


def test_synthetic_height_field_textured(gpu_ctx, oracle_mod, tmp_path):
    # the C3-style workload at small size: OBJ/MTL ingest -> hierarchy -> textured shading; hierarchy, loop form
    # and the checker agree bit for bit
    from gpuraytracer_amd import host, synthetic
    base = str(tmp_path / "grid")
    n_tris = synthetic.write_displaced_grid(base, 40)
    scene = host.load_wavefront(base)
    assert len(scene[0]) == n_tris and scene[2] == 4
    layers = synthetic.texture_layers(64)
    flags = np.array([1, 1, 0], np.int32)
    cam = camera_matrix((0.0, 2.2, -5.5), (-0.45, 0.0))
    gpu_ctx.upload_scene(*scene)
    gpu_ctx.upload_env(gradient_env())
    gpu_ctx.set_atlas_size(64)
    sc = oracle_mod.OracleScene(*scene)
    sc.upload_env(gradient_env())
    sc.set_atlas_size(64)
    for target in (gpu_ctx, sc):
        for m in (0, 1):
            for k in range(3):
                target.upload_atlas_layer(m, k, layers[k])
        target.set_has_texture(flags)
    want, wst = sc.render(cam, 777, 6, 1.0, 64, 40, 0, 2)
    for pipeline in (1, 3, 4, 5):
        img, st = _render(gpu_ctx, 64, 40, 0, 2, clock=777, pipeline=pipeline, cam=cam)
        assert np.array_equal(img, want, equal_nan=True), (pipeline, float(np.nanmax(np.abs(img - want))))
        assert st["rays"] == wst["rays"] and st["shadow_probes"] == wst["shadow_probes"]
    assert wst["rays"] > 1.05 * wst["pixel_samples"] and wst["shadow_probes"] > 0     # the camera sees the surface
    gpu_ctx.set_atlas_size(4096)


@pytest.fixture(scope="module")
def million_triangles(tmp_path_factory):
    from gpuraytracer_amd import host, synthetic
    base = str(tmp_path_factory.mktemp("c3") / "grid")
    n = synthetic.write_displaced_grid(base, 724)
    scene = host.load_wavefront(base)
    assert len(scene[0]) == n == 1048358 and scene[2] == 4
    return scene


def test_million_triangles_hierarchy_equals_loop(gpu_ctx, million_triangles):
    # What scenes of 65 536 triangles or more take and smaller ones do not: leaves of at most two triangles, lane
    # refill switched on by itself, any-hit searches with a node budget per round, a hierarchy deeper than 20.
    models = million_triangles[0]
    _upload(gpu_ctx, million_triangles, gradient_env())
    rng = np.random.default_rng(3)
    n = 120000
    o = np.empty((n, 3)); d = np.empty((n, 3))
    k = n // 3
    # a third from where the measurement's camera stands, a third from anywhere, a third aimed exactly at vertices and
    # edge midpoints of the height field (several triangles tie in depth there)
    o[:k] = np.array([0.0, 2.2, -5.5]) + rng.normal(0, 0.05, (k, 3))
    d[:k] = rng.normal(0, 1, (k, 3)) * [0.6, 0.35, 0.2] + [0.0, -0.45, 0.9]
    o[k:] = rng.uniform([-5, -1, -5], [5, 4, 5], (n - k, 3))
    d[k:2 * k] = rng.normal(0, 1, (k, 3))
    tri = rng.integers(0, len(models), n - 2 * k)
    a = models["position"][tri].astype(np.float64)
    b = a + models["u"][tri]
    target = np.where(rng.integers(0, 2, n - 2 * k)[:, None] == 0, a, 0.5 * (a + b))
    d[2 * k:] = target - o[2 * k:]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o32, d32 = o.astype(np.float32), d.astype(np.float32)
    gpu_ctx.set_option("accel", 0)
    ids0, uvt0 = gpu_ctx.closest_hit_batch(o32, d32)
    gpu_ctx.set_option("accel", 1)
    ids1, uvt1 = gpu_ctx.closest_hit_batch(o32, d32)
    assert np.array_equal(ids0, ids1), int((ids0 != ids1).sum())
    assert np.array_equal(uvt0.view(np.uint32), uvt1.view(np.uint32))
    assert (ids0 >= 0).mean() > 0.5 and len(np.unique(ids0)) > 20000
    # whole paths: the loop form is the reference's algorithm; the staged forms must give its bits
    cam = camera_matrix((0.0, 2.2, -5.5), (-0.45, 0.0))
    want, wst = _render(gpu_ctx, 64, 40, 0, 2, clock=4711, pipeline=3, batch=2, cam=cam)
    assert wst["rays"] > 1.1 * wst["pixel_samples"] and wst["shadow_probes"] > 300 and wst["glossy_probes"] > 300
    for pipeline in (1, 2, 4, 5, 7):
        img, st = _render(gpu_ctx, 64, 40, 0, 2, clock=4711, pipeline=pipeline, batch=2, cam=cam)
        same = (img.view(np.uint32) == want.view(np.uint32)) | (np.isnan(img) & np.isnan(want))
        assert same.all(), (pipeline, int((~same).sum()))
        for key in ("rays", "shadow_probes", "glossy_probes"):
            assert st[key] == wst[key], (pipeline, key)


@pytest.fixture(scope="module")
def c3_layers():
    # the 4096^2 albedo / normal / ARM layers of the measurement's 1 M-triangle workload
    from gpuraytracer_amd import synthetic
    return synthetic.texture_layers(4096)


def _upload_c3_textures(target, layers):
    target.set_atlas_size(4096)
    for m in (0, 1):
        for k in range(3):
            target.upload_atlas_layer(m, k, layers[k])
    target.set_has_texture(np.array([1, 1, 0], np.int32))


def _c3_ray_batch(models, n, seed):
    """Rays for the 1 M-triangle field: from the measurement's camera, from anywhere, aimed exactly at vertices, edge
    midpoints and centroids (several triangles tie in depth there), parallel to an axis plane, and straight down."""
    rng = np.random.default_rng(seed)
    o = np.empty((n, 3)); d = np.empty((n, 3))
    k = n // 5
    o[:k] = np.array([0.0, 2.2, -5.5]) + rng.normal(0, 0.05, (k, 3))
    d[:k] = rng.normal(0, 1, (k, 3)) * [0.6, 0.35, 0.2] + [0.0, -0.45, 0.9]
    o[k:] = rng.uniform([-5, 0.5, -5], [5, 4, 5], (n - k, 3))
    d[k:2 * k] = rng.normal(0, 1, (k, 3))
    tri = rng.integers(0, len(models), 2 * k)
    a = models["position"][tri].astype(np.float64)
    b, c = a + models["u"][tri], a + models["v"][tri]
    w = rng.integers(0, 4, 2 * k)[:, None]
    target = np.where(w == 0, a, np.where(w == 1, 0.5 * (a + b), np.where(w == 2, 0.5 * (b + c), (a + b + c) / 3)))
    d[2 * k:4 * k] = target - o[2 * k:4 * k]
    d[4 * k:] = rng.normal(0, 1, (n - 4 * k, 3)) * [0.3, 0.3, 0.3] + [0.0, -1.0, 0.0]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o32, d32 = o.astype(np.float32), d.astype(np.float32)
    flat = np.arange(4 * k, n)
    d32[flat[0::3], 0] = 0.0            # exactly zero direction components: straight down, or in an axis plane
    d32[flat[1::3], 2] = -0.0
    d32[flat[2::6], 0] = 0.0
    d32[flat[2::6], 2] = 0.0
    return o32, d32


def test_million_triangles_closest_hit_equals_checker(gpu_ctx, oracle_mod, million_triangles):
    # the 1 M-triangle field against the CPU checker's loop over all triangles (the reference's findIntersection): what
    # only scenes of 65 536 triangles or more use (leaves of two triangles, the hierarchy built by several threads, a
    # depth beyond 20) has an independent witness here; ids and (u, v, t) bit for bit, loop form and hierarchy
    models = million_triangles[0]
    _upload(gpu_ctx, million_triangles, gradient_env())
    o, d = _c3_ray_batch(models, 4000, 17)
    sc = oracle_mod.OracleScene(*million_triangles)
    want_ids, want_uvt = sc.closest_hit_batch(o, d)
    sc.close()
    assert (want_ids >= 0).mean() > 0.7 and len(np.unique(want_ids)) > 2000
    for accel in (0, 1):
        gpu_ctx.set_option("accel", accel)
        ids, uvt = gpu_ctx.closest_hit_batch(o, d)
        assert np.array_equal(ids, want_ids), (accel, int((ids != want_ids).sum()))
        hit = ids >= 0
        assert np.array_equal(uvt[hit].view(np.uint32), want_uvt[hit].view(np.uint32)), accel


def test_million_triangles_textured_render_equals_checker(gpu_ctx, oracle_mod, million_triangles, c3_layers):
    # the measurement's 1 M-triangle workload as it is measured (4096^2 albedo / normal / ARM layers on two materials,
    # the library's own choice of forms: lane refill, shared searches with a node budget, two-triangle leaves), small
    # enough for the CPU checker: 48 x 32 pixels, two samples, sums and counters bit for bit
    env = gradient_env()
    cam = camera_matrix((0.0, 2.2, -5.5), (-0.45, 0.0))
    _upload(gpu_ctx, million_triangles, env)
    _upload_c3_textures(gpu_ctx, c3_layers)
    sc = oracle_mod.OracleScene(*million_triangles)
    sc.upload_env(env)
    _upload_c3_textures(sc, c3_layers)
    want, wst = sc.render(cam, 4711, 6, 1.0, 48, 32, 0, 2)
    sc.close()
    assert wst["rays"] > 1.1 * wst["pixel_samples"] and wst["shadow_probes"] > 200 and wst["glossy_probes"] > 200
    for pipeline in (1, 5, 7):
        img, st = _render(gpu_ctx, 48, 32, 0, 2, clock=4711, pipeline=pipeline, batch=2, cam=cam)
        same = (img.view(np.uint32) == want.view(np.uint32)) | (np.isnan(img) & np.isnan(want))
        assert same.all(), (pipeline, int((~same).sum()))
        for key in ("pixel_samples", "rays", "shadow_probes", "glossy_probes"):
            assert st[key] == wst[key], (pipeline, key)
    gpu_ctx.set_atlas_size(4096)


def test_million_triangles_full_size_forms_agree(gpu_ctx, million_triangles, c3_layers):
    # 1920 x 1080 on the 1 M-triangle scene with its texture layers, 3 samples in one pass and in passes of 2 + 1: the
    # library's own choice of forms (lane refill, budgeted shared searches, chunks handed out as waves become free, tail
    # kernel on even slices) against every lane searching alone with every bounce staged, bit for bit, with equal counters
    _upload(gpu_ctx, million_triangles, gradient_env())
    _upload_c3_textures(gpu_ctx, c3_layers)
    cam = camera_matrix((0.0, 2.2, -5.5), (-0.45, 0.0))
    a, sa = _render(gpu_ctx, 1920, 1080, 0, 3, clock=99, pipeline=1, batch=3, cam=cam)
    b, sb = _render(gpu_ctx, 1920, 1080, 0, 3, clock=99, pipeline=7, batch=2, cam=cam)
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    assert same.all(), int((~same).sum())
    for key in ("rays", "shadow_probes", "glossy_probes", "pixel_samples"):
        assert sa[key] == sb[key], key
    assert sa["rays"] > 1.15 * sa["pixel_samples"] and np.isfinite(a).mean() > 0.999
    gpu_ctx.set_atlas_size(4096)


from exr_plain import read_exr_scanlines as _read_exr_scanlines  # noqa: E402


def test_host_program_renders_and_writes_exr(oracle_mod, tmp_path):
    # The C++ host (csrc/host/: Scene with the reference's method names + the gpuPT main, the counterpart of
    # RayTracer.cpp's finalRender) end to end on the GPU: OBJ/MTL ingest, upload, 4 samples, mean image to OpenEXR.
    # The file is parsed here independently of the library's own reader; rows are bottom-up as Scene::exportEXR
    # leaves them (Scene.cpp:496-509), channels alphabetical (B, G, R).
    import subprocess
    from conftest import ROOT
    from gpuraytracer_amd import host
    base = str(tmp_path / "soup")
    _write_soup(base, 7, 500)
    out = str(tmp_path / "final.exr")
    exe = os.path.join(ROOT, "gpuraytracer_amd", "gpuPT")
    assert os.path.exists(exe), "build gpuPT with __graft_entry__.build()"
    res = subprocess.run([exe, base, "64", "48", "4", out], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "Scene Loaded" in res.stdout and "samples/s" in res.stdout
    planes, names, attrs = _read_exr_scanlines(out)
    assert names == ["B", "G", "R"]
    assert attrs["compression"][1] == b"\x03"          # ZIP, what tinyexr's SaveEXR (Scene.cpp:504) picks from 16 pixels on
    scene = host.load_wavefront(base)
    sc = oracle_mod.OracleScene(*scene)
    sc.upload_env(constant_env())
    want, _ = sc.render(CAM, DEFAULT_CLOCK, 6, 1.0, 64, 48, 0, 4)          # the program's camera, clock and RECURSION
    mean = want * (np.float32(1.0) / np.float32(4.0))
    got = np.stack([planes["R"], planes["G"], planes["B"]], axis=-1)
    same = (got.view(np.uint32) == mean.view(np.uint32)) | (np.isnan(got) & np.isnan(mean))
    assert same.all(), int((~same).sum())
    assert np.abs(got[0] - got[-1]).max() > 0                              # row 0 is TEXEL.y 0: the image is not symmetric
    # and the library's own reader agrees with the independent one
    back = host.read_exr(out)
    assert np.array_equal(back[..., :3].view(np.uint32), got.view(np.uint32))


def test_host_program_loads_material_textures_from_exr(oracle_mod, tmp_path):
    # Scene::loadMaterial of the C++ host (Scene.cpp:206-229 with loadTexture, :75-102): <base>_albedo.exr, _normal.exr
    # (written by the library's EXR writer) and _arm.exr (HALF channels, PIZ-compressed, written by the tests' Python
    # writer) go through its reader into the corner of 4096^2 atlas layers
    # (the reference's atlas size) of two materials of a textured height field; the rendered mean image against the checker
    import subprocess
    from conftest import ROOT
    from gpuraytracer_amd import host, synthetic
    base = str(tmp_path / "grid")
    synthetic.write_displaced_grid(base, 24)
    from test_host import _piz_exr
    layers = list(synthetic.texture_layers(64))
    layers[2] = layers[2].astype(np.float16).astype(np.float32)            # this layer goes through a HALF file
    tex = str(tmp_path / "tex")
    for suffix, pixels in zip(("albedo", "normal"), layers):
        host.write_exr("%s_%s.exr" % (tex, suffix), pixels[..., :3])
    piz = {}
    _piz_exr(tex + "_arm.exr", np.ascontiguousarray(layers[2][..., :3]), {"R": 1, "G": 1, "B": 1}, stats=piz)
    assert piz["coded_blocks"] == 2                                        # PIZ blocks, not blocks stored as they are
    out = str(tmp_path / "final.exr")
    exe = os.path.join(ROOT, "gpuraytracer_amd", "gpuPT")
    res = subprocess.run([exe, base, "64", "48", "3", out, "-", "0=" + tex, "1=" + tex, "2=" + str(tmp_path / "missing")],
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "Material 0 textured" in res.stdout and "Material 1 textured" in res.stdout and "Material 2 not textured" in res.stdout
    scene = host.load_wavefront(base)
    sc = oracle_mod.OracleScene(*scene)
    sc.upload_env(constant_env())
    sc.set_atlas_size(4096)
    for m in (0, 1):
        for k in range(3):
            back = host.read_exr("%s_%s.exr" % (tex, ("albedo", "normal", "arm")[k]))
            assert np.array_equal(back[..., :3], layers[k][..., :3])          # float32 pixels survive the file unchanged
            sc.upload_atlas_layer(m, k, back)
    sc.set_has_texture(np.array([1, 1, 0], np.int32))
    want, _ = sc.render(CAM, DEFAULT_CLOCK, 6, 1.0, 64, 48, 0, 3)
    mean = want * (np.float32(1.0) / np.float32(3.0))
    planes, names, _ = _read_exr_scanlines(out)
    got = np.stack([planes["R"], planes["G"], planes["B"]], axis=-1)
    same = (got.view(np.uint32) == mean.view(np.uint32)) | (np.isnan(got) & np.isnan(mean))
    assert same.all(), int((~same).sum())


def test_several_contexts_in_one_process_add_up(oracle_mod, cubes, tmp_path):
    # One host process driving several GPUs (here: three contexts on the one GPU of the test machine): context r of N
    # renders SAMPLE = r + k * N, context 0 then adds the others with rt_accumulate_from in ascending order.  The checker
    # reproduces exactly that association of the sums.  Then the same through the C++ host program (--devices=0,0).
    import subprocess
    import gpuraytracer_amd as g
    from conftest import ROOT
    from gpuraytracer_amd import host
    from gpuraytracer_amd.distributed import shard_samples
    W, H, SPP, N = 40, 24, 7, 3
    ctxs = [g.Context(0) for _ in range(N)]
    try:
        for r, ctx in enumerate(ctxs):
            _upload(ctx, cubes, constant_env())
            ctx.resize(W, H)
            ctx.set_frame(CAM, DEFAULT_CLOCK, 6, 1.0)
            begin, count, stride = shard_samples(SPP, r, N)
            ctx.trace(begin, count, stride)
        for other in ctxs[1:]:
            ctxs[0].accumulate_from(other)
        with pytest.raises(g.GpuRtError):
            ctxs[1].resize(W + 8, H)
            ctxs[0].accumulate_from(ctxs[1])                   # targets differ
        got, n = ctxs[0].read_sum()
    finally:
        for ctx in ctxs:
            ctx.close()
    assert n == SPP
    sc = oracle_mod.OracleScene(*cubes)
    sc.upload_env(constant_env())
    want = None
    for r in range(N):
        part, _ = sc.render(CAM, DEFAULT_CLOCK, 6, 1.0, W, H, *shard_samples(SPP, r, N))
        want = part if want is None else (want + part).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))

    base = str(tmp_path / "soup")
    _write_soup(base, 3, 300)
    out = str(tmp_path / "two.exr")
    exe = os.path.join(ROOT, "gpuraytracer_amd", "gpuPT")
    res = subprocess.run([exe, "--devices=0,0", base, "40", "24", "5", out], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "on 2 GPUs" in res.stdout
    scene = host.load_wavefront(base)
    sc = oracle_mod.OracleScene(*scene)
    sc.upload_env(constant_env())
    a, _ = sc.render(CAM, DEFAULT_CLOCK, 6, 1.0, 40, 24, 0, 3, 2)
    b, _ = sc.render(CAM, DEFAULT_CLOCK, 6, 1.0, 40, 24, 1, 2, 2)
    mean = (a + b).astype(np.float32) * (np.float32(1.0) / np.float32(5.0))
    planes, names, _ = _read_exr_scanlines(out)
    got = np.stack([planes["R"], planes["G"], planes["B"]], axis=-1)
    same = (got.view(np.uint32) == mean.view(np.uint32)) | (np.isnan(got) & np.isnan(mean))
    assert same.all(), int((~same).sum())
