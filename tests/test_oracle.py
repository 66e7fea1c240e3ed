"""CPU tests of the checker itself: known answers worked out from the shader text, analytic
intersection cases, and agreement of its two math back ends."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, constant_env, gradient_env, load_scene
from gpuraytracer_amd.camera import camera_matrix, DEFAULT_CLOCK
from oracle import scene_np

# pcgHash known answers, by hand from raytracer.glsl:83-93
PCG_KAT = {0: 129708002, 1: 2831084092, 2: 2055130248, 95834783: 678436348, 0xFFFFFFFF: 3861530882}
PCG_CHAIN = [678436348, 2735613742, 2463557714, 2305248459, 312046001]   # hash iterated from 95834783


def test_pcg_known_answers(oracle_mod):
    for k, v in PCG_KAT.items():
        assert oracle_mod.pcg_hash(k) == v
        assert oracle_mod.pcg_hash(k, libm=True) == v
    s = DEFAULT_CLOCK
    for want in PCG_CHAIN:
        s = oracle_mod.pcg_hash(s)
        assert s == want


def test_unit_float_range(oracle_mod):
    # INV_UINT_MAX folds to exactly 2^-32 (raytracer.glsl:86): the value is hash * 2^-32 rounded to float
    v, s = oracle_mod.unit_float(0)
    assert s == 129708002 and v == np.float32(np.float32(129708002) * np.float32(2.0 ** -32))
    # hashes >= 4294967168 round up to 2^32 as float, so unitFloat can return exactly 1.0
    assert np.float32(4294967168) * np.float32(2.0 ** -32) == np.float32(1.0)


def test_pixel_seed_formula():
    # raytracer.glsl:594 in uint32 arithmetic
    def seed(x, y, S, H=512, clock=DEFAULT_CLOCK):
        return (clock + ((x * H + y + 394587) * (S + 1) * 13) + S) & 0xFFFFFFFF
    assert seed(0, 0, 0) == 100964414
    assert seed(1, 0, 0) == 100971070
    assert seed(0, 1, 0) == 100964427
    assert seed(511, 511, 15) == 232434638


def _single_triangle():
    models = np.zeros(1, dtype=scene_np.TRI_MODEL_DTYPE)
    models["position"] = (0, 0, 0)
    models["u"] = (1, 0, 0)
    models["v"] = (0, 1, 0)
    models["true_normal"] = (0, 0, 1)     # cross(u, v)
    shadings = np.zeros(1, dtype=scene_np.TRI_SHADING_DTYPE)
    shadings["normals"] = (0, 0, 1)
    materials = np.zeros(1, dtype=scene_np.MATERIAL_DTYPE)
    return models, shadings, materials


def test_single_triangle_cases(oracle_mod):
    models, shadings, materials = _single_triangle()
    sc = oracle_mod.OracleScene(models, shadings, 0, materials)
    o = np.array([[0.25, 0.25, 1], [0.5, 0.0, 1], [0.0, 0.0, 1], [1.0, 0.0, 1], [0.5, 0.5, 1],
                  [0.25, 0.25, -1], [0.25, 0.25, 1], [2, 2, 1], [0.6, 0.6, 1]], dtype=np.float32)
    d = np.array([[0, 0, -1]] * 5 + [[0, 0, 1], [0, 0, 1], [0, 0, -1], [0, 0, -1]], dtype=np.float32)
    ids, uvt = sc.closest_hit_batch(o, d)
    # interior, edge, vertex p0, vertex p1, hypotenuse (x + y == det is inside), back face, facing away, outside x2
    assert ids.tolist() == [0, 0, 0, 0, 0, -1, -1, -1, -1]
    assert np.allclose(uvt[0], [0.25, 0.25, 1.0])
    assert np.allclose(uvt[4], [0.5, 0.5, 1.0])
    # a miss leaves the initial state (0, 0, -1, 1): t = -1
    assert uvt[5].tolist() == [0.0, 0.0, -1.0]


def test_tie_later_index_wins(oracle_mod):
    # two coincident triangles: equal cross-products do not reject, so the higher index wins (:148)
    m1, s1, mat = _single_triangle()
    models = np.concatenate([m1, m1, m1])
    shadings = np.concatenate([s1, s1, s1])
    sc = oracle_mod.OracleScene(models, shadings, 0, mat)
    ids, _ = sc.closest_hit_batch(np.array([[0.25, 0.25, 1]], np.float32), np.array([[0, 0, -1]], np.float32))
    assert ids[0] == 2


def test_nan_ray_hits_last_triangle(oracle_mod):
    # every rejection is written as a comparison that is false for NaN, so a NaN ray "hits" everything
    m1, s1, mat = _single_triangle()
    sc = oracle_mod.OracleScene(np.concatenate([m1, m1]), np.concatenate([s1, s1]), 0, mat)
    ids, uvt = sc.closest_hit_batch(np.array([[np.nan] * 3], np.float32), np.array([[np.nan] * 3], np.float32))
    assert ids[0] == 1 and np.isnan(uvt[0]).any()


def test_r11g11b10_roundtrip(oracle_mod):
    # representable values survive; negatives clamp to 0; huge values clamp to the largest finite value
    for v in (0.0, 1.0, 0.5, 0.25, 65024.0, 2.0 ** -14):
        p = oracle_mod.pack_r11g11b10(v, v, 0.0)
        out = oracle_mod.unpack_r11g11b10(p)
        assert out[0] == np.float32(v) and out[1] == np.float32(v)
    assert oracle_mod.unpack_r11g11b10(oracle_mod.pack_r11g11b10(-1.0, 1e9, 1e9)).tolist() == [0.0, 65024.0, 64512.0]
    # the mantissa is truncated (6 bits), values below 2^-14 become 0 (what Mesa stores; see test_reference_glsl.py)
    assert oracle_mod.unpack_r11g11b10(oracle_mod.pack_r11g11b10(1.0 + 1.0 / 128, 0, 0))[0] == 1.0
    assert oracle_mod.unpack_r11g11b10(oracle_mod.pack_r11g11b10(1.0 + 3.0 / 128, 0, 0))[0] == np.float32(1.0 + 1.0 / 64)
    assert oracle_mod.unpack_r11g11b10(oracle_mod.pack_r11g11b10(2.0 ** -15, 2.0 ** -20, 0))[:2].tolist() == [0.0, 0.0]
    # decoding still understands denormal codes
    assert oracle_mod.unpack_r11g11b10(1)[0] == np.float32(2.0 ** -20)


def test_builtins_are_sane(oracle_mod):
    # the restated GL built-ins are approximations: polynomial sin/cos within 2 ulp, the asin/acos/atan
    # approximations within 1e-3 rad, pow(x, 5) within 1e-4 relative
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 2 * np.pi, 4000).astype(np.float32)
    assert np.abs(oracle_mod.math_batch("sin", x) - np.sin(x.astype(np.float64))).max() < 3e-7
    assert np.abs(oracle_mod.math_batch("cos", x) - np.cos(x.astype(np.float64))).max() < 3e-7
    u = rng.uniform(-1, 1, 4000).astype(np.float32)
    assert np.abs(oracle_mod.math_batch("acos", u) - np.arccos(u.astype(np.float64))).max() < 1e-3
    assert np.abs(oracle_mod.math_batch("asin", u) - np.arcsin(u.astype(np.float64))).max() < 1e-3
    yx = rng.normal(size=(4000, 2)).astype(np.float32)
    got = oracle_mod.math_batch("atan2", yx[:, 0], yx[:, 1])
    assert np.abs(got - np.arctan2(yx[:, 0].astype(np.float64), yx[:, 1].astype(np.float64))).max() < 1e-3
    p = rng.uniform(0, 1, 4000).astype(np.float32)
    assert np.allclose(oracle_mod.math_batch("pow5", p), p.astype(np.float64) ** 5, rtol=1e-4, atol=1e-12)
    assert oracle_mod.math_fn("pow5", 0.0) == 0.0 and oracle_mod.math_fn("pow5", 1.0) == 1.0
    assert oracle_mod.math_fn("acos", 1.0) == 0.0 and np.isnan(oracle_mod.math_fn("acos", 2.0))
    assert np.isfinite(oracle_mod.math_fn("atan2", 0.0, 0.0))
    # the C-library back end clamps instead
    assert oracle_mod.math_fn("acos", 2.0, libm=True) == 0.0


def test_golden_ray_batches(oracle_mod):
    # ids and (u, v, t) of the reference's findIntersection (run under llvmpipe), bit for bit
    for name in ("cubes", "_ObiWan", "box", "sphere"):
        g = np.load(os.path.join(GOLDEN, "ref_hits_%s.npz" % name))
        models, shadings, lights, materials = load_scene(name)
        for libm in (False, True):      # hit decisions use + - * / only: both back ends must agree exactly
            sc = oracle_mod.OracleScene(models, shadings, lights, materials, libm=libm)
            ids, uvt = sc.closest_hit_batch(g["o"], g["d"])
            assert np.array_equal(ids, g["ids"])
            hit = ids >= 0
            assert np.array_equal(uvt[hit].view(np.uint32), g["uvt"][hit].view(np.uint32))


def test_golden_images_and_backends(oracle_mod, cubes):
    models, shadings, lights, materials = cubes
    cam = camera_matrix()
    sc = oracle_mod.OracleScene(models, shadings, lights, materials)
    sc.upload_env(constant_env())
    sl = oracle_mod.OracleScene(models, shadings, lights, materials, libm=True)
    sl.upload_env(constant_env())
    for spp in (1, 4, 16):
        g = np.load(os.path.join(GOLDEN, "image_cubes_%dspp.npz" % spp))
        img, st = sc.render(cam, DEFAULT_CLOCK, 6, 1.0, 64, 64, 0, spp)
        assert np.array_equal(img, g["rgb_sum"])
        assert [st["pixel_samples"], st["rays"], st["shadow_probes"], st["glossy_probes"]] == g["stats"].tolist()
        # C-library math differs from the GL implementation's approximations (acos/asin/atan by up to 1e-4):
        # colours move by that much and a handful of samples take another branch.
        ref, _ = sl.render(cam, DEFAULT_CLOCK, 6, 1.0, 64, 64, 0, spp)
        rel_rmse = np.sqrt(np.mean((img - ref) ** 2)) / np.mean(ref)
        assert rel_rmse < 0.02, rel_rmse
        assert np.mean(np.abs(img - ref) > 2e-3 * np.maximum(1.0, ref)) < 0.02


def test_sum_chaining_stride_and_running_mean(oracle_mod, cubes):
    models, shadings, lights, materials = cubes
    cam = camera_matrix()
    sc = oracle_mod.OracleScene(models, shadings, lights, materials)
    sc.upload_env(gradient_env())
    W, H = 40, 24                      # not a multiple of 8
    full, st = sc.render(cam, DEFAULT_CLOCK, 6, 1.0, W, H, 0, 8)
    # two chained launches add up to the same bits
    part, _ = sc.render(cam, DEFAULT_CLOCK, 6, 1.0, W, H, 0, 3)
    part, _ = sc.render(cam, DEFAULT_CLOCK, 6, 1.0, W, H, 3, 5, rgb=part)
    assert np.array_equal(full, part)
    # an N-way split by SAMPLE mod N sums to the same image up to float association
    even, _ = sc.render(cam, DEFAULT_CLOCK, 6, 1.0, W, H, 0, 4, sample_stride=2)
    odd, _ = sc.render(cam, DEFAULT_CLOCK, 6, 1.0, W, H, 1, 4, sample_stride=2)
    assert np.allclose(even + odd, full, rtol=1e-5, atol=1e-6)
    # the reference's running mean (raytracer.glsl:609-616) equals sum / n up to rounding
    mean, _ = sc.render(cam, DEFAULT_CLOCK, 6, 1.0, W, H, 0, 8, mode="mean")
    assert np.allclose(mean, full / 8.0, rtol=2e-5, atol=1e-6)
    assert st["pixel_samples"] == W * H * 8 and st["rays"] >= st["pixel_samples"]


def test_furnace_bound(oracle_mod):
    # env-lit scene without emitters: radiance never exceeds what the brightest environment texel allows
    models, shadings, lights, materials = load_scene("box")
    assert lights == 0
    sc = oracle_mod.OracleScene(models, shadings, lights, materials)
    sc.upload_env(constant_env())
    img, st = sc.render(camera_matrix(), DEFAULT_CLOCK, 6, 1.0, 32, 32, 0, 4)
    assert st["shadow_probes"] == 0            # sampleLight returns at LIGHTS == 0 without drawing
    assert np.isfinite(img).all() and img.min() >= 0.0


def test_scene_fixture_index():
    import hashlib
    import json
    with open(os.path.join(GOLDEN, "scenes", "index.json")) as fh:
        index = json.load(fh)
    for name, meta in index.items():
        models, shadings, lights, materials = load_scene(name)
        assert len(models) == meta["count"] and lights == meta["lights"] and len(materials) == len(meta["materials"])
        assert hashlib.sha256(models.tobytes()).hexdigest() == meta["models_sha256"]
        assert hashlib.sha256(shadings.tobytes()).hexdigest() == meta["shadings_sha256"]
        # emissive triangles first, the rest by descending |cross(u, v)| (Scene.cpp:319-334)
        ke = materials["emission_ior"][shadings["material_id"]][:, :3]
        is_light = (ke * ke).sum(1) > 0
        assert is_light[:lights].all() and not is_light[lights:].any()
        c = np.cross(models["u"][lights:].astype(np.float64), models["v"][lights:].astype(np.float64))
        area = np.linalg.norm(c, axis=1)
        assert (np.diff(area) <= 1e-6 * area[:-1] + 1e-12).all()
        # true_normal is cross(u, v) in float32 (3Dobjects.hpp:46)
        u, v = models["u"], models["v"]
        tn = np.stack([u[:, 1] * v[:, 2] - v[:, 1] * u[:, 2], u[:, 2] * v[:, 0] - v[:, 2] * u[:, 0],
                       u[:, 0] * v[:, 1] - v[:, 0] * u[:, 1]], axis=1).astype(np.float32)
        assert np.array_equal(tn, models["true_normal"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/res/models"), reason="reference models not present")
def test_fixtures_match_reference_models():
    # the committed fixtures are what the numpy ingest makes of the reference's bundled OBJ/MTL files
    for name in ("cubes", "box", "sphere"):
        models, shadings, lights, materials, _ = scene_np.load_wavefront("/root/reference/res/models/" + name)
        fm, fs, fl, fmat = load_scene(name)
        assert lights == fl and models.tobytes() == fm.tobytes() and materials.tobytes() == fmat.tobytes()
        assert np.array_equal(shadings["material_id"], fs["material_id"])
