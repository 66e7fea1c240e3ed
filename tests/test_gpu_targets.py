"""GPU tests of the accumulation targets and of pass sizing: the low-resolution preview target beside the still one
(Scene::prepare's moving mode, Scene.cpp:449-457), a 3840 x 2160 target with the sample split of a two-GPU run, passes
too large for the device memory, and a closed bright room in which every camera path survives every bounce."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, constant_env, gradient_env, load_scene
from gpuraytracer_amd.camera import camera_matrix, DEFAULT_CLOCK
from test_gpu_parity import _render, _upload, _write_soup
from exr_plain import read_exr_scanlines

pytestmark = pytest.mark.gpu

CAM = camera_matrix()


def _same_bits(a, b):
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


def test_preview_target_leaves_the_still_image_alone(gpu_ctx, oracle_mod, cubes):
    # still image 4 samples -> two preview frames on the other target (RECURSION 3, another camera and clock, the size
    # Scene::prepare derives: 240 rows) -> the still image carries on to 8 samples.  The preview frames equal the
    # checker's; the still sums equal an uninterrupted render of the 8 samples, bit for bit.
    env = gradient_env()
    _upload(gpu_ctx, cubes, env)
    sc = oracle_mod.OracleScene(*cubes)
    sc.upload_env(env)
    W, H = 64, 48
    low_w, low_h = int(240.0 / H * W), 240
    cam2 = camera_matrix((0.3, 1.4, -3.5), (-0.6, 0.2))
    want_still, _ = _render(gpu_ctx, W, H, 0, 8, batch=3)
    gpu_ctx.select_target(1)
    gpu_ctx.resize(low_w, low_h)
    gpu_ctx.select_target(0)
    gpu_ctx.resize(W, H)
    gpu_ctx.set_frame(CAM, DEFAULT_CLOCK, 6, 1.0)
    gpu_ctx.trace(0, 4, 1)
    # the camera moves
    gpu_ctx.select_target(1)
    assert (gpu_ctx.width, gpu_ctx.height) == (low_w, low_h)
    gpu_ctx.set_frame(cam2, 777, 3, 1.0)
    gpu_ctx.clear()
    gpu_ctx.trace(0, 1, 1)
    frame0, n0 = gpu_ctx.read_sum()
    gpu_ctx.trace(1, 1, 1)
    frame01, n01 = gpu_ctx.read_sum()
    want0, _ = sc.render(cam2, 777, 3, 1.0, low_w, low_h, 0, 1)
    want01, _ = sc.render(cam2, 777, 3, 1.0, low_w, low_h, 0, 2)
    assert n0 == 1 and n01 == 2
    assert _same_bits(frame0, want0) and _same_bits(frame01, want01)
    gpu_ctx.clear()                              # SAMPLE 0 again: the preview starts over
    gpu_ctx.trace(0, 1, 1)
    again, n = gpu_ctx.read_sum()
    assert n == 1 and _same_bits(again, want0)
    # the camera stands still again: the still image has kept its four samples
    gpu_ctx.select_target(0)
    assert (gpu_ctx.width, gpu_ctx.height) == (W, H)
    part, n = gpu_ctx.read_sum()
    assert n == 4
    gpu_ctx.set_frame(CAM, DEFAULT_CLOCK, 6, 1.0)
    gpu_ctx.trace(4, 4, 1)
    still, n = gpu_ctx.read_sum()
    assert n == 8 and _same_bits(still, want_still)
    want, _ = sc.render(CAM, DEFAULT_CLOCK, 6, 1.0, W, H, 0, 8)
    assert _same_bits(still, want)
    with pytest.raises(Exception):
        gpu_ctx.select_target(2)
    # back to the state the other tests expect: no preview target
    gpu_ctx.select_target(1)
    gpu_ctx.resize(1, 1)
    gpu_ctx.select_target(0)


def test_host_program_moving_preview(oracle_mod, tmp_path):
    # the C++ host's Scene::prepare(moving) (csrc/host/Scene.cpp) through gpuPT --preview: 3 still samples, 2 preview
    # frames at (240 * W / H) x 240 with RECURSION 3 from a camera further back, 3 more still samples
    from gpuraytracer_amd import host
    base = str(tmp_path / "soup")
    _write_soup(base, 9, 400)
    out = str(tmp_path / "final.exr")
    exe = os.path.join(ROOT, "gpuraytracer_amd", "gpuPT")
    res = subprocess.run([exe, "--preview=2", base, "56", "40", "6", out], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    low_w = int(240.0 / 40 * 56)
    assert "preview %dx240, 2 frames" % low_w in res.stdout
    scene = host.load_wavefront(base)
    sc = oracle_mod.OracleScene(*scene)
    sc.upload_env(constant_env())
    want, _ = sc.render(CAM, DEFAULT_CLOCK, 6, 1.0, 56, 40, 0, 6)
    mean = want * (np.float32(1.0) / np.float32(6.0))
    planes, names, _ = read_exr_scanlines(out)
    got = np.stack([planes["R"], planes["G"], planes["B"]], axis=-1)
    assert _same_bits(got, mean)
    moved = camera_matrix((0.0, 1.5, -3.5), (-np.pi / 4, 0.0))
    want, _ = sc.render(moved, DEFAULT_CLOCK, 3, 1.0, low_w, 240, 0, 2)
    mean = want * (np.float32(1.0) / np.float32(2.0))
    planes, names, _ = read_exr_scanlines(out + ".preview.exr")
    got = np.stack([planes["R"], planes["G"], planes["B"]], axis=-1)
    assert got.shape == (240, low_w, 3) and _same_bits(got, mean)


def test_4k_target_rank_1_of_2(gpu_ctx, oracle_mod, cubes):
    # 3840 x 2160, the samples rank 1 of 2 renders (SAMPLE 1 and 3): the library's own forms against the single kernel
    # that loops over all triangles, bit for bit with equal counters, and SAMPLE 1 alone against the checker
    _upload(gpu_ctx, cubes, constant_env())
    a, sa = _render(gpu_ctx, 3840, 2160, 1, 2, stride=2, pipeline=1, batch=2)
    b, sb = _render(gpu_ctx, 3840, 2160, 1, 2, stride=2, pipeline=0)
    assert _same_bits(a, b)
    for k in ("pixel_samples", "rays", "shadow_probes", "glossy_probes"):
        assert sa[k] == sb[k], k
    assert sa["pixel_samples"] == 3840 * 2160 * 2
    sc = oracle_mod.OracleScene(*cubes)
    sc.upload_env(constant_env())
    want, wst = sc.render(CAM, DEFAULT_CLOCK, 6, 1.0, 3840, 2160, 1, 1)
    one, st = _render(gpu_ctx, 3840, 2160, 1, 1)
    assert _same_bits(one, want) and st["rays"] == wst["rays"]


def test_pass_is_halved_until_the_queues_fit(gpu_ctx, cubes):
    # 512 samples per pass at 3840 x 2160 would take about 530 GB of path queues: rt_trace halves the pass until the
    # queues fit the free device memory.  Same sums as with passes of 24 samples, bit for bit.
    _upload(gpu_ctx, cubes, constant_env())
    small, ss = _render(gpu_ctx, 3840, 2160, 0, 512, pipeline=1, batch=24)
    big, sb = _render(gpu_ctx, 3840, 2160, 0, 512, pipeline=1, batch=512)
    assert _same_bits(small, big)
    for k in ("pixel_samples", "rays", "shadow_probes", "glossy_probes"):
        assert ss[k] == sb[k], k
    per_sample = ss["queue_bytes"] / 24.0
    assert sb["queue_bytes"] > 2 * ss["queue_bytes"]                  # larger passes than 24 samples ...
    assert sb["queue_bytes"] < 300e9 and sb["queue_bytes"] < 0.6 * 512 * per_sample      # ... but not the 512 asked for
    # give the memory back: the other tests size their queues themselves
    _render(gpu_ctx, 64, 64, 0, 1, batch=1, rounds=2)


def _write_closed_room(base):
    """A closed room around the default camera with walls that reflect more than they receive (Kd 6, roughness 1, no
    Fresnel term): every camera path survives every bounce.  One lamp on the ceiling behind the camera."""
    lo, hi = np.array([-4.0, -1.0, -6.0]), np.array([4.0, 4.0, 3.0])
    verts, faces = [], {"Wall": [], "Lamp": []}
    normals = []

    def quad(corners, n_in, name):
        a, b, c, d = [np.array(p, dtype=np.float64) for p in corners]
        for tri in ((a, b, c), (a, c, d)):
            p0, p1, p2 = tri
            if np.dot(np.cross(p1 - p0, p2 - p0), n_in) < 0:
                p1, p2 = p2, p1
            faces[name].append((len(verts) + 1, len(normals) + 1))
            verts.extend([p0, p1, p2])
        normals.append(np.array(n_in, dtype=np.float64))

    for axis in range(3):
        for side, bound in ((0, lo), (1, hi)):
            u, v = [k for k in range(3) if k != axis]
            corners = []
            for du, dv in ((0, 0), (1, 0), (1, 1), (0, 1)):
                p = np.empty(3)
                p[axis] = bound[axis]
                p[u] = (hi if du else lo)[u]
                p[v] = (hi if dv else lo)[v]
                corners.append(p)
            n_in = np.zeros(3)
            n_in[axis] = 1.0 if side == 0 else -1.0
            quad(corners, n_in, "Wall")
    quad([(-1, 3.9, -5.5), (1, 3.9, -5.5), (1, 3.9, -4.5), (-1, 3.9, -4.5)], (0.0, -1.0, 0.0), "Lamp")
    with open(base + ".mtl", "w") as f:
        f.write("newmtl Wall\nNs 1000\nKd 6 6 6\nKs 1 1 1\nKe 0 0 0\n\n")
        f.write("newmtl Lamp\nNs 1000\nKd 1 1 1\nKs 0 0 0\nKe 5 5 5\n\n")
    with open(base + ".obj", "w") as f:
        f.write("mtllib %s.mtl\n" % os.path.basename(base))
        for p in verts:
            f.write("v %.9g %.9g %.9g\n" % tuple(p))
        for n in normals:
            f.write("vn %.9g %.9g %.9g\n" % tuple(n))
        f.write("vt 0 0\n")
        for name in ("Wall", "Lamp"):
            f.write("o %s\nusemtl %s\n" % (name, name))
            for first, n in faces[name]:
                f.write("f %d/1/%d %d/1/%d %d/1/%d\n" % (first, n, first + 1, n, first + 2, n))


def test_every_path_survives(gpu_ctx, oracle_mod, tmp_path):
    # A closed bright room: nearly every camera path goes on to the last bounce, so the queue regions fill to the brim
    # and the waves of the depth-0 kernel stop taking chunks in every round, the last one included.  The pass still
    # renders every chunk (the statistics count the camera paths and fail otherwise), with queues for the whole pass
    # (queue_rounds 1), for a third of it (the default) and for an eighth, same bits as the loop form.
    from gpuraytracer_amd import host
    base = str(tmp_path / "room")
    _write_closed_room(base)
    scene = host.load_wavefront(base)
    assert len(scene[0]) == 14 and scene[2] == 2
    _upload(gpu_ctx, scene, constant_env())
    gpu_ctx.set_option("accel", 1)
    want, wst = _render(gpu_ctx, 1920, 1080, 0, 32, pipeline=3, batch=32, rounds=1)
    assert wst["pixel_samples"] == 1920 * 1080 * 32
    assert wst["depth1_paths"] > 0.97 * wst["pixel_samples"] and wst["rays"] > 5.5 * wst["pixel_samples"]
    for rounds in (1, 3, 8):
        img, st = _render(gpu_ctx, 1920, 1080, 0, 32, pipeline=1, batch=32, rounds=rounds)
        assert _same_bits(img, want), rounds
        for k in ("pixel_samples", "rays", "shadow_probes", "glossy_probes", "depth1_paths"):
            assert st[k] == wst[k], (k, rounds)
        gpu_ctx.reset_stats()
        assert gpu_ctx.stats()["pixel_samples"] == 0
    # a small version against the checker
    sc = oracle_mod.OracleScene(*scene)
    sc.upload_env(constant_env())
    ref, rst = sc.render(CAM, DEFAULT_CLOCK, 6, 1.0, 96, 64, 0, 4)
    img, st = _render(gpu_ctx, 96, 64, 0, 4, pipeline=1, batch=4)
    assert _same_bits(img, ref) and st["rays"] == rst["rays"]


@pytest.mark.parametrize("pipeline", [0, 1])
def test_skip_nonfinite_option(gpu_ctx, oracle_mod, pipeline):
    # the opt-in guard: samples whose radiance is NaN or infinite (here: texels that are Inf / NaN) are left out of the sums
    # and counted; every other sample is added as always
    from conftest import textured_cubes
    S = 64
    (models, shadings, lights, materials), layers, flags = textured_cubes(S, edge_texels=True, nonfinite_texels=True)
    env = gradient_env()
    gpu_ctx.upload_scene(models, shadings, lights, materials)
    gpu_ctx.upload_env(env)
    gpu_ctx.set_atlas_size(S)
    for k in (0, 1, 2):
        gpu_ctx.upload_atlas_layer(2, k, layers[k])
    gpu_ctx.set_has_texture(flags)
    try:
        plain, st = _render(gpu_ctx, 64, 48, 0, 6, clock=31337, pipeline=pipeline, batch=4)
        assert (~np.isfinite(plain)).any() and st["skipped_samples"] == 0
        singles = [_render(gpu_ctx, 64, 48, s, 1, clock=31337, pipeline=pipeline)[0] for s in range(6)]
        gpu_ctx.set_option("skip_nonfinite", 1)
        guarded, sg = _render(gpu_ctx, 64, 48, 0, 6, clock=31337, pipeline=pipeline, batch=4)
        assert np.isfinite(guarded).all()
        want = np.zeros_like(plain)
        skipped = 0
        for one in singles:                                   # ascending SAMPLE order, finite samples only
            ok = np.isfinite(one).all(axis=2)
            skipped += int((~ok).sum())
            want = np.where(ok[..., None], (want + one).astype(np.float32), want)
        assert _same_bits(guarded, want)
        assert sg["skipped_samples"] == skipped > 0
        finite = np.isfinite(plain).all(axis=2)
        assert _same_bits(guarded[finite], plain[finite])
    finally:
        gpu_ctx.set_option("skip_nonfinite", 0)
        gpu_ctx.set_atlas_size(4096)


@pytest.mark.parametrize("name", ["cubes", "_ObiWan"])
def test_tile_candidate_lists_with_odd_cameras(gpu_ctx, oracle_mod, name):
    # The candidate lists of the camera rays (one per 8x8 tile, made once per frame) against the hierarchy (form 9) and the
    # loop over all triangles (form 3), bit for bit, for camera matrices that are nobody's idea of a camera: mirrored, scaled
    # by 1000 and by 1/1000, sheared, a zero column, a zero matrix, a NaN entry, an origin inside the geometry, a camera
    # that looks away from everything; and on targets that are not a whole number of tiles.
    scene = load_scene(name)
    _upload(gpu_ctx, scene, gradient_env())
    rng = np.random.default_rng(77)
    base = camera_matrix((0.2, 1.1, -2.4), (-0.5, 0.3))
    cams = [base]
    m = base.copy(); m[0:3] *= -1.0; cams.append(m)                                   # mirrored x axis
    m = base.copy(); m[0:12] *= 1000.0; cams.append(m)                                # scaled up
    m = base.copy(); m[0:12] *= 0.001; cams.append(m)                                 # scaled down
    m = base.copy(); m[4:7] += 0.7 * m[0:3]; cams.append(m)                           # sheared
    m = base.copy(); m[8:11] *= 0.05; cams.append(m)                                  # very wide field of view
    m = base.copy(); m[8:11] *= 40.0; cams.append(m)                                  # very narrow one
    m = base.copy(); m[4:8] = 0.0; cams.append(m)                                     # zero column: all rays in a plane
    m = np.zeros(16, np.float32); m[12:16] = (0.0, 1.0, -2.0, 1.0); cams.append(m)    # zero directions
    m = base.copy(); m[5] = np.nan; cams.append(m)                                    # NaN in the rotation
    m = base.copy(); m[13] = np.inf; cams.append(m)                                   # origin at infinity
    m = base.copy(); m[12:15] = (0.0, 0.0, 0.0); cams.append(m)                       # origin on the floor plane / inside
    cams.append(camera_matrix((0.0, 1.5, -3.0), (0.9, 3.1)))                          # looks away
    for _ in range(6):
        m = camera_matrix(tuple(rng.uniform([-2, 0.1, -4], [2, 3, 1])), tuple(rng.uniform([-1.2, -3.2], [0.6, 3.2])))
        m[0:12] += rng.normal(0, 0.15, 12).astype(np.float32)
        cams.append(m)
    sizes = [(64, 48), (61, 35), (8, 8), (9, 17), (200, 3)]
    for k, cam in enumerate(cams):
        w, h = sizes[k % len(sizes)]
        want, wst = _render(gpu_ctx, w, h, 0, 3, clock=1000 + k, pipeline=3, batch=2, cam=cam)
        for pipeline in (1, 9, 5):
            img, st = _render(gpu_ctx, w, h, 0, 3, clock=1000 + k, pipeline=pipeline, batch=2, cam=cam)
            assert _same_bits(img, want), (k, pipeline)
            for key in ("pixel_samples", "rays", "shadow_probes", "glossy_probes"):
                assert st[key] == wst[key], (k, pipeline, key)
    # the frame changes between two traces without a resize: the lists follow the camera
    gpu_ctx.set_option("tile_lists", -1)
    gpu_ctx.resize(64, 48)
    acc = np.zeros((48, 64, 3), np.float32)
    for k in (0, 4, 7):
        gpu_ctx.set_frame(cams[k], 555, 6, 1.0)
        gpu_ctx.trace(k, 2, 1)
    got, n = gpu_ctx.read_sum()
    sc = oracle_mod.OracleScene(*scene)
    sc.upload_env(gradient_env())
    for k in (0, 4, 7):
        part, _ = sc.render(cams[k], 555, 6, 1.0, 64, 48, k, 2)
        for s in range(2):
            one, _ = sc.render(cams[k], 555, 6, 1.0, 64, 48, k + s, 1)
            acc = (acc + one).astype(np.float32)
    assert n == 6 and _same_bits(got, acc)
