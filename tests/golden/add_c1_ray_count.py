"""Adds the ray count of the reference's CPU-runnable configuration (cubes, 512 x 512, SAMPLE 0..15) to
ref_c1_cubes512.json.  The shader does not count its rays; the CPU restatement under oracle/ does, and for this
configuration it reproduces the shader's sixteen images and their running mean digest for digest
(tests/test_reference_glsl.py::test_c1_digests_equal_reference_shader), so its count of closest-hit queries is the
shader's.  bench.py divides it by the llvmpipe time the fixture carries.
    python tests/golden/add_c1_ray_count.py"""
import hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import constant_env, load_scene          # noqa: E402
from gpuraytracer_amd.camera import camera_matrix      # noqa: E402
from oracle import oracle as oracle_mod                # noqa: E402

path = os.path.join(HERE, "ref_c1_cubes512.json")
with open(path) as f:
    c1 = json.load(f)
sc = oracle_mod.OracleScene(*load_scene(c1["scene"]))
sc.upload_env(constant_env())
mean, stats = sc.render(camera_matrix(), c1["clock"], c1["recursion"], 1.0, c1["width"], c1["height"], 0, 16, mode="mean")
assert hashlib.sha256(mean.tobytes()).hexdigest() == c1["sha256_mean_16"], "the restatement no longer reproduces the shader's image"
c1["rays_16spp"] = int(stats["rays"])
c1["rays_16spp_source"] = "closest-hit queries counted by oracle/ while it reproduced sha256_mean_16"
with open(path, "w") as f:
    json.dump(c1, f, indent=1)
print("rays_16spp =", c1["rays_16spp"], "(%.4f per pixel-sample)" % (c1["rays_16spp"] / (c1["width"] * c1["height"] * 16.0)))
