"""Lane utilisation of the traversal loops: renders the same frames with the two measurement builds of the library,
-DRT_COUNT_STEPS (gpuraytracer_amd/libgpurt_steps.so: every lane counts its node visits and triangle tests) and
-DRT_COUNT_WAVE_STEPS (libgpurt_wavesteps.so: the lane slots a wave spends on them), each in a child process, and
divides the counters.
Usage: python tools/lane_utilisation.py [cubes|obiwan|c3] [recursion]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = os.path.join(ROOT, "gpuraytracer_amd", "libgpurt_steps.so")
VARIANT = os.path.join(ROOT, "gpuraytracer_amd", "libgpurt_wavesteps.so")
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, gpuraytracer_amd as g, bench
import argparse
name, rec = sys.argv[1], int(sys.argv[2])
import numpy as np
# the scene, camera and (c3) texture layers of the bench's workload
atlas = int(sys.argv[4])
scene, cam, _, _, layers = bench.load_workload(g, argparse.Namespace(workload=name, grid=int(sys.argv[3]), atlas=atlas))
ctx = g.Context(0)
ctx.upload_scene(*scene)
if layers is not None:
    ctx.set_atlas_size(atlas)
    flags = np.zeros(len(scene[3]), dtype=np.int32)
    for m in (0, 1):
        for k in range(3):
            ctx.upload_atlas_layer(m, k, layers[k])
        flags[m] = 1
    ctx.set_has_texture(flags)
ctx.upload_env(bench.constant_env()); ctx.resize(bench.W, bench.H)
sums = torch.zeros(bench.H * bench.W * 4, dtype=torch.float32, device="cuda")
ctx.set_sum_buffer(sums.data_ptr())
ctx.set_frame(cam, g.DEFAULT_CLOCK, rec, 1.0)
ctx.trace(0, 32, 1); torch.cuda.synchronize()
print(json.dumps(ctx.stats()))
''' % ROOT


def _run(lib, name, rec, grid, atlas):
    env = dict(os.environ)
    if lib:
        env["GPURT_LIB"] = lib
    out = subprocess.run([sys.executable, "-c", CHILD, name, str(rec), str(grid), str(atlas)], env=env, check=True, capture_output=True, text=True)
    return json.loads(out.stdout.strip().splitlines()[-1])


def measure(name, recursion, grid=724, atlas=4096):
    """{counter: (steps per query, lane slots per query, utilisation)} for 32 spp at 1080p.  `grid`, `atlas`: cells per
    side of the c3 workload's height field and the edge of its texture layers."""
    a = _run(os.environ.get("GPURT_LIB_STEPS", STEPS), name, recursion, grid, atlas)
    b = _run(os.environ.get("GPURT_LIB_WAVE_STEPS", VARIANT), name, recursion, grid, atlas)
    res = {"rays": a["rays"], "probes": a["shadow_probes"] + a["glossy_probes"]}
    for k in ("bvh_extend_nodes", "bvh_extend_tris", "bvh_probe_nodes", "bvh_probe_tris"):
        unit = max(1, a["rays"] if "extend" in k else res["probes"])
        res[k] = {"per_query": a[k] / unit, "lane_slots_per_query": b[k] / unit, "utilisation": a[k] / max(1, b[k]),
                  "steps": a[k], "lane_slots": b[k]}
    res["pixel_samples"] = a["pixel_samples"]
    return res


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "cubes"
    rec = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    r = measure(name, rec)
    print(name, "RECURSION", rec, "rays", r["rays"], "probes", r["probes"])
    for k in ("bvh_extend_nodes", "bvh_extend_tris", "bvh_probe_nodes", "bvh_probe_tris"):
        print("  %-17s per query %6.2f   lane slots per query %6.2f   utilisation %.2f"
              % (k, r[k]["per_query"], r[k]["lane_slots_per_query"], r[k]["utilisation"]))
