"""Lane utilisation of the traversal loops: runs the same frames with the normal library and with one built
with -DRT_COUNT_WAVE_STEPS (GPURT_LIB_WAVE_STEPS) in two child processes and divides the counters.
Usage: python tools/lane_utilisation.py [cubes|obiwan] [recursion]"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch, gpuraytracer_amd as g, bench
name, rec = sys.argv[1], int(sys.argv[2])
scene = g.read_rtscene(os.path.join(bench.ROOT, "tests", "golden", "scenes", {"cubes": "cubes", "obiwan": "_ObiWan"}[name] + ".rtscene"))
ctx = g.Context(0)
ctx.upload_scene(*scene); ctx.upload_env(bench.constant_env()); ctx.resize(bench.W, bench.H)
sums = torch.zeros(bench.H * bench.W * 4, dtype=torch.float32, device="cuda")
ctx.set_sum_buffer(sums.data_ptr())
ctx.set_frame(g.camera_matrix(), g.DEFAULT_CLOCK, rec, 1.0)
ctx.trace(0, 32, 1); torch.cuda.synchronize()
print(json.dumps(ctx.stats()))
''' % ROOT

def run(lib, name, rec):
    env = dict(os.environ)
    if lib:
        env["GPURT_LIB"] = lib
    return json.loads(subprocess.run([sys.executable, "-c", CHILD, name, str(rec)], env=env, check=True,
                                     capture_output=True, text=True).stdout.strip().splitlines()[-1])

name = sys.argv[1] if len(sys.argv) > 1 else "cubes"
rec = int(sys.argv[2]) if len(sys.argv) > 2 else 1
a = run(None, name, rec)
b = run(os.environ.get("GPURT_LIB_WAVE_STEPS", os.path.join(ROOT, "gpuraytracer_amd", "libgpurt_ws.so")), name, rec)
print(name, "RECURSION", rec, "rays", a["rays"], "probes", a["shadow_probes"] + a["glossy_probes"])
for k in ("bvh_extend_nodes", "bvh_extend_tris", "bvh_probe_nodes", "bvh_probe_tris"):
    unit = a["rays"] if "extend" in k else a["shadow_probes"] + a["glossy_probes"]
    print("  %-17s per query %6.2f   lane slots per query %6.2f   utilisation %.2f" % (k, a[k] / unit, b[k] / unit, a[k] / max(1, b[k])))
