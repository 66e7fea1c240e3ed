"""Searches of the depth-0 kernel by the number of lanes they run with (library built with -DRT_PROBE_HISTO=1 for
the shadow searches, =2 for the glossy ones): with RECURSION 1 only the depth-0 kernel and the probe stages of depth 0
run, and the four hierarchy counters carry the numbers of searches with 1-15, 16-31, 32-47 and 48-64 lanes.
Usage: GPURT_LIB=.../histo1.so python tools/probe_histogram.py [cubes|obiwan] [batch]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpuraytracer_amd as g
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "cubes"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 512
scene = g.read_rtscene(os.path.join(bench.ROOT, "tests", "golden", "scenes", {"cubes": "cubes", "obiwan": "_ObiWan"}[name] + ".rtscene"))
ctx = g.Context(0)
ctx.set_option("profiling", 1)
ctx.set_option("batch_samples", batch)
ctx.upload_scene(*scene)
ctx.upload_env(bench.constant_env())
ctx.resize(bench.W, bench.H)
sums = torch.zeros(bench.H * bench.W * 4, dtype=torch.float32, device="cuda")
ctx.set_sum_buffer(sums.data_ptr())
ctx.set_frame(g.camera_matrix(), g.DEFAULT_CLOCK, 1, 1.0)
ctx.trace(0, batch, 1)
torch.cuda.synchronize()
ctx.reset_stats()
ctx.trace(batch, batch, 1)
torch.cuda.synchronize()
st = ctx.stats()
c = np.array([st["bvh_extend_nodes"], st["bvh_extend_tris"], st["bvh_probe_nodes"], st["bvh_probe_tris"]], dtype=np.float64)
chunks = st["pixel_samples"] / 64.0
print(name, "chunks %.0f" % chunks, "shadow probes", st["shadow_probes"], "glossy probes", st["glossy_probes"], "parked", st["parked_probes"],
      "left to the probe stages", st["deferred_probes"], "k_first ms", st["extend0_ms"], "probe stages ms", st["probe_ms"])
if len(sys.argv) > 3 and sys.argv[3] == "lists":        # library built with -DRT_PROBE_HISTO=3
    for label, v in zip(("up to 4 triangles", "5-12", "13-31", "no list (hierarchy)"), c):
        print("  camera-ray chunks with a list of %-20s %6.2f %%" % (label, 100 * v / chunks))
else:
    for label, v in zip(("1-15 lanes", "16-31", "32-47", "48-64"), c):
        print("  searches with %-10s %6.2f %% of the chunks" % (label, 100 * v / chunks))
