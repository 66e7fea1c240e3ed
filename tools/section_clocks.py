"""Cycle split of the depth-0 kernel (library built with -DRT_SECTION_CLOCKS): with RECURSION 1 only k_first
runs, and the four hierarchy counters carry the cycles its waves spent in
  closest hit (camera ray + traversal) | shading before the probe | light probe | rest of the bounce + routing.
Usage: GPURT_LIB=.../libgpurt_clk.so python tools/section_clocks.py [cubes|obiwan|c3]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpuraytracer_amd as g
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "cubes"
import argparse
args = argparse.Namespace(workload=name, grid=724, atlas=4096)
scene, cam, recursion, _, layers = bench.load_workload(g, args)
ctx = g.Context(0)
ctx.set_option("profiling", 1)
ctx.upload_scene(*scene)
if layers is not None:
    ctx.set_atlas_size(args.atlas)
    flags = np.zeros(len(scene[3]), dtype=np.int32)
    for m in (0, 1):
        for k in range(3):
            ctx.upload_atlas_layer(m, k, layers[k])
        flags[m] = 1
    ctx.set_has_texture(flags)
    layers = None
ctx.upload_env(bench.constant_env())
ctx.resize(bench.W, bench.H)
sums = torch.zeros(bench.H * bench.W * 4, dtype=torch.float32, device="cuda")
ctx.set_sum_buffer(sums.data_ptr())
ctx.set_frame(cam, g.DEFAULT_CLOCK, 1, 1.0)
ctx.trace(0, 32, 1)
torch.cuda.synchronize()
ctx.reset_stats()
ctx.trace(32, 64, 1)
torch.cuda.synchronize()
st = ctx.stats()
c = np.array([st["bvh_extend_nodes"], st["bvh_extend_tris"], st["bvh_probe_nodes"], st["bvh_probe_tris"]], dtype=np.float64)
print(name, "paths", st["pixel_samples"], "probes", st["shadow_probes"] + st["glossy_probes"], "k_first ms", st["extend0_ms"])
for label, v in zip(("closest hit", "shade_pre", "probe", "finish+route"), c):
    print("  %-13s %5.1f %%   %.0f cycles per wave-tile" % (label, 100 * v / c.sum(), v / (st["pixel_samples"] / 64)))

# Library built with -DRT_SECTION_CLOCKS=3: the counters carry the cycles of the extend stage (bounces after the first)
# instead: getting a block and its records | closest hit | shading and routing | blocks.  Shown when the argument
# "extend" follows the scene name.
if len(sys.argv) > 2 and sys.argv[2] == "extend":
    ctx.set_frame(cam, g.DEFAULT_CLOCK, recursion, 1.0)
    ctx.set_option("batch_samples", 512)
    ctx.trace(0, 512, 1)
    torch.cuda.synchronize()
    ctx.reset_stats()
    ctx.trace(512, 512, 1)
    torch.cuda.synchronize()
    st = ctx.stats()
    c = np.array([st["bvh_extend_nodes"], st["bvh_extend_tris"], st["bvh_probe_nodes"]], dtype=np.float64)
    blocks = float(st["bvh_probe_tris"])
    print(name, "extend stage: blocks", blocks, "extend ms (incl. depth 0)", st["extend_ms"], "depth 0 ms", st["extend0_ms"])
    for label, v in zip(("block + records", "closest hit", "shade + route"), c):
        print("  %-16s %5.1f %%   %.0f cycles per block of 64" % (label, 100 * v / c.sum(), v / blocks))
