"""Times the brute-force closest-hit loop alone (rt_closest_hit_batch_device) on coherent camera rays
and on incoherent random rays; prints ray-triangle tests per second and the FP32 rate they imply."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpuraytracer_amd as g  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cubes"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1920 * 1080 * 2
    scene = g.read_rtscene(os.path.join(ROOT, "tests", "golden", "scenes", name + ".rtscene"))
    count = len(scene[0])
    ctx = g.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.upload_scene(*scene)
    ctx.set_option("accel", int(os.environ.get("ACCEL", "-1")))
    cam = torch.tensor(g.camera_matrix().reshape(4, 4).T.copy(), device="cuda")
    # coherent: 8x8-tiled camera rays; incoherent: random directions from points near the camera
    idx = torch.arange(n, device="cuda")
    tile, lane = idx // 64, idx % 64
    tx = (tile % 240) * 8 + lane % 8
    ty = ((tile // 240) % 135) * 8 + lane // 8
    v = torch.stack([(tx - 960 + 0.5) / 1080, (ty + 0.5) / 1080 - 0.5, torch.full_like(tx, 0.5, dtype=torch.float32)], 1).float()
    v = v / v.norm(dim=1, keepdim=True)
    d_coh = (v @ cam[:3, :3].T).contiguous()
    o_coh = cam[:3, 3].expand(n, 3).contiguous()
    gen = torch.Generator(device="cuda").manual_seed(1)
    d_inc = torch.randn(n, 3, device="cuda", generator=gen)
    d_inc = (d_inc / d_inc.norm(dim=1, keepdim=True)).contiguous()
    o_inc = (torch.rand(n, 3, device="cuda", generator=gen) * 4 - 2).contiguous()
    ids = torch.empty(n, dtype=torch.int32, device="cuda")
    uvt = torch.empty(n, 3, device="cuda")
    for label, o, d in (("coherent", o_coh, d_coh), ("incoherent", o_inc, d_inc)):
        for _ in range(2):
            ctx.closest_hit_batch_device(o.data_ptr(), d.data_ptr(), n, ids.data_ptr(), uvt.data_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            ctx.closest_hit_batch_device(o.data_ptr(), d.data_ptr(), n, ids.data_ptr(), uvt.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tests = n * count
        print("%s %s: %d rays x %d tris in %.3f ms -> %.3e tests/s, %.1f TFLOP/s (35 flop/test), hit rate %.2f"
              % (name, label, n, count, ms, tests / ms * 1e3, tests * 35 / ms * 1e3 / 1e12, float((ids >= 0).float().mean())))
    ctx.close()


if __name__ == "__main__":
    main()
