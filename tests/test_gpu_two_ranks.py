"""The multi-rank path of bench.py on hardware, with the ranks sharing the GPU there is: `python bench.py --gpus 2`
starts its two ranks itself (torch.distributed.run as a child process), GPURT_BENCH_SHARED_GPU=1 puts both on device 0
and lets them talk over gloo (RCCL refuses two ranks on one device).  Checked: one JSON line from rank 0 for two GPUs,
strong scaling, and the same image as one rank renders (the ranks' sums are added in rank order, so the mean agrees to
rounding).  What N GPUs give in time is for a node that has them."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--steps", "2", "--warmup", "1", "--spp-per-step", "16", "--width", "320", "--height", "184", "--no-cpu-baseline",
        "--lane-utilisation", "0", "--no-bruteforce-pass"]


def _bench(n_gpus, extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n_gpus)] + ARGS, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout            # rank 0's line and nothing else
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_render_the_image_of_one():
    one = _bench(1, {})
    two = _bench(2, {"GPURT_BENCH_SHARED_GPU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["scaling"] == "strong" and two["data"] == "synthetic"
    assert "32 spp split over 2 GPUs" in two["config"]["workload"]
    assert two["nonfinite_pixels"] == one["nonfinite_pixels"]
    # 16 + 16 samples added as (rank 0's sum) + (rank 1's sum) against 32 added one after the other: rounding only
    assert abs(two["mean_radiance"] - one["mean_radiance"]) <= 1e-6 * abs(one["mean_radiance"])
