"""The N > 1 path on CPU: two gloo ranks split the SAMPLE indices, each renders its share (the CPU
checker stands in for the GPU renderer here, the product has no CPU path), and one reduce adds the
per-rank sums onto rank 0."""
import os
import socket

import numpy as np
import pytest

from conftest import constant_env, load_scene
from gpuraytracer_amd.camera import camera_matrix, DEFAULT_CLOCK
from gpuraytracer_amd.distributed import mean_image, reduce_sums, shard_samples, step_plan

W, H, SPP = 24, 16, 7          # 7 samples: 4 + 3 on two ranks, 3 + 2 + 2 on three


def test_shard_samples_partition():
    for world in (1, 2, 3, 4, 8):
        for total in (0, 1, 5, 8, 1024, 4096):
            seen = []
            for rank in range(world):
                begin, count, stride = shard_samples(total, rank, world)
                seen += [begin + k * stride for k in range(count)]
            assert sorted(seen) == list(range(total))
    with pytest.raises(ValueError):
        shard_samples(4, 2, 2)


@pytest.mark.parametrize("total,world,per_step", [(4096, 8, 128), (7, 3, 2), (5, 8, 64), (1000, 3, 128), (1, 1, 1)])
def test_step_plan_covers_every_sample_once(total, world, per_step):
    plans = [step_plan(total, r, world, per_step) for r in range(world)]
    assert len({len(p) for p in plans}) == 1                      # every rank takes the same number of steps
    seen = []
    for p in plans:
        for begin, n, stride in p:
            assert 0 <= n <= per_step and stride == world
            seen += [begin + k * stride for k in range(n)]
    assert sorted(seen) == list(range(total))
    # rank 0 is never idle in a step but possibly the last one
    assert all(n == per_step for _, n, _ in plans[0][:-1])


def _worker(rank, world, port, out_path):
    import torch
    import torch.distributed as dist
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = load_scene("cubes")
    sc = oracle.OracleScene(*scene)
    sc.upload_env(constant_env())
    begin, count, stride = shard_samples(SPP, rank, world)
    rgb, _ = sc.render(camera_matrix(), DEFAULT_CLOCK, 6, 1.0, W, H, begin, count, stride)
    sums = torch.zeros(H * W * 4, dtype=torch.float32)
    sums.view(H, W, 4)[..., :3] = torch.from_numpy(rgb)
    reduce_sums(sums, dst=0)
    if rank == 0:
        np.save(out_path, sums.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_split_and_ordered_reduce(tmp_path, oracle_mod, world):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "sum.npy")
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    total = np.load(out).reshape(H, W, 4)[..., :3]
    sc = oracle_mod.OracleScene(*load_scene("cubes"))
    sc.upload_env(constant_env())
    # the reduce adds the per-rank sums in ascending rank order: reproduce exactly that here
    want = None
    for rank in range(world):
        begin, count, stride = shard_samples(SPP, rank, world)
        part, _ = sc.render(camera_matrix(), DEFAULT_CLOCK, 6, 1.0, W, H, begin, count, stride)
        want = part if want is None else (want + part).astype(np.float32)
    assert np.array_equal(total.view(np.uint32), want.view(np.uint32))
    # and against all samples summed in SAMPLE order on one device: same samples, another association of the float sum
    one, _ = sc.render(camera_matrix(), DEFAULT_CLOCK, 6, 1.0, W, H, 0, SPP)
    assert np.allclose(mean_image(np.load(out), SPP, W, H), one / SPP, rtol=1e-5, atol=1e-6)
