"""Multi-GPU plumbing: replicate the scene, split the SAMPLE indices across ranks, sum the per-rank
accumulation buffers with one collective (RCCL over xGMI with backend "nccl", gloo on CPU).

A pixel-sample depends only on (CLOCK, pixel, SAMPLE) and the read-only scene
(raytracer.glsl:594-598 of the reference), so ranks never exchange anything but the final sums."""
import numpy as np


def shard_samples(total_samples, rank, world_size):
    """SAMPLE indices [0, total) dealt round-robin: rank r renders r, r+N, r+2N, ...
    Returns (sample_begin, sample_count, sample_stride) for rt_trace."""
    if not 0 <= rank < world_size:
        raise ValueError("rank %d not in [0, %d)" % (rank, world_size))
    count = (total_samples - rank + world_size - 1) // world_size if total_samples > rank else 0
    return rank, count, world_size


def step_plan(total_samples, rank, world_size, samples_per_step):
    """A fixed image of `total_samples` samples rendered in steps of `samples_per_step` per rank (strong scaling): every
    rank takes the same number of steps, ceil(share of rank 0 / samples_per_step), so that barriers and timing line up;
    a rank whose share is used up renders nothing in the remaining steps.  Returns a list with one
    (sample_begin, sample_count, sample_stride) per step for rt_trace (sample_count may be 0)."""
    if samples_per_step < 1:
        raise ValueError("samples_per_step must be positive")
    _, largest, _ = shard_samples(total_samples, 0, world_size)
    _, mine, _ = shard_samples(total_samples, rank, world_size)
    steps = (largest + samples_per_step - 1) // samples_per_step
    plan = []
    for i in range(steps):
        first = i * samples_per_step
        n = max(0, min(samples_per_step, mine - first))
        plan.append((rank + world_size * first, n, world_size))
    return plan


def reduce_sums(tensor, dst=0, group=None, ordered=True):
    """Adds the per-rank sum buffers onto rank `dst` (in place).  Call on every rank.

    ordered=True (default): the buffers are gathered on `dst` and added there in ascending rank order,
    ((s0 + s1) + s2) + ..., so the result does not depend on the algorithm the collective library picks and is the
    same bits from run to run.  Point-to-point xGMI links make the gather cheap (7 x 133 MB into one GPU at 4K:
    under a millisecond).  ordered=False: one reduce(SUM), whose association order is the library's business."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        return tensor
    if tensor.is_cuda and dist.get_backend(group) == "gloo":
        # gloo does not gather device tensors: the buffers go through host memory (CPU tests and shared-GPU rehearsals)
        host = tensor.cpu()
        reduce_sums(host, dst=dst, group=group, ordered=ordered)
        if dist.get_rank(group) == dst:
            tensor.copy_(host)
        return tensor
    if not ordered:
        dist.reduce(tensor, dst=dst, op=dist.ReduceOp.SUM, group=group)
        return tensor
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    parts = [torch.empty_like(tensor) for _ in range(world)] if rank == dst else None
    dist.gather(tensor, gather_list=parts, dst=dst, group=group)
    if rank == dst:
        acc = parts[0]
        for part in parts[1:]:
            acc += part
        tensor.copy_(acc)
    return tensor


def render_distributed(ctx, total_samples, sum_tensor, rank, world_size, dst=0, group=None):
    """Renders this rank's share of `total_samples` into `sum_tensor` (float32, width*height*4, on the
    context's device, already bound with ctx.set_sum_buffer) and reduces onto `dst`.
    Returns the number of samples this rank rendered."""
    begin, count, stride = shard_samples(total_samples, rank, world_size)
    if count:
        ctx.trace(begin, count, stride)
    ctx.synchronize()
    reduce_sums(sum_tensor, dst=dst, group=group)
    return count


def mean_image(sum_rgba, total_samples, width, height):
    """sum / n as [h, w, 3] (numpy), the image the reference's running mean converges to."""
    a = np.asarray(sum_rgba, dtype=np.float32).reshape(height, width, 4)[..., :3]
    return a / np.float32(total_samples)
