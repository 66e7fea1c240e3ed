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


def reduce_sums(tensor, dst=0, group=None):
    """Adds the per-rank sum buffers onto rank `dst` (in place).  Call on every rank."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(tensor, dst=dst, op=dist.ReduceOp.SUM, group=group)
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
