"""Sample-budget partition across GPUs and the combine step.

The reference is single-GPU.  Here every rank (one process per GPU) holds the whole scene + BVH and
renders a contiguous slice of the sample indices; RNG streams are a pure function of
(pixel, depth, sample index, seed) (shared.glsl:194-207), so slices are independent.  Each rank keeps
the reference's running mean over ITS samples; equal-sized slices make the job's image the mean of
the per-rank means, obtained with one sum-reduce (RCCL over xGMI on GPUs, gloo in the CPU tests) and
a 1/world scale on the root.  No other data-path collective exists.
"""


def partition_samples(total_spp, world, rank):
    """(first_sample, n_samples) of `rank`: equal slices; total_spp must divide evenly so that the
    mean of per-rank means equals the mean over all samples."""
    if total_spp % world != 0:
        raise ValueError("total_spp=%d is not a multiple of world=%d" % (total_spp, world))
    n = total_spp // world
    return rank * n, n


def combine_accumulation(local_mean, dist, world, root=0):
    """Sum-reduces the per-rank RGBA32F running means onto `root` (in place) and scales by 1/world.
    `local_mean` is a torch tensor (device memory under NCCL/RCCL, host memory under gloo).
    Returns the combined image on the root, None elsewhere.  Alpha is restored to 1."""
    if world == 1:
        return local_mean
    dist.reduce(local_mean, dst=root, op=dist.ReduceOp.SUM)
    if dist.get_rank() != root:
        return None
    local_mean.mul_(1.0 / world)
    local_mean[..., 3] = 1.0
    return local_mean
