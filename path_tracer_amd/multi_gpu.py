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


class ProgressivePreview:
    """Periodic combine for interactive use: while every rank keeps accumulating its slice,
    the root shows the mean over all samples rendered so far.

    Every `submit` hands over a COPY of the rank's running mean (a staging tensor: the reduce is in place on
    the root, the accumulation buffers are never touched, so the final image is what `combine_accumulation`
    gives without any preview) taken after the same number of local samples on every rank.  The reduce is
    started asynchronously -- RCCL runs it on its own stream, the ranks go on rendering -- and the root
    picks the finished image up at its next `poll` / `submit`.  One reduce is in flight at a time: a
    `submit` first completes the previous one, so a slow link lowers the preview rate, never the order."""

    def __init__(self, dist, world, root=0, on_preview=None):
        self.dist, self.world, self.root, self.on_preview = dist, world, root, on_preview
        self._work = None
        self._staging = None
        self.latest = None          # root only: the newest finished preview (RGBA32F, alpha 1)
        self.latest_samples = 0     # samples per pixel (over all ranks) that `latest` holds
        self._pending_samples = 0
        self.submitted = 0

    def _finish(self):
        if self._staging is None:
            return
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self.world == 1 or self.dist.get_rank() == self.root:
            img = self._staging
            if self.world > 1:
                img.mul_(1.0 / self.world)
                img[..., 3] = 1.0
            self.latest, self.latest_samples = img, self._pending_samples
            if self.on_preview is not None:
                self.on_preview(img, self._pending_samples)       # every finished preview, in order
        self._staging = None

    def submit(self, local_mean_copy, local_samples):
        """`local_mean_copy`: this rank's running mean after `local_samples` samples (the same count on every
        rank), in a tensor the preview may overwrite."""
        self._finish()
        self._staging = local_mean_copy
        self._pending_samples = local_samples * self.world
        self.submitted += 1
        if self.world > 1:
            self._work = self.dist.reduce(local_mean_copy, dst=self.root, op=self.dist.ReduceOp.SUM, async_op=True)

    def poll(self):
        """Root: the newest preview whose reduce has completed (None before the first); never blocks."""
        if self._staging is not None and (self._work is None or self._work.is_completed()):
            self._finish()
        return self.latest

    def wait(self):
        self._finish()
        return self.latest


def render_progressive(render, export, n_local, every, preview):
    """Renders `n_local` samples on this rank in chunks of `every`, feeding `preview` between chunks.
    `render(k)` accumulates k more samples; `export()` returns a copy of the running mean (device tensor under
    RCCL, host tensor under gloo).  The root's `preview.on_preview(image, samples)` sees every finished
    preview in order.  The last chunk is not previewed: the caller ends with `combine_accumulation`."""
    done = 0
    while done < n_local:
        k = min(every, n_local - done)
        render(k)
        done += k
        preview.poll()
        if done < n_local:
            preview.submit(export(), done)
    preview.wait()
    return done
