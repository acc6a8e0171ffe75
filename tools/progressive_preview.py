"""Interactive multi-GPU preview: every rank accumulates its slice of the samples, and every
`--every` frames the root receives the mean over all samples so far (one asynchronous reduce, the ranks keep
rendering).  Prints the frame rate with and without previews and checks that previews do not change the final
image.  Launch: python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/progressive_preview.py
(one rank per GPU, RCCL); `--backend gloo` stages through host memory and lets several ranks share one GPU
(rehearsal on a 1-GPU box)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--every", type=int, default=8)
    ap.add_argument("--res", default="1920x1080")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0)) if args.backend == "nccl" else 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    torch.cuda.set_device(local)
    dist.init_process_group(args.backend, rank=rank, world_size=world)
    from path_tracer_amd import capi, host, standins
    from path_tracer_amd.multi_gpu import ProgressivePreview, combine_accumulation, partition_samples, render_progressive
    W, H = [int(x) for x in args.res.split("x")]
    assets = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "assets")
    if rank == 0:
        standins.ensure_standins(assets)
    dist.barrier()
    r = host.Renderer(W, H, local)
    r.import_gltf(os.path.join(assets, "sponza/Sponza.gltf"))
    r.set_camera([-8.0, 1.6, 0.0], host.yaw_quat(-90.0), np.deg2rad(np.float32(70.0)), 0.1, 1000.0)
    r.set_seed(0x1234567, fixed=True)
    ctx = r._ctx()
    dev = "cuda:%d" % local
    stage = [torch.zeros((H, W, 4), dtype=torch.float32, device=dev) for _ in range(2)]
    flip = [0]

    def export():
        t = stage[flip[0]]
        flip[0] ^= 1
        ctx.export_albedo_device(t.data_ptr(), t.numel() * 4, 1.0)      # synchronises the context's streams
        return t.cpu() if args.backend == "gloo" else t

    def run(every):
        first, n = partition_samples(world * args.frames, world, rank)
        r.set_params(capi.default_params(rng_sample_offset=first))
        r.resize_render_buffer(W, H)                                        # restart the accumulation
        r.record_commands(); r.record_commands()                            # the two overwritten frames after a change
        ctx.synchronize()
        seen = []
        pv = ProgressivePreview(dist, world, root=0, on_preview=lambda img, s: seen.append(s))
        dist.barrier()
        t0 = time.perf_counter()
        render_progressive(lambda k: [r.record_commands() for _ in range(k)], export, n, every, pv)
        ctx.synchronize()
        dist.barrier()
        dt = time.perf_counter() - t0
        final = export()
        final = combine_accumulation(final, dist, world, root=0)
        return dt, seen, (final.cpu().numpy().copy() if final is not None else None)

    run(args.frames)                                                        # warm-up (RCCL channels, allocations)
    t_plain, _, img_plain = run(args.frames)                                # no preview: one chunk
    t_prev, seen, img_prev = run(args.every)
    if rank == 0:
        same = bool(np.array_equal(img_plain, img_prev))
        print(json.dumps(dict(world=world, backend=args.backend, res=args.res, frames_per_rank=args.frames, preview_every=args.every,
                              previews=len(seen), last_preview_samples=seen[-1] if seen else 0,
                              ms_per_frame_no_preview=round(t_plain / args.frames * 1e3, 3),
                              ms_per_frame_with_preview=round(t_prev / args.frames * 1e3, 3),
                              final_image_unchanged=same)))
        assert same
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
