#!/usr/bin/env python
"""bench.py -- headline benchmark: samples/s (and Mrays/s) on the Sponza 1920x1080 config.

  python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed / RCCL when N > 1, launched by torch.distributed.run).
A "step" = one pass of the hot path over one batch: --spp-per-step (default 32) samples per pixel of
the whole 1920x1080 image = 66.4 M pixel-samples, through the reference-mirroring host
(path_tracer_amd.host.Renderer -> C ABI -> HIP kernels).  Multi-GPU is the spp partition of
BASELINE.json: every rank holds the full scene + BVH, renders its own sample indices
(rng_sample_offset) into its own accumulation buffer, and the buffers are combined by ONE RCCL
reduce at the end of the timed region (weak scaling: per-GPU work fixed).

Rank 0 prints one JSON line; see DESIGN.md "Measurement" for every field.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x1234567
WORKLOADS = {
    # scene, (gltf, offset) list, camera pos, yaw deg, fov deg, resolution
    "sponza_1080p": dict(items=[("sponza/Sponza.gltf", (0, 0, 0))], pos=[-8.0, 1.6, 0.0], yaw=-90.0, fov=70.0,
                         res=(1920, 1080), geometry="procedural stand-in (Sponza.bin is missing from the reference checkout)"),
    # BASELINE.json config 5 (the 4096-spp convergence run is this workload repeated; tests/ hold the converged-image check)
    "sponza_4k": dict(items=[("sponza/Sponza.gltf", (0, 0, 0))], pos=[-8.0, 1.6, 0.0], yaw=-90.0, fov=70.0,
                      res=(3840, 2160), geometry="procedural stand-in (Sponza.bin is missing from the reference checkout)"),
    "spheres_720p": dict(items=[("spheres/spheres.gltf", (0, 0, 0))], pos=[7.0, 2.0, 0.0], yaw=90.0, fov=70.0,
                         res=(1280, 720), geometry="procedural stand-in (spheres.bin is missing from the reference checkout)"),
    "lion_boulder_1080p": dict(items=[("lion_head_2k/lion_head_2k.gltf", (2, 0, 0)), ("boulder_01_2k/boulder_01_2k.gltf", (-2, 0, 0))],
                               pos=[2.0, 0.25, 0.6], yaw=0.0, fov=70.0, res=(1920, 1080), geometry="real"),
    # deep incoherent paths through real, non-axis-aligned triangles: the two scanned meshes of the reference's app
    # (render_window.rs:412-426) inside a closed room (tools/make_room.py), camera inside: no path ends in the sky
    # (further instances of the same two geometries -- ident dedup, geometry/mod.rs:62-65 -- fill the floor, so that a
    # bounce meets scanned triangles about as often as a wall)
    "lion_boulder_room_1080p": dict(items=[("room/room.gltf", (0, 0, 0)), ("lion_head_2k/lion_head_2k.gltf", (2, 0, 0)),
                                           ("boulder_01_2k/boulder_01_2k.gltf", (-2, 0, 0)),
                                           ("boulder_01_2k/boulder_01_2k.gltf", (0.3, 0, -0.25)),
                                           ("lion_head_2k/lion_head_2k.gltf", (1.35, 0, 0.55)), ("lion_head_2k/lion_head_2k.gltf", (1.5, 0, -0.7)),
                                           ("lion_head_2k/lion_head_2k.gltf", (-0.85, 0, 0.95)), ("lion_head_2k/lion_head_2k.gltf", (0.75, 0, 0.95)),
                                           ("lion_head_2k/lion_head_2k.gltf", (2.2, 0, 0.9))],
                                    pos=[0.3, 0.55, 1.45], yaw=0.0, fov=70.0, res=(1920, 1080), geometry="real (scanned meshes) in a generated closed room"),
    # the Sponza stand-in rotated 30 degrees about +Y with the camera rotated along: the same image content, every
    # triangle oblique to the world axes (looser boxes)
    "sponza_rot30_1080p": dict(items=[("sponza/Sponza_rot30.gltf", (0, 0, 0))], pos=[-6.928203, 1.6, 4.0], yaw=-60.0, fov=70.0,
                               res=(1920, 1080), geometry="procedural stand-in, rotated 30 degrees about +Y"),
    "cornell_512": dict(items=[("cornell/cornell.gltf", (0, 0, 0))], pos=[0.0, 1.0, 3.9], yaw=0.0, fov=40.0,
                        res=(512, 512), geometry="real"),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_PEAK_GBS = 34500.0          # MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate
VALU_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md "Peak FP32 (vector)"
LANE_OPS_NODE, LANE_OPS_TRI = 112.0, 60.0      # algorithmic lane-ops per 8-wide node visit / per watertight triangle test
PHASE_NAMES = ["ray fetch", "node-load wait", "node test + pick", "queue append", "triangle fetch wait", "triangle test",
               "hit hand-back", "retire + loop"]
# per node visit, per triangle test (48 B of vertices; the ray direction it needs is in LDS since round 2), ray fetch (origin +
# direction planes), hit record
NODE_BYTES, TRI_BYTES, RAY_BYTES, HIT_BYTES = 80, 48, 32, 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=32)
    ap.add_argument("--workload", default="sponza_1080p", choices=sorted(WORKLOADS))
    ap.add_argument("--max-depth", type=int, default=16)
    ap.add_argument("--integrator", default="reference", choices=["reference", "lit"],
                    help="reference: the reference's rchit/rmiss shading (the headline number); lit: the build's "
                         "extension with emissive triangles + next-event estimation + GGX/glass BSDFs (SURVEY 8f.3)")
    ap.add_argument("--cpu-baseline-spp", type=int, default=-1,
                    help="samples per pixel of the CPU leg; -1 (default): about 8.3 M pixel-samples (4 spp at 1080p, "
                         "roughly 20 s on 16 cores for Sponza); 0 disables the CPU leg")
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="rehearsal of the N > 1 path on a box with ONE GPU: every rank renders on device 0 and the "
                         "accumulation reduce goes through host memory over gloo (RCCL refuses two ranks on one device)")
    ap.add_argument("--total-spp", type=int, default=0,
                    help="fixed-budget mode (BASELINE.json configs 3 and 5: '1024 spp partitioned over 8 GPUs'): the JOB renders "
                         "this many samples per pixel in all, split over the ranks by partition_samples, and the timed region is the "
                         "time to the combined image, reduce included -> \"scaling\": \"strong\".  --steps is then derived "
                         "(ceil(rank's share / --spp-per-step) batches, the last one short); 0 (default): weak scaling, --steps "
                         "batches of --spp-per-step on every rank")
    ap.add_argument("--traffic-json", default=None,
                    help="PMC-derived HBM traffic of the traversal kernel (default: profiles/traffic.json, or traffic_lit.json for "
                         "--integrator lit); valu*.json is read from the same directory")
    ap.add_argument("--force-dist", action="store_true",
                    help="with --gpus 1: bring up the RCCL process group for the one rank anyway, so that every collective of the "
                         "N > 1 path (barrier, reduce of the accumulation buffer, gathers of the per-rank timings) really runs "
                         "through RCCL on a one-GPU box; the line then says n_gpus 1 with a collective_backend")
    ap.add_argument("--launch-check", action="store_true",
                    help="only bring up the N ranks, check the world size with one all-reduce and print a JSON line; no "
                         "GPU is touched when combined with --rehearse-gloo (the CPU test of the launcher path)")
    ap.add_argument("--verify-reduce", action="store_true",
                    help="after the timed region rank 0 renders every rank's samples on its own GPU and reports the "
                         "rel-MSE of the combined image against it (small workloads only)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as typed: this process becomes the launcher.  It has not imported torch and has
        # made no HIP call; the N ranks run under torch.distributed.run as a CHILD process whose output and exit
        # code are passed on.
        raise SystemExit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    dist = None
    if world == 1 and args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        if args.rehearse_gloo:
            local_rank = 0
            if not args.launch_check:
                torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    if args.launch_check:
        ranks_seen = 1
        if dist is not None:
            t = torch.ones(1, dtype=torch.float64, device="cpu" if args.rehearse_gloo else "cuda:%d" % local_rank)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            ranks_seen = int(t.item())
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": args.gpus, "rccl_ranks": ranks_seen,
                              "backend": "gloo" if args.rehearse_gloo else "nccl"}), flush=True)
        if ranks_seen != args.gpus:
            raise SystemExit(1)
        return

    from path_tracer_amd import capi, host, standins
    wl = WORKLOADS[args.workload]
    assets = os.path.join(ROOT, "assets")
    if rank == 0 or (local_rank == 0 and not args.rehearse_gloo):
        standins.ensure_standins(assets)
    if dist is not None:
        dist.barrier()

    W, H = wl["res"]
    limits = capi.default_limits(max_vertices=10 ** 7, max_indices=3 * 10 ** 7)
    t0 = time.time()
    r = host.Renderer(W, H, local_rank, limits)
    lit = args.integrator == "lit"
    if lit:
        r.set_ref_quirks(False)       # world-space normals, texel x factor, sampler wrap modes, child nodes
    for rel, off in wl["items"]:
        r.import_gltf(os.path.join(assets, rel), off)
    if lit:
        for row, ext in lit_emitters(args.workload, r.scene_arrays()[2]):
            r._ctx().set_material_ext(row, ext)
    r.set_camera(wl["pos"], host.yaw_quat(wl["yaw"]), np.deg2rad(np.float32(wl["fov"])), 0.1, 1000.0)
    r.set_seed(SEED, fixed=True)
    from path_tracer_amd.multi_gpu import combine_accumulation, partition_samples
    spp = args.spp_per_step
    strong = args.total_spp > 0
    if strong:
        # fixed budget: the job renders --total-spp samples per pixel, rank g the slice partition_samples gives it (the
        # warm-up batches use sample indices past the budget and are overwritten: the accumulation restarts below)
        first_sample, my_spp = partition_samples(args.total_spp, world, rank)
        batches = [spp] * (my_spp // spp) + ([my_spp % spp] if my_spp % spp else [])
        job_budget = args.total_spp
    else:
        # the job's budget is world x (steps + warmup) x spp samples per pixel; rank g renders slice g of it
        job_budget = world * (args.steps + args.warmup) * spp
        first_sample, _ = partition_samples(job_budget, world, rank)
        batches = [spp] * args.steps
    params = capi.default_params(max_depth=args.max_depth, rng_sample_offset=first_sample,
                                 integrator=capi.INTEGRATOR_LIT if lit else capi.INTEGRATOR_REFERENCE)
    r.set_params(params)
    ctx = r._ctx()
    for _ in range(args.warmup):
        r.record_commands_batch(spp)
    ctx.synchronize()
    load_s = time.time() - t0
    if strong:
        # time-to-image starts from an empty accumulation: the image of the timed region is the job's image
        r.resize_render_buffer(W, H)  # resized_dirty: the next frame restarts the accumulation (renderer.rs:583-666, 988-1000)
        r.record_commands()           # ... that dirty frame: 1 spp, overwritten by the first clean one
        ctx.synchronize()
        assert r.accumulated_count == 0, r.accumulated_count

    red = None
    if dist is not None:
        red = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:%d" % local_rank)
        # untimed warm-up of the one collective the job uses: RCCL sets up its channels for this message size on
        # first use (the timed region then measures the reduce itself, 33 MB at 1080p)
        if not args.rehearse_gloo:
            dist.reduce(red, dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()

    # ---------------- timed region: exactly K steps + the one accumulation reduce ----------------
    ctx.set_profiling(1)
    ctx.reset_stats()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.synchronize()
    t_start = time.perf_counter()
    for b in batches:
        r.record_commands_batch(b)
    t_render = None
    if dist is not None:
        ctx.export_albedo_device(red.data_ptr(), red.numel() * 4, 1.0)      # (synchronises the context: every batch has finished)
        t_render = time.perf_counter() - t_start
        if args.rehearse_gloo:
            host_img = red.cpu()
            combined = combine_accumulation(host_img, dist, world, root=0)
            if combined is not None:
                red.copy_(combined)
                combined = red
        else:
            combined = combine_accumulation(red, dist, world, root=0)
        if combined is not None:
            torch.cuda.synchronize()
            ctx.import_albedo_device(combined.data_ptr(), combined.numel() * 4, 1.0)
    ctx.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if t_render is None:
        t_render = elapsed
    st = ctx.stats()
    ctx.set_profiling(0)
    # the line decomposed per rank: scene load + warm-up, rendering alone, the accumulation reduce (export, collective, import)
    mine = [load_s, t_render * 1e3, (elapsed - t_render) * 1e3, float(sum(batches))]
    per_rank = [mine]
    if dist is not None:
        # (plain tensors, like the other collectives of this script: nothing is pickled through the device)
        dev = "cpu" if args.rehearse_gloo else "cuda:%d" % local_rank
        m = torch.tensor(mine, dtype=torch.float64, device=dev)
        parts = [torch.zeros_like(m) for _ in range(world)]
        dist.all_gather(parts, m)
        per_rank = [[float(v) for v in p.cpu()] for p in parts]

    if dist is not None:
        stat_dev = "cpu" if args.rehearse_gloo else "cuda:%d" % local_rank
        t = torch.tensor([elapsed], dtype=torch.float64, device=stat_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([st["samples"], st["rays"]], dtype=torch.float64, device=stat_dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        job_samples, job_rays = float(cnt[0].item()), float(cnt[1].item())
    else:
        job_samples, job_rays = float(st["samples"]), float(st["rays"])

    reduce_check = None
    if args.verify_reduce and rank == 0:
        # The combined image (now in rank 0's accumulation buffer) against ONE context rendering every rank's slice:
        # rank g holds the running mean of samples first_sample(g) + [0, accumulated_count) -- the first two frames of a
        # Renderer are dirty frames (renderer.rs:988-1000) that restart the accumulation, so the count is read back.
        combined_img = ctx.read_image(capi.IMAGE_ALBEDO)
        n_acc = int(r.accumulated_count)
        gcam = host.camera_matrices(wl["pos"], host.yaw_quat(wl["yaw"]), np.deg2rad(np.float32(wl["fov"])), W / H, 0.1, 1000.0)
        acc = np.zeros((H, W, 4), np.float64)
        for g in range(world):
            ctx.resize(W, H)
            first_g, _ = partition_samples(job_budget, world, g)
            ctx.render_batch(capi.Uniforms(1, SEED, 0, 0.1, 100.0, (0, 0, 0)), gcam, n_acc,
                             capi.default_params(max_depth=args.max_depth, rng_sample_offset=first_g,
                                                 integrator=capi.INTEGRATOR_LIT if lit else capi.INTEGRATOR_REFERENCE))
            acc += ctx.read_image(capi.IMAGE_ALBEDO)
        a = np.asarray(combined_img, np.float64)[..., :3]
        b = (acc / world)[..., :3]
        reduce_check = {"samples_per_rank": n_acc, "ranks": world,
                        "rel_mse_vs_single_context": float(np.mean((a - b) ** 2 / (b * b + 1e-2))),
                        "max_abs_diff": float(np.max(np.abs(a - b))), "rel_mse_eps": 1e-2}

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel (k_extend, BVH traversal) ----
        # One untimed instrumented step of the same workload (profiling level 2: counting traversal kernel, every chunk
        # on one stream): nodes fetched / triangles tested per ray, lane occupancy of the wave-level blocks, per-phase
        # wave cycles, and the shading kernels' own time (in the timed region two chunks overlap on two streams, so a
        # stage's HIP-event time there includes waiting behind the other chunk's kernels).
        ctx.set_profiling(2)
        ctx.reset_stats()
        r.record_commands_batch(spp)
        ist = ctx.stats()
        ctx.set_profiling(0)
        nodes_per_ray = ist["nodes_fetched"] / max(1, ist["rays"])
        tris_per_ray = ist["tris_tested"] / max(1, ist["rays"])
        ws = ist["wave_sched"]          # wave iterations, node blocks, lanes in them, triangle blocks, lanes in them, refills
        divergence = dict(node_block_lanes=round(ws[2] / max(1, ws[1]) / 64.0, 4), tri_block_lanes=round(ws[4] / max(1, ws[3]) / 64.0, 4),
                          node_blocks_per_iteration=round(ws[1] / max(1, ws[0]), 4), tri_blocks_per_iteration=round(ws[3] / max(1, ws[0]), 4),
                          source="instrumented k_extend<stats> on one untimed step of the same workload")
        ea = ist["extend_append"]
        wave_counters = dict(rays=int(ist["rays"]), waves=int(ctx.get_option(capi.OPTION_EXTEND_WAVES)) * args.max_depth,
                             iterations=int(ws[0]), node_blocks=int(ws[1]), node_lanes=int(ws[2]), tri_blocks=int(ws[3]), tri_lanes=int(ws[4]),
                             refills=int(ws[5]), append_blocks=int(ea[0]), append_passes=int(ea[1]), refill_passes=int(ea[2]),
                             source="instrumented k_extend<stats>, one untimed step (16 launches of the reference integrator): the weights of tools/isa_budget.py")
        ph = ist["extend_phase_cycles"]
        ph_tot = float(sum(ph)) or 1.0
        phases = {n: round(c / ph_tot, 4) for n, c in zip(PHASE_NAMES, ph)}
        launches = max(1, st["extend_launches"])
        rays_per_launch = st["rays"] / launches
        avg_launch_s = st["extend_ms"] * 1e-3 / launches        # HIP events on the chunk's own stream around every k_extend launch of the timed region
        # what binds the kernel is vector-ALU work, not memory: algorithmic lane-operations per ray
        # from SURVEY.md 8d -- 112 per 8-wide node visit, 60 per triangle test -- against the fp32 vector peak
        # (157.3 TFLOP/s = 78.65 T lane-FMAs/s, MI355X_MICROARCH.md "Peak FP32 (vector)"), one lane-op counted as one FMA slot = 2 FLOP
        lane_ops_per_ray = nodes_per_ray * LANE_OPS_NODE + tris_per_ray * LANE_OPS_TRI
        achieved_tflops = lane_ops_per_ray * 2.0 * rays_per_launch / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
        bytes_per_ray = nodes_per_ray * NODE_BYTES + tris_per_ray * TRI_BYTES + RAY_BYTES + HIT_BYTES
        req_gbs = bytes_per_ray * rays_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # Counter-derived side fields come from the committed rocprofv3 --pmc summaries (profiles/traffic*.json,
        # valu*.json; PMC passes cannot run inside a timed bench).  They are only quoted when they were measured on THESE
        # sources and THIS workload: the summaries carry a hash of path_tracer_amd/csrc + include/ptx.h
        # (tools/profile_round_summary.py); after any kernel change the line says "profile_stale": true and the
        # fields are null until tools/profile_round.sh has been re-run.
        from path_tracer_amd import build as pbuild
        src_hash = pbuild.csrc_hash()
        wl_key = args.workload + ("+lit" if lit else "")
        suffix = "_lit" if lit else ""
        traffic_json = args.traffic_json or os.path.join(ROOT, "profiles", "traffic%s.json" % suffix)
        traffic = None
        tcc_hit = None
        shade_hbm_per_ray = None
        profile_stale = None        # None: no summary for this workload; True: one exists, measured on other sources
        if os.path.exists(traffic_json):
            try:
                tj = json.load(open(traffic_json))
                if tj.get("workload") == wl_key:
                    profile_stale = tj.get("csrc_sha16") != src_hash
                    if not profile_stale:
                        # PMC passes run on a shorter job; scale the measured HBM-side bytes per ray to this run's launch size
                        traffic = round(tj["hbm_bytes_per_ray"] * rays_per_launch, 1)
                        tcc_hit = tj.get("tcc_hit_rate")
                        shade_hbm_per_ray = tj.get("shade_hbm_bytes_per_ray")
            except Exception:
                traffic = None
        valu = None
        valu_shade = None
        vj = os.path.join(os.path.dirname(traffic_json), "valu%s.json" % suffix)
        if os.path.exists(vj):
            try:
                d = json.load(open(vj))
                if d.get("workload") == wl_key:
                    stale_v = d.get("csrc_sha16") != src_hash
                    profile_stale = bool(profile_stale) or stale_v
                    if not stale_v:
                        def pick(k):
                            v = {x: (round(y, 4) if isinstance(y, float) else y) for x, y in k.items() if x != "counters"}
                            v["source"] = d.get("source", "profiles/valu.json")
                            return v
                        valu = pick(d["kernels"]["k_extend<false, false>"])
                        valu["definitions"] = d.get("definitions")
                        sk = d["kernels"].get("k_shade_lit" if lit else "k_shade")
                        valu_shade = pick(sk) if sk else None
            except Exception:
                valu = None
        # HBM view: the traversal-free floor of SURVEY.md 8d (reference data layouts: 212 B (+4 B texel) per hit ray, 32 B of
        # accumulation per sample) against 8 TB/s, and the measured HBM-side bytes of the kernel (PMC) against the same peak
        floor_bytes_per_sample = (st["hits"] * 216.0 + st["samples"] * 32.0) / max(1, st["samples"])
        samples_per_s = job_samples / elapsed
        hbm = dict(peak=HBM_PEAK_GBS, unit="GB/s",
                   floor_bytes_per_sample=round(floor_bytes_per_sample, 1),
                   floor_achieved=round(floor_bytes_per_sample * samples_per_s / world / 1e9, 2),
                   floor_frac=round(floor_bytes_per_sample * samples_per_s / world / 1e9 / HBM_PEAK_GBS, 5),
                   counter_bytes_per_launch=traffic,
                   counter_achieved=round(traffic / avg_launch_s / 1e9, 2) if traffic and avg_launch_s > 0 else None,
                   counter_frac=round(traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS, 5) if traffic and avg_launch_s > 0 else None,
                   note="the 48 MB scene + BVH are served from L2 / Infinity Cache: HBM does not bound this kernel")
        l2 = dict(peak=L2_PEAK_GBS, unit="GB/s", requested_bytes_per_ray=round(bytes_per_ray, 1), requested=round(req_gbs, 2),
                  frac=round(req_gbs / L2_PEAK_GBS, 5), tcc_hit_rate=tcc_hit,
                  note="bytes the kernel asks of the memory pipe (nodes x 80 + triangles x 48 + 40 per ray); part is served by the CU's L1")
        roofline = dict(bound="valu", kernel="k_extend", achieved=round(achieved_tflops, 3), peak=VALU_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=round(achieved_tflops / VALU_PEAK_TFLOPS, 5), traffic=traffic,
                        algorithmic_lane_ops_per_ray=round(lane_ops_per_ray, 1), nodes_per_ray=round(nodes_per_ray, 2),
                        tris_per_ray=round(tris_per_ray, 2), rays_per_launch=round(rays_per_launch, 1),
                        avg_launch_ms=round(avg_launch_s * 1e3, 4), launches=int(launches),
                        extend_ms=round(st["extend_ms"], 2),
                        valu_issue=valu, hbm=hbm, l2=l2, divergence=divergence, wave_lifetime_by_phase=phases, wave_counters=wave_counters,
                        note="bound = vector ALU: `frac` prices SURVEY 8d's algorithmic lane-ops (112 per node visit, 60 per triangle "
                             "test; 1 lane-op = 1 FMA slot = 2 FLOP) against the 157.3 TFLOP/s fp32 vector peak.  The instructions the "
                             "kernel really issues, priced with the per-class issue costs measured on this chip "
                             "(profiles/r02_valu_calibration.json), are in `valu_issue`; HBM and L2 are side fields: neither binds.")

        # second kernel: k_shade streams the ray queues (HBM) and gathers one 128-byte record per hit (L2 / Infinity Cache).
        # algorithmic bytes: ray 48 + hit 8 read per ray; ShadeRec 128 (+4 texel) per hit ray; 48 written per surviving
        # ray; 16 per terminated path (every path terminates once: survivors = rays - samples).  Time: the shading
        # launches of the serial instrumented step above (their own time), one step's worth of the same rays.
        irays, ihits, isamp = ist["rays"], ist["hits"], ist["samples"]
        shade_bytes = irays * 56.0 + ihits * 132.0 + (irays - isamp) * 48.0 + isamp * 16.0
        shade = dict(bound="hbm", kernel="k_shade" if not lit else "k_shade_lit",
                     achieved=round(shade_bytes / (ist["shade_ms"] * 1e-3) / 1e9, 2) if ist["shade_ms"] > 0 else 0.0,
                     peak=HBM_PEAK_GBS, unit="GB/s", algorithmic_bytes_per_ray=round(shade_bytes / max(1, irays), 1),
                     ms_per_step=round(ist["shade_ms"], 3),
                     note="HIP-event time of the shading launches of ONE untimed step with every chunk on one stream (nothing overlaps them)")
        shade["frac"] = round(shade["achieved"] / HBM_PEAK_GBS, 5)
        # `achieved` prices ALGORITHMIC bytes: part of the 128-byte record gathers (a 33.6 MB table) is served by L2 /
        # Infinity Cache, so it can exceed what HBM itself delivers.  The HBM-side bytes by counter (FETCH_SIZE + WRITE_SIZE
        # of the same kernel, profiles/traffic*.json) over the same time stand beside it.
        if shade_hbm_per_ray and ist["shade_ms"] > 0:
            shade["hbm_counter_bytes_per_ray"] = round(shade_hbm_per_ray, 1)
            shade["hbm_counter_achieved"] = round(shade_hbm_per_ray * irays / (ist["shade_ms"] * 1e-3) / 1e9, 2)
            shade["hbm_counter_frac"] = round(shade["hbm_counter_achieved"] / HBM_PEAK_GBS, 5)
        else:
            shade["hbm_counter_bytes_per_ray"] = shade["hbm_counter_achieved"] = shade["hbm_counter_frac"] = None
        shade["valu_issue"] = valu_shade
        roofline["second_kernel"] = shade
        roofline["profile_stale"] = profile_stale
        roofline["profile_csrc_sha16"] = src_hash

        cpu = None
        parity = None
        cpu_spp = args.cpu_baseline_spp if args.cpu_baseline_spp >= 0 else max(1, min(32, round(4 * 1920 * 1080 / (W * H))))
        if world > 1 and args.cpu_baseline_spp < 0:
            cpu_spp = 0          # the CPU leg belongs to the N = 1 line; with more ranks the others would only wait for it
        if cpu_spp > 0:
            cpu, parity = cpu_baseline(r, wl, assets, W, H, cpu_spp, args.max_depth, args.workload, lit)

        ranks_render = [x[1] for x in per_rank]
        out = {
            "metric": "samples/sec (Mrays/s) Sponza 1080p at 1/2/4/8 MI355X; rel-MSE vs ref",
            "value": round(job_samples / elapsed, 1),
            "unit": "samples/s",
            "n_gpus": world, "rccl_ranks": (dist.get_world_size() if dist is not None else 1),
            "collective_backend": (None if dist is None else ("gloo (rehearsal, host-staged)" if args.rehearse_gloo else "nccl (RCCL)")),
            "steps": len(batches) if strong else args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / max(1, len(batches)), 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload + ("+lit" if lit else ""), "integrator": args.integrator,
                       "resolution": "%dx%d" % (W, H), "spp_per_step": spp,
                       "max_depth": args.max_depth, "seed": SEED, "geometry": wl["geometry"],
                       "parallelism": "spp-partition x%d, scene+BVH replicated, one RCCL reduce" % world if world > 1 else "single GPU",
                       "triangles": int(st["bvh_triangles"]), "bvh_nodes": int(st["bvh_nodes"]),
                       "paths_per_chunk": int(ctx.get_option(capi.OPTION_TARGET_PATHS)),
                       "texel_set_bytes": int(ctx.get_option(capi.OPTION_TEXEL_SET_BYTES))},
            "mrays_per_s": round(job_rays / elapsed / 1e6, 1),
            "rays_per_sample": round(job_rays / max(1.0, job_samples), 3),
            "shadow_rays_per_sample": round(st["shadow_rays"] / max(1, st["samples"]), 3), "lights": int(st["lights"]),
            "scene_load_s": round(load_s, 2),
            # the timed region taken apart (max / min over ranks; per rank in rank order): rendering alone, then the one
            # accumulation reduce -- export to the collective's buffer, RCCL reduce to rank 0, import and 1/N scale
            "time_to_image_ms": round(elapsed * 1e3, 3),
            "total_spp": args.total_spp if strong else None,
            "render_ms": {"max": round(max(ranks_render), 3), "min": round(min(ranks_render), 3), "per_rank": [round(x, 3) for x in ranks_render]},
            "reduce_ms": round(max(x[2] for x in per_rank), 3) if dist is not None else 0.0,
            "scene_load_s_per_rank": [round(x[0], 2) for x in per_rank],
            "spp_per_rank": [int(x[3]) for x in per_rank],
            "tail_launches": int(st["tail_launches"]),       # chunks of the timed region whose last bounces ran as one fused launch (short-path scenes)
            "roofline": roofline,
            "cpu_baseline": cpu,
            "parity": parity,
            "reduce_check": reduce_check,
        }
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    r.close()
    if out is not None:
        print(json.dumps(out), flush=True)


def launch_ranks(n):
    """Starts `python -m torch.distributed.run --nproc-per-node n bench.py <same arguments>` as a child process
    (one rank per GPU, rendezvous on 127.0.0.1 at a free port), relays its output and returns its exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def lit_emitters(workload, materials):
    """(material row, ptx_material_ext) pairs that turn a workload's scene into one with light sources: none of the
    reference's assets has an emitter.  Sponza: the 8 small primitives of glTF material 3 (one material row per
    primitive, material.rs:34-90) glow warm; other scenes keep the sky as their only light."""
    from path_tracer_amd import capi
    out = []
    if workload.startswith("sponza"):
        doc = json.load(open(os.path.join(ROOT, "assets", "sponza", "Sponza.gltf")))
        for row, prim in enumerate(doc["meshes"][0]["primitives"]):
            if prim.get("material") == 3 and row < len(materials):
                e = np.zeros(1, dtype=capi.MATERIAL_EXT_DTYPE)
                e["emissive"] = [24.0, 18.0, 10.0]
                e["ior"] = 1.5
                out.append((row, e))
    return out


def cpu_baseline(r, wl, assets, W, H, spp, max_depth, workload, lit):
    """Times the CPU oracle (oracle/pt_oracle.c, a port of the reference's shaders -- the reference itself
    has no CPU path) on a bounded sample of the same workload, and checks the GPU image against it."""
    from oracle import pyoracle as O
    from oracle.gltf_ingest import IngestedScene
    from path_tracer_amd import capi, host
    ing = IngestedScene(ref_quirks=not lit)
    for rel, off in wl["items"]:
        ing.import_gltf(os.path.join(assets, rel), off, decode_image=host.decode_image)   # same texels as the GPU side
    if lit:
        for row, ext in lit_emitters(workload, ing.materials):
            ing.material_ext[row] = ext[0]
        sc = O.scene_from_ingest(ing, options=O.SCENE_WORLD_NORMALS | O.SCENE_TEXTURE_TIMES_FACTOR, with_material_ext=True)
    else:
        sc = O.scene_from_ingest(ing)
    integ = O.INTEGRATOR_LIT if lit else O.INTEGRATOR_REFERENCE
    ginteg = capi.INTEGRATOR_LIT if lit else capi.INTEGRATOR_REFERENCE
    cam = O.camera_matrices(wl["pos"], O.yaw_quat(wl["yaw"]), np.deg2rad(np.float32(wl["fov"])), W / H, 0.1, 1000.0)
    threads = min(os.cpu_count() or 1, 16)
    t = time.perf_counter()
    want = sc.render(cam, W, H, spp, SEED, params=O.default_params(max_depth=max_depth, integrator=integ), threads=threads)
    dt = time.perf_counter() - t
    # the same spp frames on the GPU, from a fresh accumulation, through the C ABI
    ctx = r._ctx()
    ctx.resize(W, H)
    gcam = host.camera_matrices(wl["pos"], host.yaw_quat(wl["yaw"]), np.deg2rad(np.float32(wl["fov"])), W / H, 0.1, 1000.0)
    u = capi.Uniforms(1, SEED, 0, 0.1, 100.0, (0, 0, 0))
    ctx.render_batch(u, gcam, spp, capi.default_params(max_depth=max_depth, integrator=ginteg))
    got = ctx.read_image(capi.IMAGE_ALBEDO)
    ids = ctx.read_image(capi.IMAGE_PRIMARY_ID)
    a = np.asarray(got, np.float64)[..., :3]
    b = np.asarray(want["albedo"], np.float64)[..., :3]
    rel_mse = float(np.mean((a - b) ** 2 / (b * b + 1e-2)))
    parity = {"checked_against": "oracle/pt_oracle.c, the CPU restatement of the reference's shaders (a port: parity with the Vulkan reference itself is unpinned -- it has no fixtures and cannot run here)",
              "spp": spp, "primary_ids_bit_exact": bool(np.array_equal(ids, want["ids"])),
              "albedo_bit_exact": bool(np.array_equal(got.view(np.uint32), want["albedo"].view(np.uint32))),
              "rel_mse": rel_mse, "rel_mse_eps": 1e-2}
    cpu = {"value": round(want["stats"]["samples"] / dt, 1), "unit": "samples/s", "cores": threads, "kind": "port",
           "mrays_per_s": round(want["stats"]["rays"] / dt / 1e6, 3),
           "sample": "%d spp of the full %dx%d frame (%d pixel-samples, %.1f s)" % (spp, W, H, want["stats"]["samples"], dt)}
    return cpu, parity


if __name__ == "__main__":
    main()
