"""The N > 1 path on CPU: two gloo ranks each render their slice of the sample budget (the oracle
stands in for the GPU renderer -- same uniforms / rng_sample_offset contract as ptx_render_batch),
the slices are combined by path_tracer_amd.multi_gpu.combine_accumulation, and the result must match
a single-process render of the whole budget."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT, SEED, cornell, rel_mse

TOTAL_SPP, W, H = 8, 48, 32


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as O
    from path_tracer_amd.multi_gpu import combine_accumulation, partition_samples
    first, n = partition_samples(TOTAL_SPP, world, rank)
    sc = O.scene_from_ingest(cornell())
    cam = O.camera_matrices([0, 1.0, 3.9], O.yaw_quat(0), np.deg2rad(np.float32(40)), W / H, 0.1, 1000.0)
    # local running mean over this rank's samples: accumulated_count 0..n-1, RNG indices first..first+n-1
    r = sc.render(cam, W, H, n, SEED, first_sample=0, params=O.default_params(rng_sample_offset=first), threads=2)
    t = torch.from_numpy(r["albedo"].copy())
    combined = combine_accumulation(t, dist, world, root=0)
    if rank == 0:
        np.save(out_path, combined.numpy())
    else:
        assert combined is None
    dist.barrier()
    dist.destroy_process_group()


def test_partition_samples():
    from path_tracer_amd.multi_gpu import partition_samples
    assert [partition_samples(1024, 8, r) for r in (0, 1, 7)] == [(0, 128), (128, 128), (896, 128)]
    assert partition_samples(5, 1, 0) == (0, 5)
    with pytest.raises(ValueError):
        partition_samples(10, 4, 0)


def test_two_rank_spp_partition_matches_single_process(oracle, tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "combined.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    combined = np.load(out)
    sc = oracle.scene_from_ingest(cornell())
    cam = oracle.camera_matrices([0, 1.0, 3.9], oracle.yaw_quat(0), np.deg2rad(np.float32(40)), W / H, 0.1, 1000.0)
    whole = sc.render(cam, W, H, TOTAL_SPP, SEED)["albedo"]
    assert np.all(combined[..., 3] == 1.0)
    # same samples, different fp32 summation order (running mean vs mean of partial means)
    assert np.abs(combined - whole).max() < 2e-6
    assert rel_mse(combined, whole) < 1e-10


def _preview_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as O
    from path_tracer_amd.multi_gpu import ProgressivePreview, combine_accumulation, partition_samples, render_progressive
    first, n = partition_samples(TOTAL_SPP, world, rank)
    sc = O.scene_from_ingest(cornell())
    cam = O.camera_matrices([0, 1.0, 3.9], O.yaw_quat(0), np.deg2rad(np.float32(40)), W / H, 0.1, 1000.0)
    state = {"acc": None, "n": 0}

    def render(k):       # the oracle stands in for ptx_render_batch: k more samples onto the running mean
        r = sc.render(cam, W, H, k, SEED, first_sample=state["n"], params=O.default_params(rng_sample_offset=first),
                      threads=2, albedo=state["acc"])
        state["acc"], state["n"] = r["albedo"], state["n"] + k

    previews = []
    pv = ProgressivePreview(dist, world, root=0, on_preview=lambda img, samples: previews.append((samples, img.numpy().copy())))
    render_progressive(render, lambda: torch.from_numpy(state["acc"].copy()), n, 1, pv)
    assert state["n"] == n and pv.submitted == n - 1
    combined = combine_accumulation(torch.from_numpy(state["acc"].copy()), dist, world, root=0)
    if rank == 0:
        assert [s for s, _ in previews] == [world * m for m in range(1, n)], [s for s, _ in previews]
        np.savez(out_path, final=combined.numpy(), **{"p%d" % s: img for s, img in previews})
    else:
        assert previews == [] and pv.latest is None
    dist.barrier()
    dist.destroy_process_group()


def test_progressive_preview_two_ranks(tmp_path, oracle):
    """Previews between chunks equal the mean over the samples both ranks hold at that point, and taking
    them leaves the final combined image unchanged."""
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "preview.npz")
    mp.spawn(_preview_worker, args=(world, port, out), nprocs=world, join=True)
    got = np.load(out)
    sc = oracle.scene_from_ingest(cornell())
    cam = oracle.camera_matrices([0, 1.0, 3.9], oracle.yaw_quat(0), np.deg2rad(np.float32(40)), W / H, 0.1, 1000.0)
    n = TOTAL_SPP // world
    for m in range(1, n):
        parts = [sc.render(cam, W, H, m, SEED, params=oracle.default_params(rng_sample_offset=r * n), threads=2)["albedo"] for r in range(world)]
        want = (parts[0] + parts[1]) * np.float32(0.5)
        want[..., 3] = 1.0
        np.testing.assert_array_equal(got["p%d" % (world * m)], want)
    whole = sc.render(cam, W, H, TOTAL_SPP, SEED, threads=2)["albedo"]
    assert np.abs(got["final"] - whole).max() < 2e-6


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` as the driver types it (no WORLD_SIZE in the environment): the parent must start
    torch.distributed.run as a child process, relay its JSON line and its exit code.  --launch-check with
    --rehearse-gloo brings the ranks up over gloo and touches no GPU, so the launcher path runs here."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check", "--rehearse-gloo"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d == {"launch_check": True, "n_gpus": 2, "rccl_ranks": 2, "backend": "gloo"}
    # a rank count that does not match --gpus is an error, not a silent single-GPU run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check", "--rehearse-gloo"],
                       env=env2, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
