#!/usr/bin/env python
"""Probe: do 1-spp frames of two contexts on one GPU overlap (tail of one frame's bounces filled by the
other's)?  Aggregate frame rate of 1 vs 2 vs 3 contexts issuing frames round-robin."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import bench
from path_tracer_amd import capi, host, standins
name = sys.argv[1] if len(sys.argv) > 1 else "sponza_1080p"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wl = bench.WORKLOADS[name]
standins.ensure_standins(os.path.join(ROOT, "assets"))
W, H = wl["res"]
def make():
    r = host.Renderer(W, H, 0, capi.default_limits(max_vertices=10 ** 7, max_indices=3 * 10 ** 7))
    for rel, off in wl["items"]:
        r.import_gltf(os.path.join(ROOT, "assets", rel), off)
    r.set_camera(wl["pos"], host.yaw_quat(wl["yaw"]), np.deg2rad(np.float32(wl["fov"])), 0.1, 1000.0)
    r.set_seed(bench.SEED, fixed=True)
    for _ in range(3):
        r.record_commands()
    r._ctx().synchronize()
    return r
rs = [make() for _ in range(3)]
out = {}
for n in (1, 2, 3):
    t = time.perf_counter()
    for _ in range(frames):
        for r in rs[:n]:
            r.record_commands()
    for r in rs[:n]:
        r._ctx().synchronize()
    dt = time.perf_counter() - t
    out[n] = round(dt * 1e3 / (frames * n), 3)
print(json.dumps({"workload": name, "ms_per_frame_by_contexts": out}))
