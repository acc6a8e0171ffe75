#!/usr/bin/env python
"""Interactive-mode timing: the reference's usage pattern is one ptx_render per frame
(1 sample per pixel, renderer.rs:1002-1067).  Prints ms per frame for a run of clean frames and the
per-stage GPU time from HIP events.  usage: tools/frame_time.py [workload] [frames]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
import bench  # noqa: E402
from path_tracer_amd import capi, host, standins  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "sponza_1080p"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wl = bench.WORKLOADS[name]
standins.ensure_standins(os.path.join(ROOT, "assets"))
W, H = wl["res"]
r = host.Renderer(W, H, 0, capi.default_limits(max_vertices=10 ** 7, max_indices=3 * 10 ** 7))
for rel, off in wl["items"]:
    r.import_gltf(os.path.join(ROOT, "assets", rel), off)
r.set_camera(wl["pos"], host.yaw_quat(wl["yaw"]), np.deg2rad(np.float32(wl["fov"])), 0.1, 1000.0)
r.set_seed(bench.SEED, fixed=True)
ctx = r._ctx()
for _ in range(4):
    r.record_commands()
ctx.synchronize()
out = {"workload": name, "frames": frames}
for prof in (0, 1):
    ctx.set_profiling(prof)
    ctx.reset_stats()
    t = time.perf_counter()
    for _ in range(frames):
        r.record_commands()
    ctx.synchronize()
    dt = time.perf_counter() - t
    st = ctx.stats()
    key = "profiled" if prof else "plain"
    out[key] = {"ms_per_frame": round(dt * 1e3 / frames, 3), "msamples_per_s": round(st["samples"] / dt / 1e6, 1),
                "mrays_per_s": round(st["rays"] / dt / 1e6, 1)}
    if prof:
        out[key].update(extend_ms_per_frame=round(st["extend_ms"] / frames, 3), shade_ms_per_frame=round(st["shade_ms"] / frames, 3),
                        gpu_ms_per_frame=round(st["total_ms"] / frames, 3))
# a moving camera: every frame restarts the accumulation (renderer.rs:988-1000) and rewrites the G-buffers
ctx.set_profiling(0)
ctx.reset_stats()
t = time.perf_counter()
for k in range(frames):
    r.set_camera(wl["pos"], host.yaw_quat(wl["yaw"] + 0.05 * (k + 1)), np.deg2rad(np.float32(wl["fov"])), 0.1, 1000.0)
    r.record_commands()
ctx.synchronize()
dt = time.perf_counter() - t
out["moving_camera"] = {"ms_per_frame": round(dt * 1e3 / frames, 3)}
print(json.dumps(out))
