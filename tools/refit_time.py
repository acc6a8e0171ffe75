#!/usr/bin/env python
"""Times ptx_update_instances' refit against a full rebuild on the Sponza stand-in."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from path_tracer_amd import capi, host, standins
standins.ensure_standins(os.path.join(ROOT, "assets"))
r = host.Renderer(640, 360, 0, capi.default_limits(max_vertices=10 ** 7, max_indices=3 * 10 ** 7))
r.import_gltf(os.path.join(ROOT, "assets", "sponza/Sponza.gltf"))
r.set_camera([-8.0, 1.6, 0.0], host.yaw_quat(-90.0), np.deg2rad(np.float32(70.0)), 0.1, 1000.0)
r.set_seed(1, True)
t = time.perf_counter(); r.record_commands(); r._ctx().synchronize(); build_s = time.perf_counter() - t
t = time.perf_counter(); r.record_commands(); r._ctx().synchronize(); frame_s = time.perf_counter() - t
out = {"first_frame_with_build_s": round(build_s, 4), "plain_frame_s": round(frame_s, 4), "refit_frames_s": []}
for k in range(3):
    r.set_entity_transform(5 + k, [0.01 * (k + 1), 0.0, 0.0], (0, 0, 0, 1), (0.008, 0.008, 0.008))
    t = time.perf_counter(); r.record_commands(); r._ctx().synchronize()
    out["refit_frames_s"].append(round(time.perf_counter() - t, 4))
st = r.stats()
out["bvh_builds"], out["bvh_refits"] = st["bvh_builds"], st["bvh_refits"]
print(json.dumps(out))
