import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from path_tracer_amd import capi, host
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
wl = bench.WORKLOADS["cornell_512"]
for res in ((512, 512), (256, 256), (1024, 1024)):
    r = host.Renderer(res[0], res[1], 0)
    for rel, off in wl["items"]:
        r.import_gltf(os.path.join(ROOT, "assets", rel), off)
    r.set_camera(wl["pos"], host.yaw_quat(wl["yaw"]), np.deg2rad(np.float32(wl["fov"])), 0.1, 1000.0)
    r.set_seed(1, True)
    for _ in range(8): r.record_commands()
    r._ctx().synchronize()
    n = 256
    t0 = time.perf_counter()
    for _ in range(n): r.record_commands()
    t1 = time.perf_counter()
    r._ctx().synchronize()
    t2 = time.perf_counter()
    print(res, "issue ms/frame %.3f  total ms/frame %.3f" % ((t1 - t0) * 1e3 / n, (t2 - t0) * 1e3 / n))
    r.close()
