"""Where the waves of the traversal kernel spend their lifetime: runs the bench workload once with the instrumented
kernel (profiling level 2) and prints the per-phase shader-cycle totals of ptx_stats.extend_phase_cycles.
   python tools/phase_profile.py [spp] [workload]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from path_tracer_amd import capi, host, standins
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
wl = bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "sponza_1080p"]
assets = os.path.join(ROOT, "assets")
standins.ensure_standins(assets)
W, H = wl["res"]
r = host.Renderer(W, H, 0, capi.default_limits(max_vertices=10**7, max_indices=3 * 10**7))
for rel, off in wl["items"]:
    r.import_gltf(os.path.join(assets, rel), off)
r.set_camera(wl["pos"], host.yaw_quat(wl["yaw"]), np.deg2rad(np.float32(wl["fov"])), 0.1, 1000.0)
r.set_seed(bench.SEED, True)
r.record_commands_batch(spp)
ctx = r._ctx()
ctx.synchronize()
ctx.set_profiling(2)
ctx.reset_stats()
r.record_commands_batch(spp)
st = ctx.stats()
names = ["ray fetch", "node-load wait", "node test + pick", "queue append", "triangle fetch wait", "triangle test", "hit hand-back", "retire + loop"]
ph = st["extend_phase_cycles"]
tot = float(sum(ph)) or 1.0
ws = st["wave_sched"]
out = {"rays": st["rays"], "extend_ms": st["extend_ms"], "iterations": ws[0], "node_blocks": ws[1], "tri_blocks": ws[3],
       "phases": {n: {"cycles": int(c), "share": round(c / tot, 4), "cycles_per_iteration": round(c / max(1, ws[0]), 1)} for n, c in zip(names, ph)}}
print(json.dumps(out, indent=1))
r.close()
