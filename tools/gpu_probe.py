"""Dev probe (not shipped): time the HIP path on one scene and optionally check it against the oracle."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import *  # noqa
from oracle import pyoracle as O
from path_tracer_amd import capi, standins

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="sponza")
ap.add_argument("--res", default="1920x1080")
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--check", type=int, default=0, help="oracle spp to compare bit-exactly")
ap.add_argument("--depth", type=int, default=16)
a = ap.parse_args()
W, H = map(int, a.res.split("x"))
standins.ensure_standins(ASSETS)
if a.scene == "sponza":
    ing = ingest(("sponza/Sponza.gltf", (0, 0, 0))); cam = ([-8, 1.6, 0], -90, 70)
elif a.scene == "spheres":
    ing = ingest(("spheres/spheres.gltf", (0, 0, 0))); cam = ([7.0, 2.0, 0.0], 90, 70)
elif a.scene == "lion":
    ing = lion_boulder(); cam = ([2.0, 0.25, 0.6], 0, 70)
else:
    ing = cornell(); cam = ([0, 1.0, 3.9], 0, 40)
lim = capi.default_limits(max_vertices=10**7, max_indices=3 * 10**7)
ctx = capi.Context(0, lim)
t = time.time(); upload(ctx, ing, O); print("upload+build %.2fs" % (time.time() - t))
ocam = O.camera_matrices(cam[0], O.yaw_quat(cam[1]), np.deg2rad(np.float32(cam[2])), W / H, 0.1, 1000.0)
gcam = copy_struct(capi.Camera, ocam)
ctx.resize(W, H)
params = capi.default_params(max_depth=a.depth)
u = capi.Uniforms(1, SEED, 0, 0.1, 100.0, (0, 0, 0))
ctx.render_batch(u, gcam, a.spp, params); ctx.synchronize()
print(ctx.stats())
for prof in (0, 1):
    ctx.set_profiling(prof); ctx.reset_stats()
    t = time.time()
    for r in range(a.reps):
        u.accumulated_count = a.spp * (r + 1)
        ctx.render_batch(u, gcam, a.spp, params)
    ctx.synchronize(); dt = time.time() - t
    st = ctx.stats()
    print("profiling=%d: %.3fs  %.1f Msamples/s  %.1f Mrays/s  rays/sample %.2f  extend %.1f ms (%d launches) shade %.1f ms total %.1f ms" % (
        prof, dt, st["samples"] / dt / 1e6, st["rays"] / dt / 1e6, st["rays"] / st["samples"], st["extend_ms"], st["extend_launches"], st["shade_ms"], st["total_ms"]))
ctx.set_profiling(2); ctx.reset_stats()
ctx.render_batch(u, gcam, a.spp, params); st = ctx.stats(); ctx.set_profiling(0)
ws = st["wave_sched"]
print("nodes/ray %.2f tris/ray %.2f | wave iters %d node blocks %d (avg lanes %.1f) tri blocks %d (avg lanes %.1f) refills %d | rays %d" % (
    st["nodes_fetched"] / st["rays"], st["tris_tested"] / st["rays"], ws[0], ws[1], ws[2] / max(1, ws[1]), ws[3], ws[4] / max(1, ws[3]), ws[5], st["rays"]))
if a.check:
    sc = O.scene_from_ingest(ing)
    t = time.time(); want = sc.render(ocam, W, H, a.check, SEED, params=O.default_params(max_depth=a.depth)); print("oracle %.1fs" % (time.time() - t), want["stats"])
    got = gpu_render(ctx, O, ocam, W, H, a.check, params=params)
    assert_images_bit_equal(got, want, "probe")
    print("bit-exact vs oracle OK")
