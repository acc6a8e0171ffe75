#!/usr/bin/env python
"""Convergence of the two integrators on the lit Cornell box (the lit integrator is a non-parity
mode with its own rel-MSE report).  For each integrator: a converged image (REF_SPP samples, one seed), then
rel-MSE of N-sample images rendered with ANOTHER seed against it, for N = 1 .. 4096 -- an unbiased estimator
falls as 1/N down to the floor the converged image's own noise sets.  Also the distance between the two
converged images (different light transport models: the reference's integrator sees no emitters, no glass).
rel-MSE = mean((a-b)^2 / (b^2 + 1e-2)) over RGB (BASELINE.json's definition).
usage: tools/convergence_report.py [res] [ref_spp]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from path_tracer_amd import capi, host  # noqa: E402

RES = int(sys.argv[1]) if len(sys.argv) > 1 else 512
REF_SPP = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
SEED_REF, SEED_TEST = 0x1234567, 0x7654321


def rel_mse(a, b, eps=1e-2):
    a, b = a[..., :3].astype(np.float64), b[..., :3].astype(np.float64)
    return float(np.mean((a - b) ** 2 / (b ** 2 + eps)))


r = host.Renderer(RES, RES, 0)
r.set_ref_quirks(False)
r.import_gltf(os.path.join(ROOT, "assets", "cornell_lit", "cornell_lit.gltf"))
r.set_camera([0.0, 1.0, 3.9], host.yaw_quat(0.0), np.deg2rad(np.float32(40.0)), 0.1, 1000.0)
ctx = r._ctx()


def render(integrator, seed, spp):
    r.set_params(capi.default_params(max_depth=16, integrator=integrator))
    r.set_seed(seed, fixed=True)
    r.resize_render_buffer(RES, RES)        # restart the accumulation
    r.record_commands(); r.record_commands()
    ctx.synchronize()
    ctx.reset_stats()
    t = time.perf_counter()
    done = 0
    while done < spp:                       # chunks keep the watchdog fed on long runs
        k = min(4096, spp - done)
        r.record_commands_batch(k)
        ctx.synchronize()
        done += k
    dt = time.perf_counter() - t
    return r.read_image(capi.IMAGE_ALBEDO).copy(), dt, ctx.stats()


out = {"scene": "assets/cornell_lit/cornell_lit.gltf", "res": [RES, RES], "ref_spp": REF_SPP, "max_depth": 16,
       "rel_mse_eps": 1e-2, "integrators": {}}
converged = {}
for name, integ in (("lit", capi.INTEGRATOR_LIT), ("reference", capi.INTEGRATOR_REFERENCE)):
    ref, dt, st = render(integ, SEED_REF, REF_SPP)
    converged[name] = ref
    rows = []
    for n in (1, 4, 16, 64, 256, 1024, 4096):
        img, dtn, stn = render(integ, SEED_TEST, n)
        rows.append({"spp": n, "rel_mse": rel_mse(img, ref), "rel_mse_x_spp": rel_mse(img, ref) * n})
    out["integrators"][name] = {
        "converged_render_s": round(dt, 2), "msamples_per_s": round(st["samples"] / dt / 1e6, 1),
        "rays_per_sample": round(st["rays"] / max(1, st["samples"]), 3),
        "shadow_rays_per_sample": round(st.get("shadow_rays", 0) / max(1, st["samples"]), 3),
        "mean_radiance": [round(float(x), 5) for x in ref[..., :3].mean(axis=(0, 1))],
        "finite": bool(np.isfinite(ref).all()), "convergence": rows}
out["lit_vs_reference_converged_rel_mse"] = rel_mse(converged["lit"], converged["reference"])
print(json.dumps(out, indent=1))
r.close()
