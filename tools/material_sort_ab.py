"""A/B of the material-sorting option of the lit integrator (PTX_MATERIAL_SORT): samples/s with
the option off and on, each in a fresh process (the option is read at ptx_create), on
  sponza_1080p+lit   bench.py --integrator lit (all materials opaque: one class besides the sky)
  cornell_lit 1024^2 the lit Cornell box of the tests (emitter, glass box, rough-metal box), depth 16, 16-spp batches
usage: tools/material_sort_ab.py            (parent: runs the four children, prints one JSON)
       tools/material_sort_ab.py cornell    (child)"""
import json, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cornell_child():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from path_tracer_amd import capi, host
    r = host.Renderer(1024, 1024, 0)
    r.set_ref_quirks(False)
    r.import_gltf(os.path.join(ROOT, "assets", "cornell_lit", "cornell_lit.gltf"))
    r.set_camera([0.0, 1.0, 3.9], host.yaw_quat(0.0), np.deg2rad(np.float32(40.0)), 0.1, 1000.0)
    r.set_seed(0x1234567, fixed=True)
    r.set_params(capi.default_params(max_depth=16, integrator=capi.INTEGRATOR_LIT))
    ctx = r._ctx()
    r.record_commands_batch(32)
    ctx.synchronize(); ctx.set_profiling(1); ctx.reset_stats()
    t = time.perf_counter()
    for _ in range(8):
        r.record_commands_batch(32)
    ctx.synchronize()
    dt = time.perf_counter() - t
    st = ctx.stats()
    print(json.dumps({"samples_per_s": st["samples"] / dt, "rays_per_sample": st["rays"] / st["samples"],
                      "shadow_rays_per_sample": st["shadow_rays"] / st["samples"], "shade_ms": st["shade_ms"], "extend_ms": st["extend_ms"]}))
    r.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cornell":
        cornell_child()
        sys.exit(0)
    out = {}
    for sort in ("0", "1"):
        env = dict(os.environ, PTX_MATERIAL_SORT=sort)
        b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--integrator", "lit", "--steps", "8", "--warmup", "2", "--cpu-baseline-spp", "0"],
                           env=env, capture_output=True, text=True, timeout=600)
        d = json.loads(b.stdout.strip().splitlines()[-1])
        c = subprocess.run([sys.executable, os.path.abspath(__file__), "cornell"], env=env, capture_output=True, text=True, timeout=600)
        out["sort_" + sort] = {"sponza_1080p_lit": {"samples_per_s": d["value"], "mrays_per_s": d["mrays_per_s"], "ms_per_step": d["ms_per_step"],
                                                     "shade_ms_per_step_serial": d["roofline"]["second_kernel"]["ms_per_step"]},
                               "cornell_lit_1024": json.loads(c.stdout.strip().splitlines()[-1])}
    print(json.dumps(out, indent=1))
