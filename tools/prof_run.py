"""Small fixed workload for rocprofv3 passes: Sponza stand-in 1920x1080, N spp through the host mirror."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from path_tracer_amd import capi, host, standins
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lit = len(sys.argv) > 2 and sys.argv[2] == "lit"        # bench.py --integrator lit: emitters, real MR / normal maps, no-ref-quirks options
assets = os.path.join(ROOT, "assets")
standins.ensure_standins(assets)
r = host.Renderer(1920, 1080, 0, capi.default_limits(max_vertices=10**7, max_indices=3 * 10**7))
if lit:
    r.set_ref_quirks(False)
r.import_gltf(os.path.join(assets, "sponza/Sponza.gltf"))
if lit:
    import bench
    for row, ext in bench.lit_emitters("sponza_1080p", r.scene_arrays()[2]):
        r._ctx().set_material_ext(row, ext)
    r.set_params(capi.default_params(integrator=capi.INTEGRATOR_LIT))
r.set_camera([-8.0, 1.6, 0.0], host.yaw_quat(-90.0), np.deg2rad(np.float32(70.0)), 0.1, 1000.0)
r.set_seed(0x1234567, True)
r.record_commands_batch(spp)
r._ctx().synchronize()
st = r.stats()
print("rays %d samples %d" % (st["rays"], st["samples"]))
r.close()
