import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from path_tracer_amd import capi, host
from oracle import pyoracle as O
from oracle.gltf_ingest import IngestedScene
gltf = os.path.join(ROOT, "assets", "cornell", "cornell.gltf")
fov = np.deg2rad(np.float32(40.0))
ing = IngestedScene(); ing.import_gltf(gltf); sc = O.scene_from_ingest(ing)
cam = O.camera_matrices([0.0, 1.0, 3.9], O.yaw_quat(0.0), fov, 1.0, 0.1, 1000.0)
hc = host.camera_matrices([0.0, 1.0, 3.9], host.yaw_quat(0.0), fov, 1.0, 0.1, 1000.0)
print("cam equal", bytes(cam) == bytes(hc))
for seq in (["f"], ["f", "f"], ["f", "b4"], ["b5"]):
    r = host.Renderer(64, 64, 0); r.import_gltf(gltf)
    r.set_camera([0.0, 1.0, 3.9], host.yaw_quat(0.0), fov, 0.1, 1000.0); r.set_seed(0x1234567, True)
    r.set_params(capi.default_params(max_depth=8))
    for s in seq:
        if s == "f": r.record_commands()
        else: r.record_commands_batch(int(s[1:]))
    n = r.accumulated_count
    got = r.read_image(capi.IMAGE_ALBEDO)
    want = sc.render(cam, 64, 64, max(n, 1), 0x1234567, params=O.default_params(max_depth=8, flags=1))["albedo"]
    d = got.view(np.uint32) != want.view(np.uint32)
    print(seq, "acc", n, "diff px", d.any(-1).sum(), "maxabs", np.abs(got - want).max())
    r.close()
