"""Converged-image fixture (BASELINE.json: "rel-MSE <= 1e-3 vs the reference at 4096 spp"; config C5 at reduced
resolution): the CPU oracle renders the Sponza stand-in at 192x108 with 4096 samples per pixel, max depth 16 --
about ten minutes on 8 cores -- and the image is committed as float16-exact-enough float32 (npz).  The GPU test
renders the same 4096 samples and states the relative MSE.  python tests/golden/make_golden_converged.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import ASSETS, SEED  # noqa: E402
from oracle import pyoracle as O  # noqa: E402
from oracle.gltf_ingest import IngestedScene  # noqa: E402
from path_tracer_amd import host, standins  # noqa: E402

W, H, SPP = 192, 108, 4096
standins.ensure_standins(ASSETS)
ing = IngestedScene()
ing.import_gltf(os.path.join(ASSETS, "sponza", "Sponza.gltf"), decode_image=host.decode_image)   # the product's own codecs: same texels
sc = O.scene_from_ingest(ing)
cam = O.camera_matrices([-8.0, 1.6, 0.0], O.yaw_quat(-90.0), np.deg2rad(np.float32(70.0)), W / H, 0.1, 1000.0)
t = time.time()
r = sc.render(cam, W, H, SPP, SEED, threads=int(os.environ.get("THREADS", "8")))
print("rendered in %.0f s" % (time.time() - t), r["stats"])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sponza_standin_192x108_4096spp.npz"),
                    albedo=r["albedo"][..., :3].astype(np.float32), spp=np.uint32(SPP),
                    stats=np.array([r["stats"][k] for k in ("rays", "hits", "samples")], np.uint64))
