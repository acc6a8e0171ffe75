"""Regenerates the golden fixtures from the CPU oracle (run from the repo root:
python tests/golden/make_golden.py).  The reference ships no tests, fixtures or golden images and
cannot be run here, so these vectors pin the ORACLE (and, through the GPU parity
tests, the HIP path) against drift -- compiler contraction, libm changes, refactors -- not against
the reference's own output."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import SEED, cornell, lion_boulder  # noqa: E402
from oracle import pyoracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
import ctypes as C  # noqa: E402

L = O.lib()
f3 = lambda a: np.asarray(a, np.float32).ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731


def bits(x):
    return int(np.float32(x).view(np.uint32))


kat = {}
ins = [0, 1, 2, 0xFFFFFFFF, 0x80000000, 0x12345678, 0xDEADBEEF, 19088743, 7919, 16381, 0x3F800000, 0x3F000000]
kat["hash"] = [[i, L.orc_hash(i)] for i in ins]
kat["hash2"] = [[a, b, L.orc_hash2(a, b)] for a in ins[:6] for b in ins[4:9]]
rand_in = [(0.5, 0.5, 0, 0, 0), (0.000244140625, 0.99951171875, 3, 17, SEED), (0.25, 0.75, 15, 4095, 0xFFFFFFFF),
           (0.1337, 0.4242, 7, 123456, 42), (0.9990234375, 0.0009765625, 1, 1, 1)]
kat["rand"] = [[bits(u), bits(v), d, a, s, bits(L.orc_rand(np.float32(u), np.float32(v), d, a, s))] for u, v, d, a, s in rand_in]
normals = [(0, 0, 1), (0, 0, -1), (0, 1, 0), (1, 0, 0), (0, -1, 0), (-1, 0, 0), (0.6, 0.8, 0.0), (0.0, 0.6, -0.8),
           (0.57735026, 0.57735026, 0.57735026), (-0.57735026, 0.57735026, -0.57735026), (0.26726124, 0.5345225, 0.8017837),
           (0.0, 1e-8, 1.0), (0.7071068, 0.0, -0.7071068), (0.1, -0.2, 0.9746794), (-0.3, 0.9, 0.31622776), (0.99, 0.1, 0.1)]
kat["create_basis"] = []
kat["cosine_hemisphere_sample"] = []
for k, n in enumerate(normals):
    t = np.zeros(3, np.float32); b = np.zeros(3, np.float32)
    L.orc_create_basis(f3(n), f3(t), f3(b))
    kat["create_basis"].append([[bits(x) for x in np.float32(n)], [bits(x) for x in t], [bits(x) for x in b]])
    out = np.zeros(3, np.float32)
    u, v, d, a, s = rand_in[k % len(rand_in)]
    L.orc_cosine_hemisphere_sample(np.float32(u), np.float32(v), d + k, a, s, f3(n), f3(out))
    kat["cosine_hemisphere_sample"].append([bits(u), bits(v), d + k, a, s, [bits(x) for x in np.float32(n)], [bits(x) for x in out]])
kat["sincos"] = []
for phi in np.linspace(0, 6.2831855, 41, dtype=np.float32):
    s = C.c_float(); c = C.c_float()
    L.orc_sincos(np.float32(phi), C.byref(s), C.byref(c))
    kat["sincos"].append([bits(phi), bits(s.value), bits(c.value)])
kat["sky_color"] = []
for d in [(0, 1, 0), (0, -1, 0), (0.3, 0.5, 0.2), (0, 0, 1), (1, 2.5, 0), (0, 1e-6, 1)]:
    out = np.zeros(3, np.float32)
    L.orc_sky_color(f3(d), f3(out))
    kat["sky_color"].append([[bits(x) for x in np.float32(d)], [bits(x) for x in out]])
kat["srgb_lut"] = [bits(L.orc_srgb_to_linear(i)) for i in range(256)]
cam = O.camera_matrices([0, 1.0, 3.9], O.yaw_quat(0), np.deg2rad(np.float32(40)), 1.0, 0.1, 1000.0)
cam2 = O.camera_matrices([-8, 1.6, 0], O.yaw_quat(-90), np.deg2rad(np.float32(70)), 16 / 9, 0.1, 1000.0)
kat["camera_cornell"] = [int(x) for x in np.frombuffer(bytes(cam), np.uint32)]
kat["camera_sponza"] = [int(x) for x in np.frombuffer(bytes(cam2), np.uint32)]
kat["transform_cornell_node"] = [int(x) for x in O.transform_matrix([0, 0.00454026460647583, 0.004760861396789551],
                                                                     [-0.7072679400444031, 0, 0, 0.7069455981254578], [1, 1, 1]).view(np.uint32).ravel()]
json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=0)

# Cornell 64x64, 16 spp, max depth 8 (config C1 at reduced resolution), brute-force closest hit
sc = O.scene_from_ingest(cornell())
r = sc.render(cam, 64, 64, 16, SEED, params=O.default_params(max_depth=8, flags=1))
np.savez_compressed(os.path.join(HERE, "cornell_64x64_16spp_depth8.npz"), albedo=r["albedo"], normal=r["normal"],
                    depth=r["depth"], ids=r["ids"], stats=np.array([r["stats"][k] for k in ("rays", "hits", "samples")], np.uint64))
# Cornell 512x512, 64 spp, depth 8 (config C1 proper): summary only
r = sc.render(cam, 512, 512, 64, SEED, params=O.default_params(max_depth=8))
a = r["albedo"][..., :3].astype(np.float64)
summary = {"cornell_512_64spp_depth8": {"mean": a.mean(axis=(0, 1)).tolist(), "var": a.var(axis=(0, 1)).tolist(),
                                        "rays": int(r["stats"]["rays"]), "hits": int(r["stats"]["hits"]),
                                        "ids_crc": int(np.bitwise_xor.reduce(r["ids"].astype(np.uint64).ravel() * np.arange(1, r["ids"].size + 1, dtype=np.uint64)))}}
# lion + boulder close-up 160x90, 2 spp (textures decoded by PIL: texel-level input, not part of the fixture)
sc2 = O.scene_from_ingest(lion_boulder())
cam3 = O.camera_matrices([2.0, 0.25, 0.6], O.yaw_quat(0), np.deg2rad(np.float32(70)), 160 / 90, 0.1, 1000.0)
r2 = sc2.render(cam3, 160, 90, 2, SEED)
np.savez_compressed(os.path.join(HERE, "lion_boulder_160x90_ids.npz"), ids=r2["ids"], depth=r2["depth"],
                    stats=np.array([r2["stats"][k] for k in ("rays", "hits", "samples")], np.uint64))
json.dump(summary, open(os.path.join(HERE, "summary.json"), "w"), indent=1)
print("golden fixtures written to", HERE)
