#!/usr/bin/env python
"""Developer experiment: traversal statistics of the product's BVH8 on the Sponza stand-in for bounce-like rays
(origins on surfaces, cosine-ish directions), computed on the CPU with the same headers the kernels use."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from path_tracer_amd import standins
from oracle import pyoracle as O
from oracle.gltf_ingest import IngestedScene
standins.ensure_standins(os.path.join(ROOT, "assets"))
out = "/tmp/libbvh_lab.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread",
                       os.path.join(ROOT, "tools/lab/bvh_lab.cpp"), os.path.join(ROOT, "path_tracer_amd/csrc/bvh_build.cpp"), "-o", out])
L = C.CDLL(out)
L.lab_build.restype = C.c_void_p
L.lab_build.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
L.lab_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
ing = IngestedScene()
ing.import_gltf(os.path.join(ROOT, "assets/sponza/Sponza.gltf"), decode_image=lambda p: np.zeros((4, 4, 4), np.uint8))
inst = ing.instances(O.transform_matrix)
vs = []
for it in inst:
    off = it["off"]
    idx = ing.indices[off["index_offset"]:off["index_offset"] + off["index_count"]].reshape(-1, 3)
    p = ing.vertices["position"][off["vertex_offset"] + idx]
    m = it["xform"]
    w = np.empty_like(p)
    for r in range(3):
        w[..., r] = ((m[r, 0] * p[..., 0] + m[r, 1] * p[..., 1]) + m[r, 2] * p[..., 2]) + m[r, 3]
    vs.append(w.reshape(-1, 9))
v = np.ascontiguousarray(np.concatenate(vs), np.float32)
nn = C.c_uint64()
h = L.lab_build(v.ctypes.data, len(v), 8, C.byref(nn))
print("tris", len(v), "nodes", nn.value)
rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
# primary rays from the bench camera
def stats(rays, label):
    o = (C.c_uint64 * 8)()
    L.lab_trace(h, v.ctypes.data, rays.ctypes.data, len(rays), 1e-3, 1e4, o, None)
    r = list(o)
    print("%-10s rays %d hit %.2f nodes/ray %.2f empty %.1f%% stale-of-empty %.1f%% tris/ray %.2f slots/node %.2f hits/node %.2f" % (
        label, r[0], r[1] / r[0], r[2] / r[0], 100.0 * r[3] / r[2], 100.0 * r[4] / max(1, r[3]), r[5] / r[0], r[6] / r[2], r[7] / r[2]))
tri = v.reshape(-1, 3, 3)
area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
pick = rng.choice(len(tri), size=n, p=area / area.sum())
b = rng.random((n, 2)); su = np.sqrt(b[:, 0])
P = tri[pick, 0] * (1 - su)[:, None] + tri[pick, 1] * (su * (1 - b[:, 1]))[:, None] + tri[pick, 2] * (su * b[:, 1])[:, None]
N = np.cross(tri[pick, 1] - tri[pick, 0], tri[pick, 2] - tri[pick, 0]); N /= np.linalg.norm(N, axis=1)[:, None]
N *= rng.choice([-1.0, 1.0], size=n)[:, None]
D = rng.normal(size=(n, 3)); D /= np.linalg.norm(D, axis=1)[:, None]
D = D + N; D /= np.linalg.norm(D, axis=1)[:, None]
rays = np.ascontiguousarray(np.concatenate([P + N * 1e-3, D], 1), np.float32)
stats(rays, "bounce")
W, H = 480, 270
xs, ys = np.meshgrid(np.arange(W), np.arange(H))
fov = np.deg2rad(70.0); th = np.tan(fov / 2)
dx = ((xs + 0.5) / W * 2 - 1) * th * W / H; dy = (1 - (ys + 0.5) / H * 2) * th
# yaw -90: forward = R_y(-90) * (0,0,-1) = (1, 0, 0)?  use forward (+x), right (0,0,1)
dirs = np.stack([np.ones_like(dx), dy, dx], -1).reshape(-1, 3); dirs /= np.linalg.norm(dirs, axis=1)[:, None]
prim = np.ascontiguousarray(np.concatenate([np.tile(np.float32([-8, 1.6, 0]), (len(dirs), 1)), dirs], 1), np.float32)
stats(prim, "primary")
