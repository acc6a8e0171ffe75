#!/usr/bin/env python
"""Developer experiment: node visits / triangle tests per ray of the product's BVH8 on the Sponza stand-in under other
traversal policies and node encodings (tools/lab/trav_lab.cpp), for bounce-like and camera rays.  CPU only."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from path_tracer_amd import standins
from oracle import pyoracle as O
from oracle.gltf_ingest import IngestedScene
standins.ensure_standins(os.path.join(ROOT, "assets"))
out = "/tmp/libtrav_lab.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", os.path.join(ROOT, "tools/lab/trav_lab.cpp"),
                       os.path.join(ROOT, "tools/lab/bvh_lab.cpp"), os.path.join(ROOT, "path_tracer_amd/csrc/bvh_build.cpp"), "-o", out])
L = C.CDLL(out)
L.lab_build.restype = C.c_void_p
L.lab_build.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
L.tlab_create.restype = C.c_void_p
L.tlab_create.argtypes = [C.c_void_p, C.c_void_p]
L.tlab_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p]
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
T = L.tlab_create(h, v.ctypes.data)
rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
tri = v.reshape(-1, 3, 3)
area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
pick = rng.choice(len(tri), size=n, p=area / area.sum())
b = rng.random((n, 2)); su = np.sqrt(b[:, 0])
P = tri[pick, 0] * (1 - su)[:, None] + tri[pick, 1] * (su * (1 - b[:, 1]))[:, None] + tri[pick, 2] * (su * b[:, 1])[:, None]
N = np.cross(tri[pick, 1] - tri[pick, 0], tri[pick, 2] - tri[pick, 0]); N /= np.linalg.norm(N, axis=1)[:, None]
N *= rng.choice([-1.0, 1.0], size=n)[:, None]
D = rng.normal(size=(n, 3)); D /= np.linalg.norm(D, axis=1)[:, None]
D = D + N; D /= np.linalg.norm(D, axis=1)[:, None]
bounce = np.ascontiguousarray(np.concatenate([P + N * 1e-3, D], 1), np.float32)
W, H = 320, 180
xs, ys = np.meshgrid(np.arange(W), np.arange(H))
th = np.tan(np.deg2rad(70.0) / 2)
dx = ((xs + 0.5) / W * 2 - 1) * th * W / H; dy = (1 - (ys + 0.5) / H * 2) * th
dirs = np.stack([np.ones_like(dx), dy, dx], -1).reshape(-1, 3); dirs /= np.linalg.norm(dirs, axis=1)[:, None]
prim = np.ascontiguousarray(np.concatenate([np.tile(np.float32([-8, 1.6, 0]), (len(dirs), 1)), dirs], 1), np.float32)
print("tris", len(v), "nodes", nn.value)
for delay in (0, 1, 2, 4):
    L.tlab_set_delay(delay)
    for label, rays in (("bounce", bounce), ("primary", prim)):
        o = (C.c_uint64 * 4)()
        L.tlab_trace(T, rays.ctypes.data, len(rays), 1e-3, 1e4, 0, 0, 0, o)
        print("%-8s triangle tests %d node visits after the leaf hit: nodes/ray %6.2f tris/ray %5.2f" % (label, delay, o[2] / o[0], o[3] / o[0]))
for delay in (1, 2):
    L.tlab_set_delay(delay)
    for label, rays in (("bounce", bounce), ("primary", prim)):
        for cull in (0, 1, 5, 4):
            o = (C.c_uint64 * 4)()
            L.tlab_trace(T, rays.ctypes.data, len(rays), 1e-3, 1e4, 0, 0, cull, o)
            cc = (C.c_uint64 * 2)(); L.tlab_cull_counts(cc)
            print("%-8s delay %d, octant order, per-child stale culling %d: nodes/ray %6.2f tris/ray %5.2f  culled/ray: first child %.2f, later pops %.2f" % (label, delay, cull, o[2] / o[0], o[3] / o[0], cc[0] / o[0], cc[1] / o[0]))
L.tlab_set_delay(0)
for label, rays in (("bounce", bounce), ("primary", prim)):
    for boxes, order, cull, what in ((0, 0, 0, "product: 8-bit planes, octant order"), (0, 1, 0, "8-bit, distance order"), (0, 0, 1, "8-bit, octant order, stale culling"),
                                     (0, 1, 1, "8-bit, distance order, stale culling"), (0, 0, 2, "8-bit, octant order, group culling by the parent frame"), (0, 0, 4, "8-bit, octant, group = min over hit inner children"), (0, 0, 5, "8-bit, octant, group = min over the children left on the stack"), (2, 0, 0, "11-bit planes, octant order"), (1, 0, 0, "exact boxes, octant order"),
                                     (1, 1, 1, "exact boxes, distance order, stale culling")):
        o = (C.c_uint64 * 4)()
        L.tlab_trace(T, rays.ctypes.data, len(rays), 1e-3, 1e4, boxes, order, cull, o)
        print("%-8s %-46s hit %.3f nodes/ray %6.2f tris/ray %5.2f" % (label, what, o[1] / o[0], o[2] / o[0], o[3] / o[0]))
