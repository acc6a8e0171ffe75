"""Procedural stand-ins for the two geometry blobs the reference repository lacks.

/root/reference/.MISSING_LARGE_BLOBS lists assets/sponza/Sponza.bin and assets/spheres/spheres.bin
as absent, so configs C2/C3/C5 cannot use the authored geometry.  This module writes
a .bin with EXACTLY the byte layout the shipped .gltf declares (same accessors, counts, strides,
component types), so the ordinary glTF ingest loads it unchanged -- and the real file, if it ever
appears, drops in without code changes.  Results on these scenes are labelled "stand-in geometry".

Sponza: every primitive becomes a tube-topology grid (nu x nv quads around / along an axis, seam
duplicated) embedded in the primitive's accessor AABB:
  * slab-like primitives at ground level -> flattened tube = top + bottom sheet (floors)
  * everything else -> vertical-walled ring spanning the AABB (walls, columns, curtains, gallery
    floors; the roof becomes a parapet ring so the sky still lights the nave)
  * rings with a large footprint face inward (the camera is inside them), small ones outward
Walls carry a sinusoidal relief so the BVH is not a set of axis-aligned planes.  Triangles the
grid cannot absorb are emitted as degenerate (0,0,0) triangles and surplus vertices are parked on
the AABB corners, so the declared counts and min/max hold.  UVs tile with a 2 m period, i.e. run
past 1 (the reference's default sampler clamps them: todo.md:6).

Spheres: the ground cube and one sphere of 640 x 384 unshared quads (= 983 040 vertices, 491 520
triangles, the declared counts) scaled to the declared min/max; the reference dedups the three
"Mesh" meshes to the first one (geometry/mod.rs:62-65), the other two accessor sets get the same
data.
"""
import json
import os

import numpy as np

_COMP = {5120: "i1", 5121: "u1", 5122: "<i2", 5123: "<u2", 5125: "<u4", 5126: "<f4"}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}


def _write_accessor(buf, doc, idx, data):
    acc = doc["accessors"][idx]
    bv = doc["bufferViews"][acc["bufferView"]]
    dt = np.dtype(_COMP[acc["componentType"]])
    nc = _NCOMP[acc["type"]]
    data = np.ascontiguousarray(np.asarray(data).reshape(acc["count"], nc).astype(dt))
    start = bv.get("byteOffset", 0) + acc.get("byteOffset", 0)
    elem = dt.itemsize * nc
    stride = bv.get("byteStride", 0) or elem
    raw = data.view(np.uint8).reshape(acc["count"], elem)
    if stride == elem:
        buf[start:start + acc["count"] * elem] = raw.reshape(-1)
    else:
        view = np.lib.stride_tricks.as_strided(buf[start:], shape=(acc["count"], elem), strides=(stride, 1))
        view[:] = raw


def _grid_dims(n_tris, n_verts, aspect):
    """nu x nv quads with 2*nu*nv <= n_tris, (nu+1)*(nv+1) <= n_verts, nu/nv ~ aspect."""
    quads = n_tris // 2
    if quads < 1 or n_verts < 4:
        return 0, 0
    nu = max(1, int(round(np.sqrt(quads * aspect))))
    nu = min(nu, quads)
    nv = max(1, quads // nu)
    while (nu + 1) * (nv + 1) > n_verts and (nu > 1 or nv > 1):
        if nu >= nv and nu > 1:
            nu -= 1
        else:
            nv -= 1
    if (nu + 1) * (nv + 1) > n_verts:
        return 0, 0
    return nu, nv


def _tube_indices(nu, nv, n_tris):
    """2*nu*nv triangles over an (nu+1) x (nv+1) vertex grid, padded with degenerate triangles."""
    idx = np.zeros((n_tris, 3), dtype=np.uint32)
    if nu and nv:
        i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
        a = (i * (nv + 1) + j).reshape(-1)
        b = a + (nv + 1)
        q = np.stack([np.stack([a, b, b + 1], 1), np.stack([a, b + 1, a + 1], 1)], 1).reshape(-1, 3)
        idx[:len(q)] = q
    return idx.reshape(-1)


def _tube_vertices(nu, nv, n_verts, lo, hi, mode, inward, phase):
    """Embeds the grid in [lo,hi].  mode 'ring': axis y, rectangular cross-section with relief;
    mode 'sheet': flattened tube = top and bottom sheets."""
    lo = np.asarray(lo, np.float64)
    hi = np.asarray(hi, np.float64)
    ext = hi - lo
    pos = np.zeros((n_verts, 3), dtype=np.float64)
    nrm = np.zeros((n_verts, 3), dtype=np.float64)
    uv = np.zeros((n_verts, 2), dtype=np.float64)
    nrm[:, 1] = 1.0
    # surplus vertices sit on alternating AABB corners (keeps accessor min/max exact)
    pos[:] = np.where((np.arange(n_verts) % 2 == 0)[:, None], lo, hi)
    if nu and nv:
        s = np.linspace(0.0, 1.0, nu + 1)      # around
        t = np.linspace(0.0, 1.0, nv + 1)      # along
        S, T = np.meshgrid(s, t, indexing="ij")
        if mode == "ring":
            # rectangle perimeter parametrised by arc length, corners slightly rounded by the relief
            ex, ez = max(ext[0], 1e-9), max(ext[2], 1e-9)
            per = 2.0 * (ex + ez)
            d = S * per
            x = np.empty_like(d); z = np.empty_like(d); nx = np.zeros_like(d); nz = np.zeros_like(d)
            m0 = d <= ex
            m1 = (d > ex) & (d <= ex + ez)
            m2 = (d > ex + ez) & (d <= 2 * ex + ez)
            m3 = d > 2 * ex + ez
            x[m0] = d[m0]; z[m0] = 0.0; nz[m0] = -1.0
            x[m1] = ex; z[m1] = d[m1] - ex; nx[m1] = 1.0
            x[m2] = ex - (d[m2] - ex - ez); z[m2] = ez; nz[m2] = 1.0
            x[m3] = 0.0; z[m3] = ez - (d[m3] - 2 * ex - ez); nx[m3] = -1.0
            # relief pushes the wall inward by up to 6 % of the thinner side
            amp = 0.06 * min(ex, ez)
            bump = 0.5 * amp * (1.0 + np.sin(2 * np.pi * (7 * S + phase)) * np.cos(2 * np.pi * (3 * T + phase)))
            x = x - nx * bump
            z = z - nz * bump
            pos_g = np.stack([lo[0] + x, lo[1] + T * ext[1], lo[2] + z], -1)
            sign = -1.0 if inward else 1.0
            nrm_g = np.stack([sign * nx, np.zeros_like(nx), sign * nz], -1)
            uv_g = np.stack([d / 2.0, T * ext[1] / 2.0], -1)
        else:
            # sheet: s in [0,0.5] runs +z along the top, s in [0.5,1] runs back along the bottom
            top = S <= 0.5
            zz = np.where(top, S * 2.0, (1.0 - S) * 2.0)
            relief = 0.02 * ext[1] * np.sin(2 * np.pi * (5 * zz + phase)) * np.sin(2 * np.pi * (5 * T + phase))
            y = np.where(top, hi[1] - np.abs(relief), lo[1] + np.abs(relief))
            pos_g = np.stack([lo[0] + T * ext[0], y, lo[2] + zz * ext[2]], -1)
            nrm_g = np.stack([np.zeros_like(zz), np.where(top, 1.0, -1.0), np.zeros_like(zz)], -1)
            uv_g = np.stack([T * ext[0] / 2.0, zz * ext[2] / 2.0], -1)
        n = (nu + 1) * (nv + 1)
        pos[:n] = pos_g.reshape(-1, 3)
        nrm[:n] = nrm_g.reshape(-1, 3)
        uv[:n] = uv_g.reshape(-1, 2)
        # pin the declared min / max: first two surplus vertices, or the two grid corners
        pos = np.clip(pos, lo, hi)
    return pos.astype(np.float32), nrm.astype(np.float32), uv.astype(np.float32)


def make_sponza_bin(gltf_path, out_path):
    with open(gltf_path) as f:
        doc = json.load(f)
    acc = doc["accessors"]
    prims = doc["meshes"][0]["primitives"]
    buf = np.zeros(doc["buffers"][0]["byteLength"], dtype=np.uint8)
    # scene bounds in mesh units (the node scale is uniform)
    mins = np.min([acc[p["attributes"]["POSITION"]]["min"] for p in prims], axis=0)
    maxs = np.max([acc[p["attributes"]["POSITION"]]["max"] for p in prims], axis=0)
    scene_ext = maxs - mins
    footprint = scene_ext[0] * scene_ext[2]
    unit = 1.0 / float(doc["nodes"][0].get("scale", [1, 1, 1])[0])   # mesh units per metre

    # topology per index accessor (several primitives share one): from its first user, vertex
    # budget = the smallest vertex count among its users
    topo = {}
    for k, p in enumerate(prims):
        ia = p["indices"]
        pa = acc[p["attributes"]["POSITION"]]
        if ia not in topo:
            topo[ia] = dict(first=k, verts=pa["count"])
        else:
            topo[ia]["verts"] = min(topo[ia]["verts"], pa["count"])

    def classify(p):
        pa = acc[p["attributes"]["POSITION"]]
        lo, hi = np.array(pa["min"], np.float64), np.array(pa["max"], np.float64)
        ext = hi - lo
        slab = ext[1] <= 0.3 * min(ext[0], ext[2])
        ground = lo[1] < mins[1] + 0.1 * scene_ext[1]
        mode = "sheet" if (slab and ground) else "ring"
        inward = ext[0] * ext[2] > 0.10 * footprint
        return lo, hi, ext, mode, inward

    stats = dict(primitives=len(prims), triangles=0, degenerate=0, vertices=0)
    for ia, tp in topo.items():
        p = prims[tp["first"]]
        lo, hi, ext, mode, _ = classify(p)
        n_tris = acc[ia]["count"] // 3
        if mode == "ring":
            aspect = 2.0 * (ext[0] + ext[2]) / max(ext[1], 1e-6)
        else:
            aspect = 2.0 * ext[2] / max(ext[0], 1e-6)
        aspect = float(np.clip(aspect, 1.0 / 64, 64.0))
        nu, nv = _grid_dims(n_tris, tp["verts"], aspect)
        tp["nu"], tp["nv"] = nu, nv
        _write_accessor(buf, doc, ia, _tube_indices(nu, nv, n_tris))

    for k, p in enumerate(prims):
        tp = topo[p["indices"]]
        lo, hi, ext, mode, inward = classify(p)
        n_verts = acc[p["attributes"]["POSITION"]]["count"]
        pos, nrm, uv = _tube_vertices(tp["nu"], tp["nv"], n_verts, lo, hi, mode, inward, phase=0.137 * k)
        uv = uv / np.float32(unit)   # 2 m period
        # declared min/max must hold exactly: pin two vertices (surplus ones if any)
        n_grid = (tp["nu"] + 1) * (tp["nv"] + 1)
        if n_verts - n_grid >= 2:
            pos[n_grid] = np.array(acc[p["attributes"]["POSITION"]]["min"], np.float32)
            pos[n_grid + 1] = np.array(acc[p["attributes"]["POSITION"]]["max"], np.float32)
        _write_accessor(buf, doc, p["attributes"]["POSITION"], pos)
        _write_accessor(buf, doc, p["attributes"]["NORMAL"], nrm)
        if "TEXCOORD_0" in p["attributes"]:
            _write_accessor(buf, doc, p["attributes"]["TEXCOORD_0"], uv)
        if "TANGENT" in p["attributes"]:
            tan = np.zeros((n_verts, 4), np.float32); tan[:, 0] = 1.0; tan[:, 3] = 1.0
            _write_accessor(buf, doc, p["attributes"]["TANGENT"], tan)
        n_tris = acc[p["indices"]]["count"] // 3
        stats["triangles"] += n_tris
        stats["degenerate"] += n_tris - 2 * tp["nu"] * tp["nv"]
        stats["vertices"] += n_verts
    tmp = out_path + ".tmp%d" % os.getpid()
    buf.tofile(tmp)
    os.replace(tmp, out_path)
    return stats


def make_spheres_bin(gltf_path, out_path):
    with open(gltf_path) as f:
        doc = json.load(f)
    acc = doc["accessors"]
    buf = np.zeros(doc["buffers"][0]["byteLength"], dtype=np.uint8)
    # ground: unit cube, 4 vertices per face
    faces = [((1, 0, 0), (0, 1, 0), (0, 0, 1)), ((-1, 0, 0), (0, 0, 1), (0, 1, 0)),
             ((0, 1, 0), (0, 0, 1), (1, 0, 0)), ((0, -1, 0), (1, 0, 0), (0, 0, 1)),
             ((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (0, 1, 0), (1, 0, 0))]
    cp, cn, cuv, ci = [], [], [], []
    for n, a, b in faces:
        n, a, b = np.array(n, float), np.array(a, float), np.array(b, float)
        base = len(cp)
        for (sa, sb) in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
            cp.append(n + sa * a + sb * b); cn.append(n); cuv.append(((sa + 1) / 2, (sb + 1) / 2))
        ci += [base, base + 1, base + 2, base, base + 2, base + 3]
    cube = doc["meshes"][0]["primitives"][0]
    _write_accessor(buf, doc, cube["attributes"]["POSITION"], np.array(cp, np.float32))
    _write_accessor(buf, doc, cube["attributes"]["NORMAL"], np.array(cn, np.float32))
    _write_accessor(buf, doc, cube["attributes"]["TEXCOORD_0"], np.array(cuv, np.float32))
    _write_accessor(buf, doc, cube["indices"], np.array(ci, np.uint32))

    sph = doc["meshes"][1]["primitives"][0]
    pa = acc[sph["attributes"]["POSITION"]]
    n_verts, n_tris = pa["count"], acc[sph["indices"]]["count"] // 3
    n_lon, n_lat = 640, 384
    assert n_lon * n_lat * 4 == n_verts and n_lon * n_lat * 2 == n_tris
    radii = np.array(pa["max"], np.float64)
    i, j = np.meshgrid(np.arange(n_lon), np.arange(n_lat), indexing="ij")
    corners = []
    for di, dj in ((0, 0), (1, 0), (1, 1), (0, 1)):
        lon = 2 * np.pi * (i + di) / n_lon
        lat = np.pi * (j + dj) / n_lat
        d = np.stack([np.sin(lat) * np.cos(lon), np.cos(lat), np.sin(lat) * np.sin(lon)], -1)
        corners.append((d, np.stack([(i + di) / n_lon, (j + dj) / n_lat], -1)))
    d = np.stack([c[0] for c in corners], 2).reshape(-1, 3)       # quad-major, 4 corners each
    uv = np.stack([c[1] for c in corners], 2).reshape(-1, 2)
    pos = (d * radii).astype(np.float32)
    nrm = d.astype(np.float32)
    q = np.arange(n_lon * n_lat, dtype=np.uint32)[:, None] * 4
    idx = np.concatenate([q + 0, q + 2, q + 1, q + 0, q + 3, q + 2], 1).reshape(-1)   # outward winding
    _write_accessor(buf, doc, sph["indices"], idx)
    for mesh in doc["meshes"][1:]:
        a = mesh["primitives"][0]["attributes"]
        _write_accessor(buf, doc, a["POSITION"], pos)
        _write_accessor(buf, doc, a["NORMAL"], nrm)
        _write_accessor(buf, doc, a["TEXCOORD_0"], uv.astype(np.float32))
    tmp = out_path + ".tmp%d" % os.getpid()
    buf.tofile(tmp)
    os.replace(tmp, out_path)
    return dict(triangles=n_tris + 12, vertices=n_verts + 24)


def make_sponza_rotated(gltf_path, out_path, degrees=30.0):
    """The same scene with its one root node rotated about +Y (same .bin, same textures): every triangle of the
    axis-aligned stand-in becomes oblique to the world axes, which loosens every box of the acceleration structure --
    the `sponza_rot30_1080p` workload of bench.py."""
    import json
    import math
    doc = json.load(open(gltf_path))
    h = math.radians(degrees) * 0.5
    for n in doc["nodes"]:
        n["rotation"] = [0.0, math.sin(h), 0.0, math.cos(h)]
    doc.setdefault("asset", {})["extras"] = {"note": "Sponza.gltf with its root node rotated %g degrees about +Y (path_tracer_amd/standins.py)" % degrees}
    json.dump(doc, open(out_path, "w"))


def ensure_standins(assets_dir):
    """Writes the missing .bin files next to their .gltf (idempotent).  Returns the paths written."""
    made = []
    rot = os.path.join(assets_dir, "sponza", "Sponza_rot30.gltf")
    if not os.path.exists(rot):
        make_sponza_rotated(os.path.join(assets_dir, "sponza", "Sponza.gltf"), rot, 30.0)
        made.append(rot)
    sp = os.path.join(assets_dir, "sponza", "Sponza.bin")
    if not os.path.exists(sp):
        make_sponza_bin(os.path.join(assets_dir, "sponza", "Sponza.gltf"), sp)
        made.append(sp)
    sh = os.path.join(assets_dir, "spheres", "spheres.bin")
    if not os.path.exists(sh):
        make_spheres_bin(os.path.join(assets_dir, "spheres", "spheres.gltf"), sh)
        made.append(sh)
    return made


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    print(ensure_standins(os.path.join(here, "..", "assets")))
