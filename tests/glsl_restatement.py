"""A SECOND, independent CPU restatement of the reference's shaders -- TEST INFRASTRUCTURE.

oracle/pt_oracle.c and the HIP kernels were written by one hand and share their expression trees, so a shared
misreading of the GLSL would pass every oracle-vs-GPU test.  This file restates the same path from the GLSL text
again, in numpy float32, sharing NO code with the oracle or the product:

  raygen        /root/reference/src/shaders/rgen.glsl:20-141
  closest hit   /root/reference/src/shaders/rchit.glsl:158-254
  miss          /root/reference/src/shaders/rmiss.glsl:8-34
  rng, sampling /root/reference/src/shaders/shared.glsl:156-168,194-207,236-271
  host side     /root/reference/src/renderer.rs:915-955,970-980 (instances, camera), src/scene/gltf.rs:14-129 (import),
                src/scene/mod.rs:557-580 (Transform -> Mat4), src/material.rs:34-90

Deliberately different where the reference leaves the arithmetic to its driver or its maths crate: its own glTF
reader, float64 matrix inverses (glam inverts in fp32), numpy's sin / cos (the oracle pins a polynomial), and a
Moeller-Trumbore triangle test, brute force over all triangles (the oracle uses edge functions and a BVH).  So the
two agree to rounding, not bit for bit; tests/test_oracle_cpu.py states the tolerances.

Textures (round 3): base-colour textures are sampled the way the reference's descriptors make the hardware sample them --
rchit.glsl:192-198 `texture(textures[idx], uv).xyz` with the factor dropped; graphics.rs:453-459 SamplerCreateInfo::default()
= nearest, clamp-to-edge, LOD 0; graphics.rs:492-516 8-bit colour -> R8G8B8A8_SRGB, i.e. the IEC 61966-2-1 decode on
fetch -- written here from those lines and the Vulkan rules for unnormalised texel coordinates (floor(u * W), clamped),
with the decode evaluated in float64 per texel (the oracle and the kernels read a 256-entry fp32 table).  The decoded
8-bit texels themselves are an INPUT (`decode`): which JPEG decoder produced them is a third-party boundary of its own
(image 0.25.6 / zune-jpeg in the reference), not something a restatement of the shaders can arbitrate.
"""
import json
import os
import struct

import numpy as np

F = np.float32


def _quat_to_mat3(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def load_gltf(path, offset=(0.0, 0.0, 0.0), decode=None):
    """Root nodes with a mesh only (gltf.rs:32-34), one entity per primitive: world-space triangle vertices,
    OBJECT-space vertex normals (rchit.glsl:175-179 never transforms them), material (base colour, roughness, and the
    base-colour texture's RGBA8 texels if it has one: material.rs:43-68).  `offset` is added to the node translation
    (gltf.rs:36-44); `decode(path) -> HxWx{3,4} uint8` decodes an image file."""
    doc = json.load(open(path))
    base = os.path.dirname(path)
    blobs = [open(os.path.join(base, b["uri"]), "rb").read() for b in doc["buffers"]]

    def accessor(i):
        a = doc["accessors"][i]
        bv = doc["bufferViews"][a["bufferView"]]
        fmt, size = {5126: ("f", 4), 5123: ("H", 2), 5125: ("I", 4), 5121: ("B", 1)}[a["componentType"]]
        ncomp = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}[a["type"]]
        stride = bv.get("byteStride", size * ncomp)
        off = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        raw = blobs[bv["buffer"]]
        rows = [struct.unpack_from("<%d%s" % (ncomp, fmt), raw, off + k * stride) for k in range(a["count"])]
        return np.array(rows)

    tris, nrms, uvs, inst, prim, mats = [], [], [], [], [], []
    entity = 0
    scene = doc["scenes"][doc.get("scene", 0)]
    for ni in scene["nodes"]:
        node = doc["nodes"][ni]
        if "mesh" not in node:
            continue
        T = np.array(node.get("translation", [0, 0, 0]), float) + np.array(offset, float)
        R = _quat_to_mat3(node.get("rotation", [0, 0, 0, 1]))
        S = np.array(node.get("scale", [1, 1, 1]), float)
        M = (R * S[None, :]).astype(F)                    # Mat4::from_scale_rotation_translation, upper 3x3, as fp32
        Tf = T.astype(F)
        for p in doc["meshes"][node["mesh"]]["primitives"]:
            pos = accessor(p["attributes"]["POSITION"]).astype(F)
            nrm = accessor(p["attributes"]["NORMAL"]).astype(F)
            idx = accessor(p["indices"]).astype(np.int64).reshape(-1, 3) if "indices" in p else np.arange(len(pos)).reshape(-1, 3)
            m = doc["materials"][p["material"]].get("pbrMetallicRoughness", {}) if "material" in p else {}
            tex = None
            if "baseColorTexture" in m:
                img = doc["images"][doc["textures"][m["baseColorTexture"]["index"]]["source"]]
                px = np.asarray(decode(os.path.join(base, img["uri"])))
                if px.shape[2] == 3:                      # graphics.rs:507-516: RGB8 is padded to RGBA8 with alpha 255
                    px = np.concatenate([px, np.full(px.shape[:2] + (1,), 255, np.uint8)], 2)
                tex = px
            tuv = accessor(p["attributes"]["TEXCOORD_0"]).astype(F) if "TEXCOORD_0" in p["attributes"] else np.zeros((len(pos), 2), F)
            world = (pos @ M.T + Tf).astype(F)            # as_instance transform applied by the acceleration structure
            tris.append(world[idx]); nrms.append(nrm[idx]); uvs.append(tuv[idx])
            inst.append(np.full(len(idx), entity)); prim.append(np.arange(len(idx)))
            mats.append((np.array(m.get("baseColorFactor", [1, 1, 1, 1])[:3], F), F(m.get("roughnessFactor", 1.0)), tex))
            entity += 1
    return dict(tris=np.concatenate(tris).astype(F), nrms=np.concatenate(nrms).astype(F), uvs=np.concatenate(uvs).astype(F),
                inst=np.concatenate(inst), prim=np.concatenate(prim), mats=mats)


def merge(*scenes):
    """Several imports into one scene, entity indices continuing (Scene::import_gltf called once per file)."""
    out = dict(tris=[], nrms=[], uvs=[], inst=[], prim=[], mats=[])
    for sc in scenes:
        out["inst"].append(sc["inst"] + len(out["mats"]))
        for k in ("tris", "nrms", "uvs", "prim"):
            out[k].append(sc[k])
        out["mats"] += sc["mats"]
    return {k: (np.concatenate(v) if k != "mats" else v) for k, v in out.items()}


def _texture(tex, uv):
    """texture(sampler2D, uv).xyz through VkSamplerCreateInfo defaults (nearest, clamp to edge, LOD 0) on an
    R8G8B8A8_SRGB image: texel (floor(u W), floor(v H)) clamped into the image, sRGB -> linear on fetch."""
    h, w = tex.shape[:2]
    x = np.clip(np.floor(uv[:, 0].astype(F) * F(w)), 0, w - 1).astype(np.int64)
    y = np.clip(np.floor(uv[:, 1].astype(F) * F(h)), 0, h - 1).astype(np.int64)
    c = tex[y, x, :3].astype(np.float64) / 255.0
    lin = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    return lin.astype(F)


def camera(pos, yaw_deg, fov_deg, aspect, near, far):
    """view_inverse, proj_inverse of renderer.rs:970-980: glam perspective_rh (depth 0..1) and look_to_rh."""
    h = np.deg2rad(yaw_deg) / 2
    R = _quat_to_mat3([0, np.sin(h), 0, np.cos(h)])
    fwd, up = R @ np.array([0, 0, -1.0]), R @ np.array([0, 1.0, 0])
    f = fwd / np.linalg.norm(fwd)
    s = np.cross(f, up); s /= np.linalg.norm(s)
    u = np.cross(s, f)
    eye = np.array(pos, float)
    view = np.array([[s[0], s[1], s[2], -s @ eye], [u[0], u[1], u[2], -u @ eye], [-f[0], -f[1], -f[2], f @ eye], [0, 0, 0, 1]])
    t = 1.0 / np.tan(np.deg2rad(np.float32(fov_deg)).astype(float) / 2)
    proj = np.array([[t / aspect, 0, 0, 0], [0, t, 0, 0], [0, 0, far / (near - far), near * far / (near - far)], [0, 0, -1, 0]])
    return np.linalg.inv(view).astype(F), np.linalg.inv(proj).astype(F)


def _hash(s):
    s = s.astype(np.uint32)
    s = s ^ (s >> np.uint32(16)); s = s * np.uint32(0x7FEB352D)
    s = s ^ (s >> np.uint32(15)); s = s * np.uint32(0x846CA68B)
    return s ^ (s >> np.uint32(16))


def _rand(uv, depth, acc, seed):
    st = _hash(_hash(uv[:, 0].view(np.uint32)) ^ (_hash(uv[:, 1].view(np.uint32)) * np.uint32(7919)))
    st = _hash(st ^ np.uint32(depth)); st = _hash(st ^ np.uint32(acc)); st = _hash(st ^ np.uint32(seed & 0xFFFFFFFF))
    return (st.astype(F) / F(4294967296.0)).astype(F)


def _normalize(v):
    return (v / np.sqrt(np.sum(v * v, axis=1, dtype=F), dtype=F)[:, None]).astype(F)


def _cosine_hemisphere(uv, depth, acc, seed, n):
    u1, u2 = _rand(uv, depth, acc, seed + 7919), _rand(uv, depth, acc, seed + 16381)
    r = np.sqrt(u1, dtype=F); phi = (F(2.0) * F(3.1415926535) * u2).astype(F)
    x, y, z = r * np.cos(phi, dtype=F), r * np.sin(phi, dtype=F), np.sqrt(np.maximum(F(0), F(1) - u1), dtype=F)
    sg = np.where(n[:, 2] >= 0, F(1), F(-1)).astype(F)            # sign(n.z), 0 -> 1
    a = F(-1) / (sg + n[:, 2]); b = n[:, 0] * n[:, 1] * a
    tx = np.stack([F(1) + sg * n[:, 0] * n[:, 0] * a, sg * b, -sg * n[:, 0]], 1)
    bt = np.stack([b, sg + n[:, 1] * n[:, 1] * a, -n[:, 1]], 1)
    return (tx * x[:, None] + bt * y[:, None] + n * z[:, None]).astype(F)


def _trace(sc, o, d, tmin, tmax):
    """closest hit of each ray over all triangles (Moeller-Trumbore, both faces); ties -> lowest (instance, primitive)"""
    if len(o) * len(sc["tris"]) > (1 << 22):                        # bounded temporaries: a block of rays at a time
        step = max(1, (1 << 22) // len(sc["tris"]))
        parts = [_trace(sc, o[i:i + step], d[i:i + step], tmin, tmax) for i in range(0, len(o), step)]
        return tuple(np.concatenate([p[k] for p in parts]) for k in range(5))
    v0, e1, e2 = sc["tris"][:, 0][None], (sc["tris"][:, 1] - sc["tris"][:, 0])[None], (sc["tris"][:, 2] - sc["tris"][:, 0])[None]
    D, O = d[:, None, :], o[:, None, :]
    p = np.cross(D, e2); det = np.sum(e1 * p, -1)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = F(1) / det
        tv = O - v0
        u = np.sum(tv * p, -1) * inv
        q = np.cross(tv, e1)
        v = np.sum(D * q, -1) * inv
        t = np.sum(e2 * q, -1) * inv
    ok = (det != 0) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > tmin) & (t < tmax)
    t = np.where(ok, t, np.inf)
    order = np.lexsort((sc["prim"], sc["inst"]))                  # argmin over this order: first minimum = lowest ids
    best = order[np.argmin(t[:, order], axis=1)]
    rows = np.arange(len(o))
    hit = np.isfinite(t[rows, best])
    return hit, best, t[rows, best].astype(F), u[rows, best].astype(F), v[rows, best].astype(F)


def render(sc, view_inv, proj_inv, W, H, spp, seed, max_depth=16, z_near=0.1, z_far=100.0):
    ys, xs = np.mgrid[0:H, 0:W]
    xs, ys = xs.ravel(), ys.ravel()
    uv = np.stack([(xs.astype(F) + F(0.5)) / F(W), ((H - ys - 1).astype(F) + F(0.5)) / F(H)], 1).astype(F)
    dd = (uv * F(2) - F(1)).astype(F)
    n = len(xs)
    tgt = (np.stack([dd[:, 0], dd[:, 1], np.ones(n, F), np.ones(n, F)], 1) @ proj_inv.T).astype(F)
    tn = _normalize(tgt[:, :3])
    dir0 = _normalize((np.concatenate([tn, np.zeros((n, 1), F)], 1) @ view_inv.T).astype(F)[:, :3])
    org0 = np.tile((np.array([0, 0, 0, 1], F) @ view_inv.T)[:3].astype(F), (n, 1))
    albedo = np.zeros((n, 3), F); normal_img = np.zeros((n, 4), F); depth_img = np.zeros(n, F)
    ids = np.full((n, 2), 0xFFFFFFFF, np.uint32)
    for acc in range(spp):
        o, d = org0.copy(), dir0.copy()
        att = np.ones((n, 3), F)
        alive = np.ones(n, bool)
        for depth in range(max_depth):
            idx = np.nonzero(alive)[0]
            if len(idx) == 0:
                break
            hit, tri, t, bu, bv = _trace(sc, o[idx], d[idx], F(0.001), F(10000.0))
            pay_att = np.zeros((len(idx), 3), F); nrm = np.zeros((len(idx), 3), F)
            dist = np.full(len(idx), F(3.402823466e38)); done = np.ones(len(idx), bool)
            new_o, new_d = o[idx].copy(), d[idx].copy()
            m = ~hit                                              # rmiss.glsl
            dy = d[idx][m][:, 1]
            sky = np.where((dy > 0)[:, None], (np.ones(3, F)[None] * (F(1) - dy)[:, None] + np.array([0.25, 0.5, 1.0], F)[None] * dy[:, None]),
                           np.full(3, F(0.03))[None]).astype(F)
            pay_att[m] = sky
            h = np.nonzero(hit)[0]                                # rchit.glsl
            if len(h):
                tr = tri[h]
                b0 = (F(1) - bu[h] - bv[h]).astype(F)
                N = _normalize((sc["nrms"][tr, 0] * b0[:, None] + sc["nrms"][tr, 1] * bu[h][:, None] + sc["nrms"][tr, 2] * bv[h][:, None]).astype(F))
                rd, ro = d[idx][h], o[idx][h]
                wp = (ro + rd * t[h][:, None]).astype(F)
                base = np.array([sc["mats"][i][0] for i in sc["inst"][tr]], F)
                tuv = (sc["uvs"][tr, 0] * b0[:, None] + sc["uvs"][tr, 1] * bu[h][:, None] + sc["uvs"][tr, 2] * bv[h][:, None]).astype(F)
                for e in np.unique(sc["inst"][tr]):                # rchit.glsl:192-198: the texel replaces the factor
                    if sc["mats"][e][2] is not None:
                        sel = sc["inst"][tr] == e
                        base[sel] = _texture(sc["mats"][e][2], tuv[sel])
                rough = np.array([sc["mats"][i][1] for i in sc["inst"][tr]], F)
                refl = (rd - F(2) * np.sum(N * rd, 1, dtype=F)[:, None] * N).astype(F)
                rr = _normalize(_cosine_hemisphere(uv[idx][h], depth, acc, seed, N))
                sd = (refl * (F(1) - rough)[:, None] + rr * rough[:, None]).astype(F)
                up = np.sum(sd * N, 1, dtype=F) > 0
                pay_att[h] = np.where(up[:, None], base, F(0))
                done[h] = ~up
                so = (wp + N * F(0.001)).astype(F)
                new_o[h], new_d[h], nrm[h] = so, sd, N
                dist[h] = np.sqrt(np.sum((so - ro) ** 2, 1, dtype=F), dtype=F)
            att[idx] = (att[idx] * pay_att).astype(F)
            if depth == 0 and acc == 0:
                depth_img[idx] = np.clip((dist - F(z_near)) / (F(z_far) - F(z_near)), 0, 1).astype(F)
                normal_img[idx] = np.concatenate([nrm, np.ones((len(idx), 1), F)], 1)
                ids[idx[hit], 0] = sc["inst"][tri[hit]]; ids[idx[hit], 1] = sc["prim"][tri[hit]]
            o[idx], d[idx] = new_o, new_d
            alive[idx] = ~done                                    # rgen.glsl:101-103: the product so far is kept at max depth
        a = F(acc)
        albedo = (att / (a + F(1)) + albedo * a / (a + F(1))).astype(F)
    img = lambda x, c: x.reshape(H, W, c) if c > 1 else x.reshape(H, W)
    return dict(albedo=np.concatenate([albedo, np.ones((n, 1), F)], 1).reshape(H, W, 4), normal=img(normal_img, 4),
                depth=img(depth_img, 1), ids=ids.reshape(H, W, 2))
