"""numpy restatement of the reference's glTF ingest -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/src/scene/gltf.rs:14-152 (import_gltf, calculate_normals),
src/scene/geometry/mod.rs:54-103 (Geometry::create ident dedup),
src/material.rs:34-90 (Material::from_gltf), src/graphics.rs:383-516 (texture
format / ident cache) and src/renderer.rs:912-955 (entity order -> Offsets rows
and TLAS instances).  The `gltf` crate itself is third-party (gltf 1.4.1, not
vendored): only the subset the shipped assets use is restated -- external .bin
buffers, float VEC2/VEC3 attributes, u8/u16/u32 indices, byteStride, TRS or
matrix node transforms.  Image decoding (image 0.25.6 -> zune-jpeg / png in
the reference) is PIL here: parity unpinned at the texel level.

Nothing under path_tracer_amd/ imports this module.
"""
import json
import os

import numpy as np

VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("_p0", "<f4"), ("normal", "<f4", 3), ("_p1", "<f4"),
                         ("uv", "<f4", 2), ("padding", "<f4", 2)])
MATERIAL_DTYPE = np.dtype([("base_color", "<f4", 4), ("has_base_texture", "<u4"), ("base_texture_indice", "<u4"),
                           ("metallic", "<f4"), ("roughness", "<f4")])
OFFSETS_DTYPE = np.dtype([("vertex_offset", "<u4"), ("vertex_count", "<u4"), ("material_offset", "<u4"),
                          ("padding", "<u4"), ("index_offset", "<u4"), ("index_count", "<u4")])
INSTANCE_DTYPE = np.dtype([("xform", "<f4", (3, 4)), ("geometry", "<u4"), ("off", OFFSETS_DTYPE)])
# build extension: what glTF materials carry beyond material.rs:70-76
MATERIAL_EXT_DTYPE = np.dtype([("emissive", "<f4", 3), ("transmission", "<f4"), ("ior", "<f4"), ("mr_texture", "<u4"),
                               ("normal_texture", "<u4"), ("emissive_texture", "<u4")])
assert MATERIAL_EXT_DTYPE.itemsize == 32
assert VERTEX_DTYPE.itemsize == 48 and MATERIAL_DTYPE.itemsize == 32
assert OFFSETS_DTYPE.itemsize == 24 and INSTANCE_DTYPE.itemsize == 76

_COMP = {5120: "i1", 5121: "u1", 5122: "<i2", 5123: "<u2", 5125: "<u4", 5126: "<f4"}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def _accessor(doc, buffers, idx):
    acc = doc["accessors"][idx]
    bv = doc["bufferViews"][acc["bufferView"]]
    dt = np.dtype(_COMP[acc["componentType"]])
    nc = _NCOMP[acc["type"]]
    start = bv.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = bv.get("byteStride", 0) or dt.itemsize * nc
    buf = buffers[bv["buffer"]]
    count = acc["count"]
    out = np.lib.stride_tricks.as_strided(
        np.frombuffer(buf, dtype=np.uint8, count=(count - 1) * stride + dt.itemsize * nc, offset=start),
        shape=(count, dt.itemsize * nc), strides=(stride, 1))
    out = np.ascontiguousarray(out).view(dt).reshape(count, nc)
    if acc.get("normalized") and dt.kind in "ui":
        out = out.astype(np.float32) / np.float32(np.iinfo(dt).max)
    return out


def _vnormalize(v):
    """glam Vec3::normalize: v * (1/sqrt(dot)) in f32."""
    d = (v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]) + v[..., 2] * v[..., 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        return v * (np.float32(1.0) / np.sqrt(d))[..., None]


def calculate_normals(positions, indices):
    """gltf.rs:132-152 -- per-face unit normal summed per vertex, then normalised."""
    p = positions.astype(np.float32)
    tri = indices.reshape(-1, 3)
    a, b, c = p[tri[:, 0]], p[tri[:, 1]], p[tri[:, 2]]
    e1, e2 = b - a, c - a
    fn = np.stack([e1[:, 1] * e2[:, 2] - e2[:, 1] * e1[:, 2],
                   e1[:, 2] * e2[:, 0] - e2[:, 2] * e1[:, 0],
                   e1[:, 0] * e2[:, 1] - e2[:, 0] * e1[:, 1]], axis=1).astype(np.float32)
    fn = _vnormalize(fn)
    normals = np.zeros_like(p)
    # sequential += in (triangle, corner) order, as the reference's loop does
    np.add.at(normals, tri.reshape(-1), np.repeat(fn, 3, axis=0))
    return _vnormalize(normals)


def _decompose(node):
    """gltf Node::transform().decomposed() -> (translation, rotation xyzw, scale)."""
    if "matrix" in node:
        m = np.array(node["matrix"], dtype=np.float32).reshape(4, 4).T  # column-major in file
        t = m[:3, 3].copy()
        sx, sy, sz = (np.linalg.norm(m[:3, i]) for i in range(3))
        if np.linalg.det(m[:3, :3]) < 0:
            sx = -sx
        r = m[:3, :3] / np.array([sx, sy, sz], dtype=np.float32)
        # rotation matrix -> quaternion
        tr = r[0, 0] + r[1, 1] + r[2, 2]
        if tr > 0:
            s = np.sqrt(tr + 1.0) * 2
            q = [(r[2, 1] - r[1, 2]) / s, (r[0, 2] - r[2, 0]) / s, (r[1, 0] - r[0, 1]) / s, 0.25 * s]
        elif r[0, 0] > r[1, 1] and r[0, 0] > r[2, 2]:
            s = np.sqrt(1.0 + r[0, 0] - r[1, 1] - r[2, 2]) * 2
            q = [0.25 * s, (r[0, 1] + r[1, 0]) / s, (r[0, 2] + r[2, 0]) / s, (r[2, 1] - r[1, 2]) / s]
        elif r[1, 1] > r[2, 2]:
            s = np.sqrt(1.0 + r[1, 1] - r[0, 0] - r[2, 2]) * 2
            q = [(r[0, 1] + r[1, 0]) / s, 0.25 * s, (r[1, 2] + r[2, 1]) / s, (r[0, 2] - r[2, 0]) / s]
        else:
            s = np.sqrt(1.0 + r[2, 2] - r[0, 0] - r[1, 1]) * 2
            q = [(r[0, 2] + r[2, 0]) / s, (r[1, 2] + r[2, 1]) / s, 0.25 * s, (r[1, 0] - r[0, 1]) / s]
        return t, np.array(q, dtype=np.float32), np.array([sx, sy, sz], dtype=np.float32)
    t = np.array(node.get("translation", [0, 0, 0]), dtype=np.float32)
    q = np.array(node.get("rotation", [0, 0, 0, 1]), dtype=np.float32)
    s = np.array(node.get("scale", [1, 1, 1]), dtype=np.float32)
    return t, q, s


def _quat_mul(a, b):
    ax, ay, az, aw = (np.float32(v) for v in a)
    bx, by, bz, bw = (np.float32(v) for v in b)
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], dtype=np.float32)


def _quat_rotate(q, v):
    b = np.asarray(q[:3], np.float32)
    w = np.float32(q[3])
    v = np.asarray(v, np.float32)
    return (v * (w * w - np.dot(b, b)) + b * (np.dot(v, b) * np.float32(2)) + np.cross(b, v) * (w * np.float32(2))).astype(np.float32)


def _compose(parent, child):
    """parent * child for TRS triples.  Exact when the parent's scale is uniform (every shipped asset);
    a non-uniformly scaled rotated parent has no TRS form and keeps the per-axis product."""
    pt, pq, ps = parent
    ct, cq, cs = child
    return (pt + _quat_rotate(pq, ps * ct)).astype(np.float32), _quat_mul(pq, cq), (ps * cs).astype(np.float32)


class IngestedScene:
    """What the reference's Scene + Renderer hold after import_gltf calls:
    shared vertex / index / material buffers, the bindless texture table and the
    entity list (geometry, transform, material) in push order."""

    def __init__(self, ref_quirks=True):
        # ref_quirks=False (the --no-ref-quirks ingest): child nodes are imported with their
        # accumulated transforms (gltf.rs:32-34 stops at the roots) and the glTF sampler's wrap mode is
        # honoured (graphics.rs:446-451 reads it and drops it)
        self.ref_quirks = ref_quirks
        self.vertices = np.zeros(0, dtype=VERTEX_DTYPE)
        self.indices = np.zeros(0, dtype=np.uint32)
        self.materials = np.zeros(0, dtype=MATERIAL_DTYPE)
        self.material_ext = np.zeros(0, dtype=MATERIAL_EXT_DTYPE)
        self.normal_scale = np.zeros(0, dtype=np.float32)
        self.textures = []          # list of (rgba8 HxWx4 uint8, srgb flag)
        self.texture_lookup = {}    # ident -> bindless index   (renderer.rs:867-873)
        self.geometries = {}        # ident -> (id, v_first, v_count, i_first, i_count)
        self.entities = []          # (geometry tuple, (pos, quat, scale), material index)

    # scene/mod.rs:270-337 / 405-467: bump-append, returns the element offset
    def add_vertices(self, v):
        first = len(self.vertices)
        self.vertices = np.concatenate([self.vertices, v])
        return first

    def add_indices(self, i):
        first = len(self.indices)
        self.indices = np.concatenate([self.indices, i.astype(np.uint32)])
        return first

    def add_material(self, m, ext=None, normal_scale=1.0):
        self.materials = np.concatenate([self.materials, m])
        if ext is None:
            ext = np.zeros(1, dtype=MATERIAL_EXT_DTYPE)
            ext["ior"] = 1.5
        self.material_ext = np.concatenate([self.material_ext, ext])
        self.normal_scale = np.concatenate([self.normal_scale, np.float32([normal_scale])])    # glTF normalTexture.scale
        return len(self.materials) - 1

    def import_gltf(self, path, position_offset=(0.0, 0.0, 0.0), decode_image=None):
        """gltf.rs:14-129.  Root nodes of every scene that carry a mesh; no child recursion."""
        with open(path, "r") as f:
            doc = json.load(f)
        base = os.path.dirname(path)
        buffers = []
        for b in doc.get("buffers", []):
            with open(os.path.join(base, b["uri"]), "rb") as f:
                buffers.append(f.read())
        offset = np.array(position_offset, dtype=np.float32)
        for scene in doc.get("scenes", []):
            work = [(ni, None) for ni in scene.get("nodes", [])]
            while work:
                ni, parent = work.pop(0)
                node = doc["nodes"][ni]
                t, q, s = _decompose(node)
                if not self.ref_quirks:
                    # children in depth-first order after their parent; TRS composed parent * child
                    if parent is not None:
                        t, q, s = _compose(parent, (t, q, s))
                    work = [(c, (t, q, s)) for c in node.get("children", [])] + work
                if "mesh" not in node:
                    continue
                mesh_index = node["mesh"]
                mesh = doc["meshes"][mesh_index]
                transform = (t + offset, q, s)                                   # gltf.rs:37-41
                for prim_index, prim in enumerate(mesh["primitives"]):
                    mat_index = self._material_from_gltf(doc, prim.get("material"), base, decode_image)
                    attrs = prim["attributes"]
                    positions = _accessor(doc, buffers, attrs["POSITION"]).astype(np.float32)
                    if "indices" in prim:                                        # gltf.rs:60-68
                        indices = _accessor(doc, buffers, prim["indices"]).reshape(-1).astype(np.uint32)
                    else:
                        indices = np.arange(len(positions), dtype=np.uint32)
                    if "NORMAL" in attrs:                                        # gltf.rs:78-91
                        normals = _accessor(doc, buffers, attrs["NORMAL"]).astype(np.float32)
                    else:
                        normals = calculate_normals(positions, indices)
                    if "TEXCOORD_0" in attrs:                                    # gltf.rs:94-98
                        uvs = _accessor(doc, buffers, attrs["TEXCOORD_0"]).astype(np.float32)
                    else:
                        uvs = np.zeros((len(positions), 2), dtype=np.float32)
                    n = min(len(positions), len(normals), len(uvs))               # zip() truncates
                    verts = np.zeros(n, dtype=VERTEX_DTYPE)
                    verts["position"] = positions[:n]
                    verts["normal"] = normals[:n]
                    verts["uv"] = uvs[:n]
                    ident = "%s_%d" % (mesh.get("name", "%d" % mesh_index), prim_index)   # gltf.rs:112-116
                    geom = self._geometry_create(ident, verts, indices)
                    self.entities.append((geom, transform, mat_index))

    def _geometry_create(self, ident, verts, indices):
        """geometry/mod.rs:54-103: an existing ident returns the FIRST geometry unchanged."""
        if ident in self.geometries:
            return self.geometries[ident]
        v_first = self.add_vertices(verts)
        i_first = self.add_indices(indices)
        geom = (len(self.geometries), v_first, len(verts), i_first, len(indices))
        self.geometries[ident] = geom
        return geom

    def _texture_from_gltf(self, doc, tex, base, decode_image, srgb):
        """Texture::from_gltf + Renderer::add_texture (graphics.rs:383-516, renderer.rs:867-904): bindless index,
        cached by ident."""
        gtex = doc["textures"][tex["index"]]
        image = doc["images"][gtex["source"]]
        ident = "gltf_%s" % image["uri"]                                     # material.rs:59-61
        flags = 1 if srgb else 0                                             # R8G8B8A8_SRGB / _UNORM
        if not srgb:
            ident += "#unorm"
        if not self.ref_quirks:
            samp = doc.get("samplers", [{}])[gtex["sampler"]] if "sampler" in gtex else {}
            if samp.get("wrapS", 10497) == 10497 and samp.get("wrapT", 10497) == 10497:
                flags |= 2                                                   # wrap = repeat (glTF default)
                ident += "#repeat"
            if samp.get("magFilter", 9729) == 9729:
                flags |= 4                                                   # LINEAR (also when the sampler leaves it open)
                ident += "#linear"
        if ident not in self.texture_lookup:                                 # graphics.rs:396-398
            rgba = np.asarray((decode_image or decode_image_pil)(os.path.join(base, image["uri"])), np.uint8)
            if rgba.shape[2] == 3:                                            # graphics.rs:507-516
                rgba = np.concatenate([rgba, np.full(rgba.shape[:2] + (1,), 255, np.uint8)], axis=2)
            rgba = np.ascontiguousarray(rgba)
            self.textures.append((rgba, flags))
            self.texture_lookup[ident] = len(self.textures) - 1
        return self.texture_lookup[ident]

    def _material_from_gltf(self, doc, mat_idx, base, decode_image):
        """material.rs:34-90.  Every primitive appends a new Material row (no dedup)."""
        m = np.zeros(1, dtype=MATERIAL_DTYPE)
        ext = np.zeros(1, dtype=MATERIAL_EXT_DTYPE)
        ext["ior"] = 1.5
        normal_scale = 1.0
        pbr = {}
        if mat_idx is not None:
            gm = doc["materials"][mat_idx]
            pbr = gm.get("pbrMetallicRoughness", {})
            x = gm.get("extensions", {})
            strength = np.float32(x.get("KHR_materials_emissive_strength", {}).get("emissiveStrength", 1.0))
            ext["emissive"] = np.array(gm.get("emissiveFactor", [0, 0, 0]), dtype=np.float32) * strength
            ext["transmission"] = np.float32(x.get("KHR_materials_transmission", {}).get("transmissionFactor", 0.0))
            ext["ior"] = np.float32(x.get("KHR_materials_ior", {}).get("ior", 1.5))
        m["base_color"] = np.array(pbr.get("baseColorFactor", [1, 1, 1, 1]), dtype=np.float32)
        m["metallic"] = np.float32(pbr.get("metallicFactor", 1.0))
        m["roughness"] = np.float32(pbr.get("roughnessFactor", 1.0))
        tex = pbr.get("baseColorTexture")
        if tex is not None:
            index = self._texture_from_gltf(doc, tex, base, decode_image, srgb=True)
            m["has_base_texture"] = 1
            m["base_texture_indice"] = index
        mrt = pbr.get("metallicRoughnessTexture")
        if mrt is not None and not self.ref_quirks:          # build extension (lit integrator); the reference never reads it
            uri = doc["images"][doc["textures"][mrt["index"]]["source"]]["uri"]
            if os.path.exists(os.path.join(base, uri)):     # optional: without the file the factors stand alone
                ext["mr_texture"] = self._texture_from_gltf(doc, mrt, base, decode_image, srgb=False) + 1
        if mat_idx is not None and not self.ref_quirks:      # build extensions (lit integrator), both optional like the one above
            for key, field, srgb in (("normalTexture", "normal_texture", False), ("emissiveTexture", "emissive_texture", True)):
                t = doc["materials"][mat_idx].get(key)
                if t is not None and os.path.exists(os.path.join(base, doc["images"][doc["textures"][t["index"]]["source"]]["uri"])):
                    ext[field] = self._texture_from_gltf(doc, t, base, decode_image, srgb=srgb) + 1
                    if key == "normalTexture":
                        normal_scale = float(t.get("scale", 1.0))
        return self.add_material(m, ext, normal_scale)

    def instances(self, xform_fn):
        """renderer.rs:912-955: one Offsets row + TLAS instance per entity, in push order.
        xform_fn(pos, quat, scale) -> 3x4 row-major (scene/mod.rs:557-566 + renderer.rs:1416-1423)."""
        inst = np.zeros(len(self.entities), dtype=INSTANCE_DTYPE)
        for i, (geom, (pos, quat, scale), mat) in enumerate(self.entities):
            gid, v_first, v_count, i_first, i_count = geom
            inst[i]["xform"] = xform_fn(pos, quat, scale)
            inst[i]["geometry"] = gid
            inst[i]["off"] = (v_first, v_count, mat, 0, i_first, i_count)
        return inst


def decode_image_pil(path):
    """8-bit RGB/RGBA -> RGBA8 with A=255 pad (graphics.rs:507-516)."""
    from PIL import Image
    im = Image.open(path)
    if im.mode not in ("RGB", "RGBA"):
        im = im.convert("RGBA" if "A" in im.mode or im.mode == "P" else "RGB")
    a = np.asarray(im, dtype=np.uint8)
    if a.shape[2] == 3:
        a = np.concatenate([a, np.full(a.shape[:2] + (1,), 255, dtype=np.uint8)], axis=2)
    return np.ascontiguousarray(a)
