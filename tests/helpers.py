"""Shared test plumbing: scenes from the oracle-side ingest are fed to BOTH the oracle and the
C ABI, so the two sides see byte-identical inputs."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "assets")
SEED = 0x1234567


def ingest(*items, decode_image=None, ref_quirks=True):
    """items: (gltf path relative to assets/, position offset)"""
    from oracle.gltf_ingest import IngestedScene
    ing = IngestedScene(ref_quirks=ref_quirks)
    for rel, off in items:
        ing.import_gltf(os.path.join(ASSETS, rel), off, decode_image=decode_image)
    return ing


def cornell():
    return ingest(("cornell/cornell.gltf", (0, 0, 0)))


def cornell_lit():
    # tools/make_cornell_lit.py: emissive ceiling quad, glass short box, rough-metal tall box
    return ingest(("cornell_lit/cornell_lit.gltf", (0, 0, 0)), ref_quirks=False)


def lion_boulder():
    # render_window.rs:412-426: lion at x=+2, boulder at x=-2
    return ingest(("lion_head_2k/lion_head_2k.gltf", (2, 0, 0)), ("boulder_01_2k/boulder_01_2k.gltf", (-2, 0, 0)))


def stress_scene(n_textures=70, n_confetti=12000, n_layers=24, seed=7):
    """Synthetic scene for the paths ordinary assets do not reach: more textures than the shading
    stage keeps in LDS (64), sRGB and linear textures, uv far outside [0, 1] (clamp), stacks of
    exactly coincident quads (equal-t ties -> lowest (instance, primitive)), and leaf-heavy nodes
    that overflow the traversal's per-wave triangle queue."""
    from oracle.gltf_ingest import IngestedScene, MATERIAL_DTYPE, VERTEX_DTYPE
    rng = np.random.default_rng(seed)
    ing = IngestedScene()
    for t in range(n_textures):
        ing.textures.append((rng.integers(0, 256, size=(4 + t % 3, 3 + t % 5, 4), dtype=np.uint8), t % 2))
    mats = []
    for t in range(n_textures):
        m = np.zeros(1, dtype=MATERIAL_DTYPE)
        m["base_color"] = rng.random(4, dtype=np.float32)
        m["has_base_texture"] = 1 if t % 4 != 3 else 0
        m["base_texture_indice"] = t
        m["metallic"] = 0.0
        m["roughness"] = np.float32([1.0, 0.0, 0.5, 0.25][t % 4])
        mats.append(ing.add_material(m))

    def add_entity(name, pos3, idx, mat, xf=((0, 0, 0), (0, 0, 0, 1), (1, 1, 1))):
        v = np.zeros(len(pos3), dtype=VERTEX_DTYPE)
        v["position"] = pos3
        from oracle.gltf_ingest import calculate_normals
        v["normal"] = calculate_normals(np.asarray(pos3, np.float32), np.asarray(idx, np.uint32))
        v["uv"] = rng.uniform(-1.5, 2.5, size=(len(pos3), 2)).astype(np.float32)
        geom = ing._geometry_create(name, v, np.asarray(idx, np.uint32))
        ing.entities.append((geom, (np.float32(xf[0]), np.float32(xf[1]), np.float32(xf[2])), mat))

    # confetti: small random triangles in a 4 m cube, one entity per material
    per = n_confetti // n_textures
    for t in range(n_textures):
        c = rng.uniform(-2, 2, size=(per, 1, 3)).astype(np.float32)
        tri = (c + rng.normal(0, 0.08, size=(per, 3, 3)).astype(np.float32)).reshape(-1, 3)
        add_entity("confetti%d" % t, tri, np.arange(per * 3), mats[t])
    # coincident layers: the same quad n_layers times, as separate instances AND duplicated inside one geometry
    quad = np.float32([[-1.5, -1.5, -2.5], [1.5, -1.5, -2.5], [1.5, 1.5, -2.5], [-1.5, 1.5, -2.5]])
    dup_idx = np.tile(np.uint32([0, 1, 2, 0, 2, 3]), n_layers)
    add_entity("stack", quad, dup_idx, mats[1])
    for l in range(n_layers):
        add_entity("layer", quad, np.uint32([0, 1, 2, 0, 2, 3]), mats[(5 + l) % n_textures])   # same ident: one geometry, many instances
    # a floor so that paths bounce
    floor = np.float32([[-6, -2.2, -6], [6, -2.2, -6], [6, -2.2, 6], [-6, -2.2, 6]])
    add_entity("floor", floor, np.uint32([0, 2, 1, 0, 3, 2]), mats[3])
    return ing


def upload(ctx, ing, oracle, options=0, with_material_ext=False):
    """Drives the C ABI the way the reference's loaders drive Scene/Renderer."""
    from path_tracer_amd import capi
    if options:
        ctx.set_scene_options(options)
    for rgba, srgb in ing.textures:
        ctx.add_texture(rgba, srgb)
    for i, m in enumerate(ing.materials):
        mi = ctx.add_material(m)
        if with_material_ext:
            ctx.set_material_ext(mi, ing.material_ext[i])
            if float(ing.normal_scale[i]) != 1.0:
                ctx.set_material_normal_scale(mi, ing.normal_scale[i])
    geoms = {}
    for ident, (gid, v_first, v_count, i_first, i_count) in ing.geometries.items():
        fv = ctx.add_vertices(ing.vertices[v_first:v_first + v_count])
        fi = ctx.add_indices(ing.indices[i_first:i_first + i_count])
        assert fv == v_first and fi == i_first
        geoms[gid] = ctx.create_geometry(fv, v_count, fi, i_count)
    inst = ing.instances(oracle.transform_matrix).astype(capi.INSTANCE_DTYPE)
    for i in range(len(inst)):
        inst[i]["geometry"] = geoms[int(inst[i]["geometry"])]
    ctx.set_instances(inst)
    return inst


def copy_struct(dst_type, src):
    import ctypes as C
    dst = dst_type()
    C.memmove(C.byref(dst), C.byref(src), C.sizeof(dst_type))
    return dst


def gpu_render(ctx, oracle, cam, width, height, n_samples, seed=SEED, first_sample=0, z_near=0.1, z_far=100.0,
               params=None, batch=True):
    from path_tracer_amd import capi
    ctx.resize(width, height)
    gcam = copy_struct(capi.Camera, cam)
    u = capi.Uniforms(1, seed, first_sample, z_near, z_far, (0, 0, 0))
    if batch:
        ctx.render_batch(u, gcam, n_samples, params)
    else:
        for s in range(n_samples):
            u.accumulated_count = first_sample + s
            ctx.render(u, gcam, params)
    return dict(albedo=ctx.read_image(capi.IMAGE_ALBEDO), normal=ctx.read_image(capi.IMAGE_NORMAL),
                depth=ctx.read_image(capi.IMAGE_DEPTH), ids=ctx.read_image(capi.IMAGE_PRIMARY_ID))


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_images_bit_equal(got, want, what):
    for k in ("ids", "normal", "depth", "albedo"):
        g, w = bits(got[k]), bits(want[k])
        bad = np.argwhere(g != w)
        assert len(bad) == 0, "%s: %s differs at %d elements, first %s: got %s want %s" % (
            what, k, len(bad), bad[0], got[k][tuple(bad[0])], want[k][tuple(bad[0])])


def rel_mse(a, b, eps=1e-2):
    """mean over pixels/channels of (a-b)^2 / (b^2 + eps) on RGB."""
    a = np.asarray(a, np.float64)[..., :3]
    b = np.asarray(b, np.float64)[..., :3]
    return float(np.mean((a - b) ** 2 / (b * b + eps)))


def write_mr_quad_gltf(dirpath, with_maps=False):
    """A lit test scene written from scratch: a 4 x 4 m floor quad with uv in [0, 2] carrying a base-colour texture
    and a metallicRoughnessTexture (procedural PNGs, G = roughness, B = metallic), under a small emissive quad.
    with_maps: the floor also names a normalTexture (a field of bumps, tangent space, +Y up) and the lamp an
    emissiveTexture (a striped sRGB image).  Returns the .gltf path."""
    import json
    import struct
    from PIL import Image
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:32, 0:32]
    mr = np.zeros((32, 32, 3), np.uint8)
    mr[..., 1] = np.where((xx // 4 + yy // 4) % 2 == 0, 40, 230)          # roughness checker
    mr[..., 2] = np.where(xx < 16, 255, 0)                                 # metallic half
    mr[..., 0] = rng.integers(0, 256, (32, 32))                            # occlusion channel: unused
    Image.fromarray(mr).save(os.path.join(dirpath, "mr.png"))
    Image.fromarray(rng.integers(60, 256, (16, 16, 3), dtype=np.uint8)).save(os.path.join(dirpath, "base.png"))
    if with_maps:
        gy, gx = np.mgrid[0:64, 0:64].astype(np.float64)
        hx, hy = 0.7 * np.cos(gx * (2 * np.pi / 16)), 0.7 * np.sin(gy * (2 * np.pi / 8))      # slopes of a bumpy height field
        n = np.stack([-hx, -hy, np.ones_like(hx)], -1)
        n /= np.linalg.norm(n, axis=-1, keepdims=True)
        Image.fromarray(np.uint8(np.round((n * 0.5 + 0.5) * 255))).save(os.path.join(dirpath, "normal.png"))
        em = np.zeros((8, 8, 3), np.uint8)
        em[:, ::2] = (255, 255, 255); em[:, 1::2] = (40, 90, 200)
        Image.fromarray(em).save(os.path.join(dirpath, "emit.png"))

    def quad(y, half, flip):
        p = [[-half, y, -half], [half, y, -half], [half, y, half], [-half, y, half]]
        uv = [[0, 0], [2, 0], [2, 2], [0, 2]]
        idx = [0, 2, 1, 0, 3, 2] if not flip else [0, 1, 2, 0, 2, 3]
        return p, uv, idx
    blob = b""
    views, accessors, meshes = [], [], []
    for name, (p, uv, idx), mat in (("floor", quad(0.0, 2.0, False), 0), ("lamp", quad(2.0, 0.3, True), 1)):
        nrm = [[0, 1, 0]] * 4 if name == "floor" else [[0, -1, 0]] * 4
        first = len(accessors)
        for data, comp, typ in ((p, "f", "VEC3"), (nrm, "f", "VEC3"), (uv, "f", "VEC2"), (idx, "H", "SCALAR")):
            flat = [x for row in data for x in row] if typ != "SCALAR" else data
            raw = struct.pack("<%d%s" % (len(flat), comp), *flat)
            raw += b"\0" * (-len(raw) % 4)
            views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(raw)})
            acc = {"bufferView": len(views) - 1, "componentType": 5126 if comp == "f" else 5123, "count": len(data), "type": typ}
            if typ == "VEC3" and data is p:
                acc["min"] = [min(r[i] for r in p) for i in range(3)]; acc["max"] = [max(r[i] for r in p) for i in range(3)]
            accessors.append(acc)
            blob += raw
        meshes.append({"name": name, "primitives": [{"attributes": {"POSITION": first, "NORMAL": first + 1, "TEXCOORD_0": first + 2},
                                                       "indices": first + 3, "material": mat}]})
    open(os.path.join(dirpath, "mrquad.bin"), "wb").write(blob)
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0, 1]}],
           "nodes": [{"name": "floor", "mesh": 0}, {"name": "lamp", "mesh": 1}], "meshes": meshes,
           "materials": [{"name": "floor", "pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicRoughnessTexture": {"index": 1},
                                                                      "roughnessFactor": 0.9, "metallicFactor": 0.8}},
                         {"name": "lamp", "emissiveFactor": [1.0, 0.9, 0.7], "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 12.0}},
                          "pbrMetallicRoughness": {"baseColorFactor": [0, 0, 0, 1]}}],
           "textures": [{"source": 0, "sampler": 0}, {"source": 1, "sampler": 1}],
           "samplers": [{"magFilter": 9729, "wrapS": 10497, "wrapT": 10497}, {"magFilter": 9728, "wrapS": 33071, "wrapT": 33071}],
           "images": [{"uri": "base.png"}, {"uri": "mr.png"}],
           "buffers": [{"uri": "mrquad.bin", "byteLength": len(blob)}], "bufferViews": views, "accessors": accessors}
    if with_maps:
        doc["materials"][0]["normalTexture"] = {"index": 2, "scale": 0.6}
        doc["materials"][1]["emissiveTexture"] = {"index": 3}
        doc["textures"] += [{"source": 2, "sampler": 0}, {"source": 3, "sampler": 1}]
        doc["images"] += [{"uri": "normal.png"}, {"uri": "emit.png"}]
    path = os.path.join(dirpath, "mrquad.gltf")
    json.dump(doc, open(path, "w"))
    return path


RCCL_STUB = os.path.join(ROOT, "tests", "native", "librccl_stub.so")


def build_rccl_stub(force=False):
    """tests/native/rccl_stub.cpp -> tests/native/librccl_stub.so (plain g++ against the HIP runtime API; built on the CPU
    box by __graft_entry__.build(), travels to the GPU box with the tree)."""
    import subprocess
    src = os.path.join(ROOT, "tests", "native", "rccl_stub.cpp")
    if not force and os.path.exists(RCCL_STUB) and os.path.getmtime(RCCL_STUB) >= os.path.getmtime(src):
        return RCCL_STUB
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           src, "-o", RCCL_STUB, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    return RCCL_STUB
