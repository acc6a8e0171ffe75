"""CPU tests of the oracle (oracle/pt_oracle.c, oracle/gltf_ingest.py): known answers, golden
fixtures, and the semantics the build defines where the reference is silent."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from helpers import ASSETS, SEED, bits, cornell, ingest, lion_boulder

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KAT = json.load(open(os.path.join(GOLDEN, "kat.json")))


def f32(b):
    return np.uint32(b).view(np.float32)


def fp(a):
    return np.asarray(a, np.float32).ctypes.data_as(C.POINTER(C.c_float))


def py_hash(s):
    """Independent restatement of shared.glsl:156-161."""
    s &= 0xFFFFFFFF
    s ^= s >> 16
    s = (s * 0x7feb352d) & 0xFFFFFFFF
    s ^= s >> 15
    s = (s * 0x846ca68b) & 0xFFFFFFFF
    s ^= s >> 16
    return s


def test_hash_known_answers(oracle):
    L = oracle.lib()
    for i, want in KAT["hash"]:
        assert L.orc_hash(i) == want == py_hash(i)
    for a, b, want in KAT["hash2"]:
        assert L.orc_hash2(a, b) == want == py_hash(py_hash(a) ^ ((py_hash(b) * 7919) & 0xFFFFFFFF))


def test_rand_known_answers_and_range(oracle):
    L = oracle.lib()
    for ub, vb, d, a, s, want in KAT["rand"]:
        got = L.orc_rand(f32(ub), f32(vb), d, a, s)
        assert bits(np.float32(got)) == want
        # shared.glsl:194-207 restated in Python
        st = py_hash(py_hash(ub) ^ ((py_hash(vb) * 7919) & 0xFFFFFFFF))
        st = py_hash(st ^ d); st = py_hash(st ^ a); st = py_hash(st ^ s)
        assert np.float32(got) == np.float32(np.float32(st) / np.float32(4294967296.0))
    r = [L.orc_rand(np.float32(0.3), np.float32(0.6), 1, k, 5) for k in range(2000)]
    assert 0.0 <= min(r) and max(r) <= 1.0 and 0.45 < np.mean(r) < 0.55


def test_sincos_accuracy_and_golden(oracle):
    L = oracle.lib()
    for pb, sb, cb in KAT["sincos"]:
        s, c = C.c_float(), C.c_float()
        L.orc_sincos(f32(pb), C.byref(s), C.byref(c))
        assert (bits(np.float32(s.value)), bits(np.float32(c.value))) == (sb, cb)
    for phi in np.linspace(0, 2 * math.pi, 1001):
        s, c = C.c_float(), C.c_float()
        L.orc_sincos(np.float32(phi), C.byref(s), C.byref(c))
        p = float(np.float32(phi))
        assert abs(s.value - math.sin(p)) < 3e-7 and abs(c.value - math.cos(p)) < 3e-7   # far inside GLSL's 2^-11


def test_create_basis_and_hemisphere_sample(oracle):
    L = oracle.lib()
    for nb, tb, bb in KAT["create_basis"]:
        n = np.array([f32(x) for x in nb], np.float32)
        t, b = np.zeros(3, np.float32), np.zeros(3, np.float32)
        L.orc_create_basis(fp(n), fp(t), fp(b))
        assert [bits(x) for x in t] == tb and [bits(x) for x in b] == bb
        nn = n / np.linalg.norm(n)
        if abs(np.linalg.norm(n) - 1) < 1e-6:       # Duff et al.: orthonormal for unit normals
            assert abs(np.dot(t, nn)) < 1e-6 and abs(np.dot(b, nn)) < 1e-6 and abs(np.dot(t, b)) < 1e-6
            assert abs(np.linalg.norm(t) - 1) < 1e-6 and abs(np.linalg.norm(b) - 1) < 1e-6
    for ub, vb, d, a, s, nb, ob in KAT["cosine_hemisphere_sample"]:
        n = np.array([f32(x) for x in nb], np.float32)
        out = np.zeros(3, np.float32)
        L.orc_cosine_hemisphere_sample(f32(ub), f32(vb), d, a, s, fp(n), fp(out))
        assert [bits(x) for x in out] == ob
        if abs(np.linalg.norm(n) - 1) < 1e-6:
            assert np.dot(out, n) >= -1e-6 and abs(np.linalg.norm(out) - 1) < 1e-5
    # cosine-weighted: E[cos theta] = 2/3
    n = np.array([0, 0, 1], np.float32)
    zs = []
    for k in range(4000):
        out = np.zeros(3, np.float32)
        L.orc_cosine_hemisphere_sample(np.float32(0.37), np.float32(0.11), 0, k, 99, fp(n), fp(out))
        zs.append(out[2])
    assert abs(np.mean(zs) - 2 / 3) < 0.02


def test_sky_and_srgb(oracle):
    L = oracle.lib()
    for db, ob in KAT["sky_color"]:
        out = np.zeros(3, np.float32)
        L.orc_sky_color(fp([f32(x) for x in db]), fp(out))
        assert [bits(x) for x in out] == ob
    out = np.zeros(3, np.float32)
    L.orc_sky_color(fp([0, -0.5, 0]), fp(out))
    assert np.allclose(out, 0.03)                       # rmiss.glsl:13
    L.orc_sky_color(fp([0, 1.0, 0]), fp(out))
    assert np.allclose(out, [0.25, 0.5, 1.0])           # rmiss.glsl:10 at direction.y = 1
    lut = [bits(np.float32(L.orc_srgb_to_linear(i))) for i in range(256)]
    assert lut == KAT["srgb_lut"]
    assert L.orc_srgb_to_linear(0) == 0.0 and L.orc_srgb_to_linear(255) == 1.0
    assert abs(L.orc_srgb_to_linear(128) - 0.21586) < 1e-4


def test_camera_and_transform_matrices(oracle):
    cam = oracle.camera_matrices([0, 1.0, 3.9], oracle.yaw_quat(0), np.deg2rad(np.float32(40)), 1.0, 0.1, 1000.0)
    assert [int(x) for x in np.frombuffer(bytes(cam), np.uint32)] == KAT["camera_cornell"]
    cam2 = oracle.camera_matrices([-8, 1.6, 0], oracle.yaw_quat(-90), np.deg2rad(np.float32(70)), 16 / 9, 0.1, 1000.0)
    assert [int(x) for x in np.frombuffer(bytes(cam2), np.uint32)] == KAT["camera_sponza"]
    # against float64 linear algebra: glam perspective_rh / look_to_rh semantics (renderer.rs:970-980)
    vi = np.array(cam2.view_inverse, np.float64).reshape(4, 4).T
    pi = np.array(cam2.proj_inverse, np.float64).reshape(4, 4).T
    assert np.allclose(vi[:3, 3], [-8, 1.6, 0], atol=1e-6)
    assert np.allclose(vi[:3, :3] @ np.array([0, 0, -1.0]), [1, 0, 0], atol=1e-6)       # yaw -90 deg looks down +X
    t = math.tan(math.radians(70) / 2)
    tgt = pi @ np.array([0.5, -0.25, 1, 1.0])
    d = tgt[:3] / np.linalg.norm(tgt[:3])
    want = np.array([0.5 * (16 / 9) * t, -0.25 * t, -1.0]); want /= np.linalg.norm(want)   # SURVEY.md 8 a8 closed form
    assert np.allclose(d, want, atol=1e-5)
    m = oracle.transform_matrix([0, 0.00454026460647583, 0.004760861396789551], [-0.7072679400444031, 0, 0, 0.7069455981254578], [1, 1, 1])
    assert [int(x) for x in m.view(np.uint32).ravel()] == KAT["transform_cornell_node"]
    assert np.allclose(m[:, :3] @ m[:, :3].T, np.eye(3), atol=1e-6)


def test_cornell_golden_image(oracle):
    g = np.load(os.path.join(GOLDEN, "cornell_64x64_16spp_depth8.npz"))
    sc = oracle.scene_from_ingest(cornell())
    cam = oracle.camera_matrices([0, 1.0, 3.9], oracle.yaw_quat(0), np.deg2rad(np.float32(40)), 1.0, 0.1, 1000.0)
    r = sc.render(cam, 64, 64, 16, SEED, params=oracle.default_params(max_depth=8, flags=1))
    for k in ("ids", "normal", "depth", "albedo"):
        assert np.array_equal(bits(r[k]), bits(g[k])), k
    assert [r["stats"][k] for k in ("rays", "hits", "samples")] == list(g["stats"])
    # sample-by-sample accumulation equals one call (rgen.glsl:124-141 running mean)
    alb = np.zeros((64, 64, 4), np.float32)
    for s in range(16):
        sc.render(cam, 64, 64, 1, SEED, first_sample=s, params=oracle.default_params(max_depth=8, flags=1), albedo=alb, want_ids=False)
    assert np.array_equal(bits(alb), bits(g["albedo"]))
    # thread count does not change the result
    r1 = sc.render(cam, 64, 64, 2, SEED, threads=1)
    r8 = sc.render(cam, 64, 64, 2, SEED, threads=8)
    assert np.array_equal(bits(r1["albedo"]), bits(r8["albedo"]))


def test_lion_boulder_golden_ids_and_bvh_equals_brute_force(oracle):
    g = np.load(os.path.join(GOLDEN, "lion_boulder_160x90_ids.npz"))
    sc = oracle.scene_from_ingest(lion_boulder())
    cam = oracle.camera_matrices([2.0, 0.25, 0.6], oracle.yaw_quat(0), np.deg2rad(np.float32(70)), 160 / 90, 0.1, 1000.0)
    r = sc.render(cam, 160, 90, 2, SEED)
    assert np.array_equal(r["ids"], g["ids"]) and np.array_equal(bits(r["depth"]), bits(g["depth"]))
    assert [r["stats"][k] for k in ("rays", "hits", "samples")] == list(g["stats"])
    # the oracle's BVH path returns exactly what its brute-force path returns (subset of rays)
    rng = np.random.default_rng(7)
    for _ in range(60):
        o = np.array([2.0, 0.25, 0.6]) + rng.normal(size=3) * 0.05
        d = np.array([0, 0, -1.0]) + rng.normal(size=3) * 0.25
        assert sc.trace(o, d) == sc.trace(o, d, flags=1)


def _one_tri_scene(oracle, tris, normals=None):
    from oracle.gltf_ingest import INSTANCE_DTYPE, MATERIAL_DTYPE, VERTEX_DTYPE
    v = np.zeros(len(tris) * 3, VERTEX_DTYPE)
    v["position"] = np.asarray(tris, np.float32).reshape(-1, 3)
    v["normal"] = (0, 0, 1) if normals is None else normals
    idx = np.arange(len(v), dtype=np.uint32)
    m = np.zeros(1, MATERIAL_DTYPE); m["base_color"] = (0.5, 0.5, 0.5, 1); m["roughness"] = 1.0
    inst = np.zeros(1, INSTANCE_DTYPE)
    inst[0]["xform"] = np.eye(4, dtype=np.float32)[:3]
    inst[0]["off"] = (0, len(v), 0, 0, 0, len(idx))
    return oracle.OracleScene(v, idx, m, inst)


def test_trace_semantics(oracle):
    tri = [[(-1, -1, -2), (1, -1, -2), (0, 1, -2)]]
    sc = _one_tri_scene(oracle, tri * 2 + [[(-1, -1, -3), (1, -1, -3), (0, 1, -3)]])
    h = sc.trace([0, 0, 0], [0, 0, -1])
    assert h is not None and h[0] == 2.0 and (h[3], h[4]) == (0, 0)        # equal t -> lowest primitive id
    assert abs(h[1] - 0.25) < 1e-6 and abs(h[2] - 0.5) < 1e-6              # (u, v) = weights of vertex 1, 2
    assert sc.trace([0, 0, -4], [0, 0, 1])[4] == 2                         # both faces are hit (no culling)
    assert sc.trace([0, 0, 0], [0, 0, -2.0])[0] == 1.0                     # t is parametric in an unnormalised direction
    assert sc.trace([0, 0, 0], [0, 0, -1], t_min=2.0)[0] == 3.0            # t_min < t strictly
    assert sc.trace([0, 0, 0], [0, 0, -1], t_max=2.0) is None              # t < t_max strictly
    assert sc.trace([0, 0, 0], [0, 0, 1]) is None
    assert sc.trace([0, 0, 0], [np.nan, 0, -1]) is None and sc.trace([np.inf, 0, 0], [0, 0, -1]) is None
    deg = _one_tri_scene(oracle, [[(0, 0, -2), (0, 0, -2), (0, 0, -2)], [(-1, -1, -5), (1, -1, -5), (0, 1, -5)]])
    assert deg.trace([0, 0, 0], [0, 0, -1])[4] == 1                        # degenerate triangles are never hit
    # a ray through a shared edge hits (watertight edge functions), ties resolved by id
    quad = _one_tri_scene(oracle, [[(-1, -1, -2), (1, -1, -2), (1, 1, -2)], [(-1, -1, -2), (1, 1, -2), (-1, 1, -2)]])
    for x in np.linspace(-0.9, 0.9, 37):
        assert quad.trace([x, x, 0], [0, 0, -1]) is not None


def test_ingest_semantics(oracle, tmp_path):
    ing = cornell()
    assert (len(ing.vertices), len(ing.indices), len(ing.materials), len(ing.entities)) == (64, 96, 8, 8)
    inst = ing.instances(oracle.transform_matrix)
    assert list(inst["off"]["material_offset"]) == list(range(8)) and list(inst["geometry"]) == list(range(8))
    # three meshes named "Mesh" -> one geometry, three instances, four materials (geometry/mod.rs:62-65)
    from path_tracer_amd import standins
    standins.ensure_standins(ASSETS)
    sp = ingest(("spheres/spheres.gltf", (0, 0, 0)))
    assert len(sp.geometries) == 2 and len(sp.entities) == 4 and len(sp.materials) == 4
    assert len(sp.vertices) == 24 + 983040 and len(sp.indices) == 36 + 1474560
    off = sp.instances(oracle.transform_matrix)["off"]
    assert off["vertex_offset"][1] == off["vertex_offset"][2] == off["vertex_offset"][3] == 24
    assert list(off["material_offset"]) == [0, 1, 2, 3]
    # a primitive without NORMAL / TEXCOORD_0 / indices: generated normals, zero uvs, sequential indices (gltf.rs:60-98)
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 0], [0, 1, 0], [-1, 0, 0]], np.float32)
    (tmp_path / "t.bin").write_bytes(pos.tobytes())
    doc = {"asset": {"version": "2.0"}, "scenes": [{"nodes": [0, 1]}],
           "nodes": [{"mesh": 0, "translation": [1, 2, 3], "children": [2]}, {"name": "no mesh"}, {"mesh": 0}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0}}]}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 6, "type": "VEC3"}],
           "bufferViews": [{"buffer": 0, "byteLength": 72}], "buffers": [{"uri": "t.bin", "byteLength": 72}]}
    (tmp_path / "t.gltf").write_text(json.dumps(doc))
    from oracle.gltf_ingest import IngestedScene
    t = IngestedScene()
    t.import_gltf(str(tmp_path / "t.gltf"), (10, 0, 0))
    assert len(t.entities) == 1                                            # root nodes with a mesh only; child node 2 is not visited
    assert list(t.indices) == [0, 1, 2, 3, 4, 5] and np.all(t.vertices["uv"] == 0)
    assert np.allclose(t.vertices["normal"], [0, 0, 1])
    assert np.allclose(t.entities[0][1][0], [11, 2, 3])                     # translation + position_offset (gltf.rs:39)
    assert t.materials["roughness"][0] == 1.0 and t.materials["metallic"][0] == 1.0 and t.materials["has_base_texture"][0] == 0
    assert list(t.geometries) == ["0_0"]                                   # ident = mesh index when unnamed (gltf.rs:112-116)


def test_oracle_against_an_independent_numpy_restatement_of_the_glsl(oracle):
    """tests/glsl_restatement.py restates rgen / rchit / rmiss / shared.glsl a second time, in numpy float32, with no
    code in common with oracle/pt_oracle.c: its own glTF reader, float64 matrix inverses, numpy's sin / cos, a
    Moeller-Trumbore brute-force closest hit.  The reference itself cannot run here (parity unpinned), so this does
    not pin the oracle -- it removes the failure mode of ONE author misreading the GLSL once for both the oracle and
    the kernels.  Agreement bar: primary (instance, primitive) ids exactly; normal and depth G-buffers to 2e-7; the
    accumulated throughput to 1e-5 relative per pixel (measured: 2.5e-7) -- the two differ in rounding only."""
    import glsl_restatement as G
    from helpers import ASSETS, SEED, cornell
    sc = G.load_gltf(os.path.join(ASSETS, "cornell/cornell.gltf"))
    assert len(sc["tris"]) == 32 and len(sc["mats"]) == 8
    ing = cornell()
    # glossier copies of two materials on both sides, so that reflect() and the unnormalised mix() (rchit.glsl:209-220) matter
    for row, rough in ((1, 0.5), (5, 0.0), (6, 0.25)):
        ing.materials[row]["roughness"] = rough
        sc["mats"][row] = (sc["mats"][row][0], np.float32(rough), None)
    osc = oracle.scene_from_ingest(ing)
    for (w, h), spp, depth, pos, yaw in (((48, 48), 6, 8, [0, 1.0, 3.9], 0.0), ((40, 24), 3, 16, [0.3, 1.2, 3.2], 9.0)):
        vi, pi = G.camera(pos, yaw, 40.0, w / h, 0.1, 1000.0)
        got = G.render(sc, vi, pi, w, h, spp, SEED, max_depth=depth)
        cam = oracle.camera_matrices(pos, oracle.yaw_quat(yaw), np.deg2rad(np.float32(40)), w / h, 0.1, 1000.0)
        assert np.abs(np.array(cam.view_inverse, np.float32).reshape(4, 4).T - vi).max() < 1e-6
        assert np.abs(np.array(cam.proj_inverse, np.float32).reshape(4, 4).T - pi).max() < 1e-6
        want = osc.render(cam, w, h, spp, SEED, params=oracle.default_params(max_depth=depth))
        assert np.array_equal(got["ids"], want["ids"])
        assert (want["ids"][..., 0] != 0xFFFFFFFF).mean() > 0.5
        assert np.abs(got["normal"] - want["normal"]).max() <= 2e-7
        assert np.abs(got["depth"] - want["depth"]).max() <= 2e-7
        a, b = got["albedo"][..., :3].astype(np.float64), want["albedo"][..., :3].astype(np.float64)
        rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-3)
        # a path off the mirror-like materials can take another branch under different rounding (a grazing edge, a
        # scatter direction on the horizon): measured 1 pixel of 2304; everything else agrees to 4e-7
        bad = rel.max(-1) > 1e-5
        assert bad.mean() <= 0.002, (bad.sum(), rel.max())
        assert rel[~bad].max() <= 2e-6


def test_oracle_textures_against_the_independent_restatement(oracle):
    """The texture path (rchit.glsl:192-198, graphics.rs:453-459,492-516) had one author on both sides.
    tests/glsl_restatement.py now samples base-colour textures from the cited lines alone -- nearest, clamp to edge, LOD 0,
    sRGB decode in float64 per texel, factor dropped, RGB padded to RGBA -- on the reference's own lion head and boulder
    (2048 x 2048 JPEG base colour) with its own glTF reader and its brute-force Moeller-Trumbore closest hit.  Both sides
    get the same decoded 8-bit texels (which JPEG decoder made them is a separate third-party boundary).  Bar: primary ids
    exactly (measured), depth to 2e-7, normal to 2e-5, first-bounce throughput (= the texel) to 1e-6 relative, accumulated
    throughput to 1e-6 in >= 97 % of the pixels and 2e-4 everywhere."""
    import glsl_restatement as G
    from helpers import ASSETS, SEED
    from path_tracer_amd import host
    from oracle.gltf_ingest import IngestedScene
    # (brute force over every triangle per ray: sizes chosen for ~40 s of numpy)
    for rel, off, pos, (w, h), spp, depth in (("lion_head_2k/lion_head_2k.gltf", (2, 0, 0), [2.0, 0.22, 0.42], (48, 27), 2, 8),
                                              ("boulder_01_2k/boulder_01_2k.gltf", (-2, 0, 0), [-2.1, 0.45, 1.5], (32, 18), 1, 8)):
        sc = G.load_gltf(os.path.join(ASSETS, rel), off, decode=host.decode_image)
        assert all(m[2] is not None and m[2].shape == (2048, 2048, 4) for m in sc["mats"])
        ing = IngestedScene()
        ing.import_gltf(os.path.join(ASSETS, rel), off, decode_image=host.decode_image)
        assert len(sc["tris"]) * 3 == len(ing.indices)
        osc = oracle.scene_from_ingest(ing)
        vi, pi = G.camera(pos, 0.0, 70.0, w / h, 0.1, 1000.0)
        got = G.render(sc, vi, pi, w, h, spp, SEED, max_depth=depth)
        cam = oracle.camera_matrices(pos, oracle.yaw_quat(0.0), np.deg2rad(np.float32(70)), w / h, 0.1, 1000.0)
        want = osc.render(cam, w, h, spp, SEED, params=oracle.default_params(max_depth=depth))
        assert want["stats"]["textured_hits"] == want["stats"]["hits"] > w * h // 4
        same = np.all(got["ids"] == want["ids"], -1)
        assert same.mean() >= 0.998, (~same).sum()                    # measured: every pixel
        # barycentrics of millimetre triangles from two different triangle tests: 7e-6 on the normal (measured)
        assert np.abs(got["normal"] - want["normal"])[same].max() <= 2e-5 and np.abs(got["depth"] - want["depth"])[same].max() <= 2e-7
        a, b = got["albedo"][..., :3].astype(np.float64), want["albedo"][..., :3].astype(np.float64)
        rel_err = (np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max(-1)
        # measured: 98 % of the pixels within 1e-6, all within 1e-4 (the scatter direction inherits the normal's 7e-6)
        assert (rel_err[same] <= 1e-6).mean() >= 0.97 and rel_err[same].max() <= 2e-4, ((rel_err > 1e-6).mean(), rel_err.max())
        # one bounce only: the throughput IS the texel (or the sky): the sampling rule itself, to the last bit but one
        g1 = G.render(sc, vi, pi, w, h, 1, SEED, max_depth=1)
        w1 = osc.render(cam, w, h, 1, SEED, params=oracle.default_params(max_depth=1))
        a1, b1 = g1["albedo"][..., :3].astype(np.float64), w1["albedo"][..., :3].astype(np.float64)
        assert np.array_equal(g1["ids"], w1["ids"]) and (np.abs(a1 - b1) / np.maximum(np.abs(b1), 1e-3)).max() <= 1e-6
