"""CPU tests of the build extensions past the reference on the oracle side:
the lit integrator's specification (oracle/pt_oracle.c raygen_pixel_lit), its light table, the
no-ref-quirks ingest (child nodes, sampler wrap, material extensions) and the scene options."""
import json
import os

import numpy as np

from helpers import SEED, cornell, cornell_lit

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cam(oracle):
    return oracle.camera_matrices([0, 1.0, 3.9], oracle.yaw_quat(0), np.deg2rad(np.float32(40)), 1.0, 0.1, 1000.0)


def _lit_scene(oracle):
    return oracle.scene_from_ingest(cornell_lit(), options=oracle.SCENE_WORLD_NORMALS | oracle.SCENE_TEXTURE_TIMES_FACTOR,
                                    with_material_ext=True)


def test_lit_golden_image(oracle):
    g = np.load(os.path.join(GOLDEN, "cornell_lit_48x48_8spp_depth8.npz"))
    r = _lit_scene(oracle).render(_cam(oracle), 48, 48, 8, SEED,
                                  params=oracle.default_params(max_depth=8, flags=1, integrator=oracle.INTEGRATOR_LIT))
    for k in ("albedo", "normal", "depth", "ids"):
        assert np.array_equal(np.ascontiguousarray(r[k]).view(np.uint32), np.ascontiguousarray(g[k]).view(np.uint32)), k
    assert [r["stats"][k] for k in ("rays", "hits", "samples", "shadow_rays")] == [int(x) for x in g["stats"]]
    # BVH traversal == brute force for the lit integrator too (occlusion queries included)
    r2 = _lit_scene(oracle).render(_cam(oracle), 48, 48, 8, SEED,
                                   params=oracle.default_params(max_depth=8, integrator=oracle.INTEGRATOR_LIT))
    assert np.array_equal(r2["albedo"].view(np.uint32), r["albedo"].view(np.uint32))


def test_lit_textured_golden_image(oracle, tmp_path):
    """The textured lit paths of the oracle (base colour, metallic-roughness, normal map with normalTexture.scale 0.6, emissive
    map) against the committed vector (tests/golden/make_golden_lit.py): a change to raygen_pixel_lit, to the texture
    fetch or to the glTF ingest shows up here before it can move both sides of a GPU parity test together."""
    from helpers import write_mr_quad_gltf
    from oracle.gltf_ingest import IngestedScene
    g = np.load(os.path.join(GOLDEN, "mr_quad_maps_48x48_4spp_depth6.npz"))
    ing = IngestedScene(ref_quirks=False)
    ing.import_gltf(write_mr_quad_gltf(str(tmp_path), with_maps=True))
    assert np.array_equal(ing.normal_scale, g["normal_scale"]) and float(ing.normal_scale[0]) == np.float32(0.6)
    sc = oracle.scene_from_ingest(ing, options=oracle.SCENE_WORLD_NORMALS | oracle.SCENE_TEXTURE_TIMES_FACTOR, with_material_ext=True)
    cam = oracle.camera_matrices([0.0, 1.2, 3.5], oracle.yaw_quat(0), np.deg2rad(np.float32(55.0)), 1.0, 0.1, 1000.0)
    r = sc.render(cam, 48, 48, 4, SEED, params=oracle.default_params(max_depth=6, integrator=oracle.INTEGRATOR_LIT))
    for k in ("albedo", "normal", "depth", "ids"):
        assert np.array_equal(np.ascontiguousarray(r[k]).view(np.uint32), np.ascontiguousarray(g[k]).view(np.uint32)), k
    assert [r["stats"][k] for k in ("rays", "hits", "samples", "shadow_rays", "textured_hits")] == [int(x) for x in g["stats"]]
    # and the scale matters: the same scene with scale 1 gives other normals
    ing.normal_scale[:] = 1.0
    r1 = oracle.scene_from_ingest(ing, options=oracle.SCENE_WORLD_NORMALS | oracle.SCENE_TEXTURE_TIMES_FACTOR, with_material_ext=True).render(
        cam, 48, 48, 1, SEED, params=oracle.default_params(max_depth=2, integrator=oracle.INTEGRATOR_LIT))
    assert not np.array_equal(r1["normal"].view(np.uint32), np.ascontiguousarray(g["normal"]).view(np.uint32))


def test_light_table(oracle):
    sc = _lit_scene(oracle)
    lights = sc.lights()
    assert len(lights) == 2                     # the ceiling quad's two triangles
    cdf = [l[2] for l in lights]
    assert cdf == sorted(cdf) and cdf[-1] == 1.0
    total = 0.0
    for verts, le, _, pdf_area in lights:
        area = 0.5 * np.linalg.norm(np.cross(verts[1] - verts[0], verts[2] - verts[0]))
        total += pdf_area * area                # pick probability
        np.testing.assert_allclose(le, [17.0, 17.0 * 0.85, 17.0 * 0.6], rtol=1e-6)
    assert abs(total - 1.0) < 1e-5
    # no material extensions -> no lights, and the lit integrator degenerates to sky lighting only
    assert len(oracle.scene_from_ingest(cornell()).lights()) == 0


def test_lit_image_properties(oracle):
    sc = _lit_scene(oracle)
    r = sc.render(_cam(oracle), 64, 64, 16, SEED, params=oracle.default_params(max_depth=8, integrator=oracle.INTEGRATOR_LIT))
    a = r["albedo"][..., :3]
    assert np.isfinite(a).all() and (a >= 0).all()
    # a camera ray that hits the emitter returns exactly its radiance (emission is counted on camera paths)
    ids = r["ids"]
    on_light = ids[..., 0] == 4                 # entity 4 = the "light" mesh (cornell.gltf node order)
    assert on_light.sum() > 4
    inner = on_light & np.roll(on_light, 1, 0) & np.roll(on_light, -1, 0) & np.roll(on_light, 1, 1) & np.roll(on_light, -1, 1)
    np.testing.assert_allclose(a[inner], np.broadcast_to(np.float32([17.0, 14.45, 10.2]), a[inner].shape), rtol=1e-5)
    # the floor is lit from above: brighter than with the reference integrator's sky-only transport scaled by albedo
    assert a[40:60, 10:54].mean() > 0.05
    assert r["stats"]["shadow_rays"] > 0
    # the sample partition across GPUs (rng_sample_offset) holds for the lit integrator: 2 x 8 samples == 16
    lo = sc.render(_cam(oracle), 64, 64, 8, SEED, params=oracle.default_params(max_depth=8, integrator=oracle.INTEGRATOR_LIT))
    hi = sc.render(_cam(oracle), 64, 64, 8, SEED,
                   params=oracle.default_params(max_depth=8, integrator=oracle.INTEGRATOR_LIT, rng_sample_offset=8))
    comb = 0.5 * (lo["albedo"].astype(np.float64) + hi["albedo"].astype(np.float64))
    np.testing.assert_allclose(comb[..., :3], a, rtol=2e-5, atol=2e-6)


def test_world_normals_option(oracle):
    cam = _cam(oracle)
    ing = cornell()
    ref = oracle.scene_from_ingest(ing).render(cam, 32, 32, 1, SEED, params=oracle.default_params(max_depth=2))
    wn = oracle.scene_from_ingest(ing, options=oracle.SCENE_WORLD_NORMALS).render(cam, 32, 32, 1, SEED,
                                                                                  params=oracle.default_params(max_depth=2))
    assert np.array_equal(ref["ids"], wn["ids"])
    floor = wn["normal"][30, 16, :3]
    assert abs(floor[1]) > 0.99                               # +-Y in world space
    assert abs(ref["normal"][30, 16, 1]) < 0.1                # the reference's object-space normal (rotated -90 deg about X)
    # unit length either way
    hit = wn["ids"][..., 0] != 0xFFFFFFFF
    np.testing.assert_allclose(np.linalg.norm(wn["normal"][hit][:, :3], axis=1), 1.0, atol=1e-5)


def _write_child_gltf(tmp_path):
    """A parent node (translated, rotated 90 deg about Y, uniformly scaled x2) with a mesh-carrying child."""
    pos = np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]])
    idx = np.uint16([0, 1, 2, 0])               # padded to 8 bytes
    blob = pos.tobytes() + idx.tobytes()
    (tmp_path / "tri.bin").write_bytes(blob)
    h = float(np.sqrt(0.5))
    doc = {
        "asset": {"version": "2.0"},
        "scenes": [{"nodes": [0]}],
        "nodes": [{"name": "parent", "translation": [1, 2, 3], "rotation": [0, h, 0, h], "scale": [2, 2, 2], "children": [1]},
                  {"name": "child", "mesh": 0, "translation": [0.5, 0, 0], "scale": [1, 3, 1]}],
        "meshes": [{"name": "tri", "primitives": [{"attributes": {"POSITION": 0}, "indices": 1, "material": 0}]}],
        "materials": [{"name": "m", "emissiveFactor": [1, 0.5, 0.25],
                       "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 4.0},
                                      "KHR_materials_transmission": {"transmissionFactor": 0.25},
                                      "KHR_materials_ior": {"ior": 1.33}},
                       "pbrMetallicRoughness": {"baseColorFactor": [0.5, 0.5, 0.5, 1], "metallicFactor": 0.0}}],
        "buffers": [{"uri": "tri.bin", "byteLength": len(blob)}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 6}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3", "min": [0, 0, 0], "max": [1, 1, 0]},
                      {"bufferView": 1, "componentType": 5123, "count": 3, "type": "SCALAR"}],
    }
    path = tmp_path / "child.gltf"
    path.write_text(json.dumps(doc))
    return str(path)


def test_no_ref_quirks_ingest_children_and_extensions(oracle, tmp_path):
    from oracle.gltf_ingest import IngestedScene
    path = _write_child_gltf(tmp_path)
    ref = IngestedScene()                        # gltf.rs:32-34: root nodes only -> the child mesh is dropped
    ref.import_gltf(path)
    assert len(ref.entities) == 0
    ing = IngestedScene(ref_quirks=False)
    ing.import_gltf(path)
    assert len(ing.entities) == 1
    (pos, quat, scale) = ing.entities[0][1]
    # parent: p -> R_y(90)(2 p) + (1,2,3); child origin (0.5, 0, 0) -> (1, 2, 3 - 1)
    np.testing.assert_allclose(pos, [1.0, 2.0, 2.0], atol=1e-6)
    np.testing.assert_allclose(scale, [2.0, 6.0, 2.0], atol=1e-6)
    m = oracle.transform_matrix(pos, quat, scale)
    np.testing.assert_allclose(m[:, :3] @ np.float32([1, 0, 0]), [0, 0, -2], atol=1e-5)     # x -> -z, scaled by 2
    np.testing.assert_allclose(ing.material_ext[0]["emissive"], [4.0, 2.0, 1.0])
    assert ing.material_ext[0]["transmission"] == np.float32(0.25) and ing.material_ext[0]["ior"] == np.float32(1.33)
    # the reference-mode ingest parses the same extension rows (they are inert in the reference integrator)
    sc = oracle.scene_from_ingest(ing, with_material_ext=True)
    assert len(sc.lights()) == 1


def test_texture_wrap_repeat_and_factor(oracle):
    """One textured quad seen head-on, uv in [0, 2]: clamp repeats the edge texel, repeat tiles; the factor
    option multiplies the texel by baseColorFactor."""
    from oracle.gltf_ingest import IngestedScene, MATERIAL_DTYPE, VERTEX_DTYPE
    tex = np.zeros((2, 2, 4), np.uint8)
    tex[0, 0] = (255, 0, 0, 255); tex[0, 1] = (0, 255, 0, 255); tex[1, 0] = (0, 0, 255, 255); tex[1, 1] = (255, 255, 255, 255)

    def scene(flags, options):
        ing = IngestedScene()
        ing.textures.append((tex, flags))
        m = np.zeros(1, dtype=MATERIAL_DTYPE)
        m["base_color"] = [0.5, 0.25, 1.0, 1.0]; m["has_base_texture"] = 1; m["roughness"] = 1.0
        mi = ing.add_material(m)
        v = np.zeros(4, dtype=VERTEX_DTYPE)
        v["position"] = [[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]]
        v["normal"] = [[0, 0, 1]] * 4
        v["uv"] = [[0, 2], [2, 2], [2, 0], [0, 0]]
        geom = ing._geometry_create("quad", v, np.uint32([0, 1, 2, 0, 2, 3]))
        ing.entities.append((geom, (np.float32([0, 0, 0]), np.float32([0, 0, 0, 1]), np.float32([1, 1, 1])), mi))
        return oracle.scene_from_ingest(ing, options=options)

    cam = oracle.camera_matrices([0, 0, 2.0], oracle.yaw_quat(0), np.deg2rad(np.float32(60)), 1.0, 0.1, 1000.0)
    # max_depth 1: the image is the first hit's attenuation = albedo (or 0 where the scatter is absorbed)
    def albedo_at(sc, px):
        r = sc.render(cam, 64, 64, 1, SEED, params=oracle.default_params(max_depth=1))
        return r["albedo"][px[1], px[0], :3]
    # a pixel in the quad's upper-right quarter: uv ~ (1.6, 0.4) -> clamp: texel (1, 0) = green; repeat: uv (0.6, 0.4) -> texel (1, 0) green as well
    # a pixel with uv ~ (1.2, 1.2): clamp -> texel (1, 1) white; repeat -> (0.2, 0.2) -> texel (0, 0) red
    px = (38, 38)
    clamp = albedo_at(scene(1, 0), px)
    rep = albedo_at(scene(3, 0), px)
    fac = albedo_at(scene(3, oracle.SCENE_TEXTURE_TIMES_FACTOR), px)
    if clamp.sum() > 0 and rep.sum() > 0 and fac.sum() > 0:          # (absorbed samples are black)
        np.testing.assert_allclose(clamp, [1, 1, 1], atol=1e-6)
        np.testing.assert_allclose(rep, [1, 0, 0], atol=1e-6)
        np.testing.assert_allclose(fac, [0.5, 0, 0], atol=1e-6)
    else:
        raise AssertionError("sample absorbed at the probe pixel; pick another")


def test_texture_bilinear_filter(oracle):
    """A 4-texel ramp across a quad seen head-on: nearest gives steps, bilinear (flag 4) reproduces the ramp
    between texel centres, holds the edge texel under clamp and blends across the seam under repeat."""
    from oracle.gltf_ingest import IngestedScene, MATERIAL_DTYPE, VERTEX_DTYPE
    tex = np.zeros((1, 4, 4), np.uint8)
    tex[0, :, 0] = [0, 85, 170, 255]; tex[0, :, 1] = [255, 170, 85, 0]; tex[0, :, 3] = 255

    def scene(flags):
        ing = IngestedScene()
        ing.textures.append((tex, flags))
        m = np.zeros(1, dtype=MATERIAL_DTYPE)
        m["base_color"] = [1, 1, 1, 1]; m["has_base_texture"] = 1; m["roughness"] = 1.0
        mi = ing.add_material(m)
        v = np.zeros(4, dtype=VERTEX_DTYPE)
        v["position"] = [[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]]
        v["normal"] = [[0, 0, 1]] * 4
        v["uv"] = [[0, 1], [1, 1], [1, 0], [0, 0]]
        geom = ing._geometry_create("quad", v, np.uint32([0, 1, 2, 0, 2, 3]))
        ing.entities.append((geom, (np.float32([0, 0, 0]), np.float32([0, 0, 0, 1]), np.float32([1, 1, 1])), mi))
        return oracle.scene_from_ingest(ing)

    W = 96
    cam = oracle.camera_matrices([0, 0, 2.0], oracle.yaw_quat(0), np.deg2rad(np.float32(60)), 1.0, 0.1, 1000.0)
    px = np.arange(W)
    world_x = ((px + 0.5) / W * 2 - 1) * np.tan(np.deg2rad(30.0)) * 2.0
    u = (world_x + 1) / 2
    inside = (u > 0.01) & (u < 0.99)

    def row(flags):
        r = scene(flags).render(cam, W, W, 1, SEED, params=oracle.default_params(max_depth=1))
        a = r["albedo"][W // 2, :, :3]
        assert (r["ids"][W // 2, inside, 0] == 0).all()
        return a, a.sum(axis=1) > 0            # (an absorbed sample is black)

    x = u * 4 - 0.5
    near, ok = row(0)
    want_near = np.clip(np.floor(u * 4), 0, 3) / 3
    sel = inside & ok
    np.testing.assert_allclose(near[sel, 0], want_near[sel], atol=1e-6)
    lin, ok = row(4)
    sel = inside & ok
    np.testing.assert_allclose(lin[sel, 0], np.clip(x, 0, 3)[sel] / 3, atol=1e-5)
    np.testing.assert_allclose(lin[sel, 1], 1 - np.clip(x, 0, 3)[sel] / 3, atol=1e-5)
    rep, ok = row(4 | 2)
    sel = inside & ok
    xr = np.where(x < 0, x + 4, x)                       # texel -1 is texel 3 under repeat
    i0 = np.floor(xr).astype(int) % 4
    fr = xr - np.floor(xr)
    ramp = np.array([0, 1, 2, 3]) / 3
    want = ramp[i0] * (1 - fr) + ramp[(i0 + 1) % 4] * fr
    np.testing.assert_allclose(rep[sel, 0], want[sel], atol=1e-5)
    assert np.abs(rep[sel, 0] - lin[sel, 0]).max() > 0.3  # the seam differs from the clamped edge
    srgb, ok = row(4 | 1)                                # decoded before the blend
    sel = inside & ok
    lut = np.array([oracle.lib().orc_srgb_to_linear(c) for c in (0, 85, 170, 255)])
    xc = np.clip(x, 0, 3); j = np.minimum(np.floor(xc).astype(int), 2); f = xc - j
    np.testing.assert_allclose(srgb[sel, 0], (lut[j] * (1 - f) + lut[j + 1] * f)[sel], atol=1e-5)


def test_mis_keeps_the_expectation(oracle):
    """Multiple importance sampling re-weights the two ways of reaching an emitter; the image it converges to is
    the one plain next-event estimation converges to (oracle flag bit1), and its noise is lower."""
    ing = cornell_lit()
    sc = oracle.scene_from_ingest(ing, options=oracle.SCENE_WORLD_NORMALS, with_material_ext=True)
    cam = oracle.camera_matrices([0, 1.0, 3.9], oracle.yaw_quat(0), np.deg2rad(np.float32(40)), 1.0, 0.1, 1000.0)
    W, spp = 24, 3072
    imgs = {}
    for name, flags in (("mis", 0), ("nee", 2)):
        halves = [sc.render(cam, W, W, spp, seed, params=oracle.default_params(max_depth=8, integrator=oracle.INTEGRATOR_LIT, flags=flags),
                            want_ids=False)["albedo"][..., :3].astype(np.float64) for seed in (SEED, SEED + 99)]
        imgs[name] = halves
    mis = (imgs["mis"][0] + imgs["mis"][1]) / 2
    nee = (imgs["nee"][0] + imgs["nee"][1]) / 2
    # image-wide energy: equal within the noise of the estimators
    assert abs(mis.mean() - nee.mean()) < 0.02 * nee.mean(), (mis.mean(), nee.mean())
    # per pixel: the difference between the two estimators is no larger than what two independent runs of one of them show
    noise_nee = np.abs(imgs["nee"][0] - imgs["nee"][1]).mean()
    noise_mis = np.abs(imgs["mis"][0] - imgs["mis"][1]).mean()
    assert np.abs(mis - nee).mean() < 1.5 * max(noise_nee, noise_mis), (np.abs(mis - nee).mean(), noise_nee, noise_mis)
    assert noise_mis < noise_nee, (noise_mis, noise_nee)


def test_metallic_roughness_texture_ingest_and_effect(oracle, tmp_path):
    """glTF metallicRoughnessTexture: read only without the reference's quirks, as a linear (non-sRGB) texture with its
    own sampler; optional when the image file is absent; it changes what the lit integrator renders."""
    from helpers import write_mr_quad_gltf
    from oracle.gltf_ingest import IngestedScene
    path = write_mr_quad_gltf(str(tmp_path))
    ref = IngestedScene()
    ref.import_gltf(path)
    assert ref.material_ext["mr_texture"].tolist() == [0, 0] and len(ref.textures) == 1
    ing = IngestedScene(ref_quirks=False)
    ing.import_gltf(path)
    assert ing.material_ext["mr_texture"].tolist() == [2, 0]
    assert [f for _, f in ing.textures] == [1 | 2 | 4, 0]              # base: sRGB, repeat, linear; mr: UNORM, clamp, nearest
    cam = oracle.camera_matrices([0, 1.2, 3.5], oracle.yaw_quat(0), np.deg2rad(np.float32(55)), 1.0, 0.1, 1000.0)
    P = oracle.default_params(max_depth=4, integrator=oracle.INTEGRATOR_LIT)
    opts = oracle.SCENE_WORLD_NORMALS | oracle.SCENE_TEXTURE_TIMES_FACTOR
    with_tex = oracle.scene_from_ingest(ing, options=opts, with_material_ext=True).render(cam, 48, 48, 16, SEED, params=P)["albedo"]
    ing.material_ext["mr_texture"][0] = 0
    without = oracle.scene_from_ingest(ing, options=opts, with_material_ext=True).render(cam, 48, 48, 16, SEED, params=P)["albedo"]
    assert np.isfinite(with_tex).all() and np.abs(with_tex - without).mean() > 1e-3
    os.remove(os.path.join(str(tmp_path), "mr.png"))
    gone = IngestedScene(ref_quirks=False)
    gone.import_gltf(path)
    assert gone.material_ext["mr_texture"].tolist() == [0, 0] and len(gone.textures) == 1


def test_normal_and_emissive_textures_ingest_and_effect(oracle, tmp_path):
    """glTF normalTexture / emissiveTexture (ptx_material_ext.normal_texture / .emissive_texture): read only without the
    reference's quirks -- the normal map as a linear texture, the emissive map as sRGB --, optional when the file is
    absent.  A white emissive map is the same image bit for bit as none (hits on the emitter AND light samples go
    through the texture); a striped one darkens the scene; a normal map changes the shading, not the geometry."""
    from helpers import write_mr_quad_gltf
    from oracle.gltf_ingest import IngestedScene
    from PIL import Image
    path = write_mr_quad_gltf(str(tmp_path), with_maps=True)
    ref = IngestedScene()
    ref.import_gltf(path)
    assert ref.material_ext["normal_texture"].tolist() == [0, 0] and ref.material_ext["emissive_texture"].tolist() == [0, 0]
    ing = IngestedScene(ref_quirks=False)
    ing.import_gltf(path)
    assert ing.material_ext["mr_texture"].tolist() == [2, 0]
    assert ing.material_ext["normal_texture"].tolist() == [3, 0] and ing.material_ext["emissive_texture"].tolist() == [0, 4]
    assert [f for _, f in ing.textures] == [1 | 2 | 4, 0, 2 | 4, 1]     # normal: UNORM with the floor's sampler; emissive: sRGB, clamp, nearest
    cam = oracle.camera_matrices([0, 1.2, 3.5], oracle.yaw_quat(0), np.deg2rad(np.float32(55)), 1.0, 0.1, 1000.0)
    P = oracle.default_params(max_depth=4, integrator=oracle.INTEGRATOR_LIT)
    opts = oracle.SCENE_WORLD_NORMALS | oracle.SCENE_TEXTURE_TIMES_FACTOR

    def render(ingested, spp=16):
        return oracle.scene_from_ingest(ingested, options=opts, with_material_ext=True).render(cam, 48, 48, spp, SEED, params=P)
    full = render(ing)
    assert np.isfinite(full["albedo"]).all()
    # normal map off: same first hits and depth, another shading normal and another image
    ing.material_ext["normal_texture"][0] = 0
    flat = render(ing)
    assert np.array_equal(flat["ids"], full["ids"]) and np.array_equal(flat["depth"], full["depth"])
    floor = full["ids"][..., 0] == 0
    assert np.abs(flat["normal"][floor][:, :3] - full["normal"][floor][:, :3]).max() > 0.2
    assert np.allclose(np.linalg.norm(full["normal"][floor][:, :3], axis=1), 1.0, atol=1e-5)
    assert np.allclose(flat["normal"][floor][:, :3], [0, 1, 0])
    assert np.abs(flat["albedo"] - full["albedo"]).mean() > 1e-3
    # emissive map: white == none, stripes emit less
    striped = flat["albedo"]
    ing.material_ext["emissive_texture"][1] = 0
    none = render(ing)["albedo"]
    ing.textures[3] = (np.full((8, 8, 4), 255, np.uint8), ing.textures[3][1])
    ing.material_ext["emissive_texture"][1] = 4
    white = render(ing)["albedo"]
    assert np.array_equal(white.view(np.uint32), none.view(np.uint32))
    assert striped[..., :3].mean() < 0.95 * none[..., :3].mean()
    for f in ("normal.png", "emit.png"):
        os.remove(os.path.join(str(tmp_path), f))
    gone = IngestedScene(ref_quirks=False)
    gone.import_gltf(path)
    assert gone.material_ext["normal_texture"].tolist() == [0, 0] and gone.material_ext["emissive_texture"].tolist() == [0, 0]
    assert len(gone.textures) == 2


def test_normal_map_tangent_frame_orientation(oracle, tmp_path):
    """The tangent frame of the normal map (no tangents in the vertex format: derived from positions and uvs).  On the test
    floor u runs along +x and glTF's v along +z: a texel that says +X in tangent space must tilt the shading normal towards
    +x (T = dP/du), one that says +Y towards -z (B = -dP/dv: glTF's v runs downwards, the map's +Y is up), and the flat texel
    (128, 128, 255) leaves the normal at +y to within its own quantisation."""
    from helpers import write_mr_quad_gltf
    from oracle.gltf_ingest import IngestedScene
    path = write_mr_quad_gltf(str(tmp_path), with_maps=True)
    ing = IngestedScene(ref_quirks=False)
    ing.import_gltf(path)
    assert ing.material_ext["normal_texture"][0] == 3
    cam = oracle.camera_matrices([0, 1.2, 3.5], oracle.yaw_quat(0), np.deg2rad(np.float32(55)), 1.0, 0.1, 1000.0)
    P = oracle.default_params(max_depth=2, integrator=oracle.INTEGRATOR_LIT)
    opts = oracle.SCENE_WORLD_NORMALS | oracle.SCENE_TEXTURE_TIMES_FACTOR
    for rgb, want in (((255, 128, 128), [1, 0, 0]), ((128, 255, 128), [0, 0, -1]), ((0, 128, 128), [-1, 0, 0]), ((128, 128, 255), [0, 1, 0])):
        ing.textures[2] = (np.tile(np.uint8(list(rgb) + [255]), (4, 4, 1)), ing.textures[2][1])
        r = oracle.scene_from_ingest(ing, options=opts, with_material_ext=True).render(cam, 32, 32, 1, SEED, params=P)
        floor = r["ids"][..., 0] == 0
        n = r["normal"][floor][:, :3]
        assert floor.sum() > 100 and np.abs(n - np.float32(want)).max() < 0.02, (rgb, n[0])
