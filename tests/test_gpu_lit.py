"""GPU parity of the build extensions past the reference: the lit integrator
(emissive triangles, next-event estimation with any-hit occlusion rays, GGX / glass BSDFs, Russian
roulette) and the no-ref-quirks scene options, HIP path through the C ABI vs oracle/pt_oracle.c
(raygen_pixel_lit is the specification).  Bar: bit-exact, ids / G-buffers / RGBA32F radiance."""
import numpy as np
import pytest

from helpers import SEED, assert_images_bit_equal, cornell, cornell_lit, gpu_render, lion_boulder, stress_scene, upload

pytestmark = pytest.mark.gpu


def cam_cornell(oracle, aspect=1.0):
    return oracle.camera_matrices([0, 1.0, 3.9], oracle.yaw_quat(0), np.deg2rad(np.float32(40)), aspect, 0.1, 1000.0)


@pytest.fixture(scope="module")
def lit_pair(oracle, gpu_ctx_factory):
    from path_tracer_amd import capi
    ing = cornell_lit()
    ctx = gpu_ctx_factory()
    upload(ctx, ing, oracle, options=capi.SCENE_WORLD_NORMALS | capi.SCENE_TEXTURE_TIMES_FACTOR, with_material_ext=True)
    sc = oracle.scene_from_ingest(ing, options=oracle.SCENE_WORLD_NORMALS | oracle.SCENE_TEXTURE_TIMES_FACTOR,
                                  with_material_ext=True)
    return sc, ctx


def test_lit_cornell_64_bit_exact(oracle, lit_pair):
    from path_tracer_amd import capi
    sc, ctx = lit_pair
    assert len(sc.lights()) == 2
    cam = cam_cornell(oracle)
    ctx.reset_stats()
    want = sc.render(cam, 64, 64, 8, SEED, params=oracle.default_params(max_depth=8, flags=1, integrator=oracle.INTEGRATOR_LIT))
    got = gpu_render(ctx, oracle, cam, 64, 64, 8, params=capi.default_params(max_depth=8, integrator=capi.INTEGRATOR_LIT))
    assert_images_bit_equal(got, want, "lit cornell 64x64x8")
    st = ctx.stats()
    assert st["lights"] == 2
    assert st["rays"] == want["stats"]["rays"] and st["hits"] == want["stats"]["hits"]
    assert st["shadow_rays"] == want["stats"]["shadow_rays"] > 0
    # the light is visible and the box is lit by it
    assert got["albedo"][..., :3].max() > 5.0 and np.isfinite(got["albedo"]).all()


def test_lit_cornell_384_depth16_bit_exact(oracle, lit_pair):
    from path_tracer_amd import capi
    sc, ctx = lit_pair
    cam = cam_cornell(oracle, 384 / 256)
    want = sc.render(cam, 384, 256, 3, SEED, params=oracle.default_params(integrator=oracle.INTEGRATOR_LIT))
    got = gpu_render(ctx, oracle, cam, 384, 256, 3, params=capi.default_params(integrator=capi.INTEGRATOR_LIT))
    assert_images_bit_equal(got, want, "lit cornell 384x256x3 depth 16")


def test_lit_batch_equals_repeated_and_resumes(oracle, lit_pair):
    from path_tracer_amd import capi
    sc, ctx = lit_pair
    cam = cam_cornell(oracle)
    p = capi.default_params(max_depth=6, integrator=capi.INTEGRATOR_LIT)
    a = gpu_render(ctx, oracle, cam, 80, 48, 5, params=p, batch=True)
    b = gpu_render(ctx, oracle, cam, 80, 48, 5, params=p, batch=False)
    assert_images_bit_equal(a, b, "lit batch vs 5x render")


def test_reference_integrator_unchanged_by_material_ext(oracle, lit_pair, gpu_ctx_factory):
    """Emission / transmission rows and the light table do not leak into the reference integrator."""
    from path_tracer_amd import capi
    sc, ctx = lit_pair
    cam = cam_cornell(oracle)
    want = sc.render(cam, 64, 64, 4, SEED, params=oracle.default_params(max_depth=8))
    got = gpu_render(ctx, oracle, cam, 64, 64, 4, params=capi.default_params(max_depth=8))
    assert_images_bit_equal(got, want, "reference integrator on the lit scene")


def test_world_normals_option_reference_integrator(oracle, gpu_ctx_factory):
    """PTX_SCENE_WORLD_NORMALS: the Cornell box is rotated -90 deg about X (cornell.gltf:26-31), so its
    object-space normals are wrong in the reference; with the option the floor's normal is +Y."""
    from path_tracer_amd import capi
    ing = cornell()
    ctx = gpu_ctx_factory()
    upload(ctx, ing, oracle, options=capi.SCENE_WORLD_NORMALS)
    sc = oracle.scene_from_ingest(ing, options=oracle.SCENE_WORLD_NORMALS)
    cam = cam_cornell(oracle)
    want = sc.render(cam, 64, 64, 4, SEED, params=oracle.default_params(max_depth=8))
    got = gpu_render(ctx, oracle, cam, 64, 64, 4, params=capi.default_params(max_depth=8))
    assert_images_bit_equal(got, want, "cornell, world normals")
    n = got["normal"][60, 32, :3]          # a floor pixel
    assert abs(n[1]) > 0.99, n
    plain = oracle.scene_from_ingest(ing).render(cam, 64, 64, 1, SEED, params=oracle.default_params(max_depth=8))
    assert abs(plain["normal"][60, 32, 1]) < 0.1        # the reference's object-space normal there


def test_textured_lit_and_texture_options(oracle, gpu_ctx_factory):
    """lion + boulder (2k textures, no emitters: sky light only) through the lit integrator, with texel x
    factor and bilinear filtering; and the stress scene's 70 textures with every wrap / filter combination,
    reference integrator."""
    from path_tracer_amd import capi
    opts = capi.SCENE_WORLD_NORMALS | capi.SCENE_TEXTURE_TIMES_FACTOR
    ing = lion_boulder()
    ing.textures = [(rgba, flags | capi.TEXTURE_LINEAR) for rgba, flags in ing.textures]      # 2k textures, bilinear
    rng = np.random.default_rng(3)                 # synthetic metallic-roughness maps on the real uv layouts
    for i in range(len(ing.materials)):
        ing.textures.append((rng.integers(0, 256, (64, 64, 4), dtype=np.uint8), capi.TEXTURE_REPEAT | (capi.TEXTURE_LINEAR if i else 0)))
        ing.material_ext[i]["mr_texture"] = len(ing.textures)
        ing.materials[i]["metallic"] = 1.0
    ctx = gpu_ctx_factory()
    upload(ctx, ing, oracle, options=opts, with_material_ext=True)
    sc = oracle.scene_from_ingest(ing, options=opts, with_material_ext=True)
    cam = oracle.camera_matrices([2.0, 0.25, 0.6], oracle.yaw_quat(0), np.deg2rad(np.float32(70)), 320 / 180, 0.1, 1000.0)
    want = sc.render(cam, 320, 180, 3, SEED, params=oracle.default_params(max_depth=8, integrator=oracle.INTEGRATOR_LIT))
    got = gpu_render(ctx, oracle, cam, 320, 180, 3, params=capi.default_params(max_depth=8, integrator=capi.INTEGRATOR_LIT))
    assert_images_bit_equal(got, want, "lit lion+boulder")
    assert want["stats"]["textured_hits"] > 0 and want["stats"]["shadow_rays"] == 0

    ing = stress_scene()
    # wrap = repeat on the odd ones, bilinear on two of three: all four sampler combinations, sRGB and linear texels
    ing.textures = [(rgba, flags | (2 if i % 2 else 0) | (4 if i % 3 else 0)) for i, (rgba, flags) in enumerate(ing.textures)]
    ctx = gpu_ctx_factory()
    upload(ctx, ing, oracle, options=capi.SCENE_TEXTURE_TIMES_FACTOR)
    sc = oracle.scene_from_ingest(ing, options=oracle.SCENE_TEXTURE_TIMES_FACTOR)
    cam = oracle.camera_matrices([0.0, 0.2, 4.5], oracle.yaw_quat(0), np.deg2rad(np.float32(60)), 1.0, 0.1, 1000.0)
    want = sc.render(cam, 128, 128, 2, SEED, params=oracle.default_params(max_depth=5))
    got = gpu_render(ctx, oracle, cam, 128, 128, 2, params=capi.default_params(max_depth=5))
    assert_images_bit_equal(got, want, "stress scene, repeat + factor")


def test_lit_stress_scene_many_lights(oracle, gpu_ctx_factory):
    """Thousands of emissive triangles (CDF binary search), mixed metallic / rough / glass materials, ties."""
    from path_tracer_amd import capi
    ing = stress_scene(n_confetti=6000)
    rng = np.random.default_rng(11)
    for i in range(len(ing.materials)):
        ing.materials[i]["metallic"] = np.float32([0.0, 1.0, 0.5, 0.0][i % 4])
        if i % 7 == 0:
            ing.material_ext[i]["emissive"] = rng.uniform(0.5, 4.0, 3).astype(np.float32)
        if i % 5 == 1:
            ing.material_ext[i]["transmission"] = np.float32(0.7)
            ing.material_ext[i]["ior"] = np.float32(1.33)
        if i % 3 == 0:
            # a metallic-roughness map from the linear (even-numbered) textures, beyond the 64 descriptors kept in LDS too
            ing.material_ext[i]["mr_texture"] = (2 * i) % 70 + 1
        if i % 2 == 1:
            # tangent-space normal maps (random texels: perturbed normals above and below the surface, confetti with
            # random uvs: degenerate and mirrored tangent frames), again from the linear textures
            ing.material_ext[i]["normal_texture"] = (2 * i + 4) % 70 + 1
        if i % 14 == 0:
            ing.material_ext[i]["emissive_texture"] = (3 * i + 1) % 70 + 1      # emitters with an emissive map (hits and light samples)
    ctx = gpu_ctx_factory()
    upload(ctx, ing, oracle, options=capi.SCENE_WORLD_NORMALS, with_material_ext=True)
    sc = oracle.scene_from_ingest(ing, options=oracle.SCENE_WORLD_NORMALS, with_material_ext=True)
    assert len(sc.lights()) > 500
    cam = oracle.camera_matrices([0.0, 0.2, 4.5], oracle.yaw_quat(0), np.deg2rad(np.float32(60)), 1.0, 0.1, 1000.0)
    want = sc.render(cam, 128, 128, 2, SEED, params=oracle.default_params(max_depth=6, integrator=oracle.INTEGRATOR_LIT))
    got = gpu_render(ctx, oracle, cam, 128, 128, 2, params=capi.default_params(max_depth=6, integrator=capi.INTEGRATOR_LIT))
    assert_images_bit_equal(got, want, "lit stress scene")
    assert ctx.stats()["lights"] == len(sc.lights())


def test_texel_sets_every_mode_against_oracle_and_plain_taps(oracle, gpu_ctx_factory):
    """Texel sets (PTX_OPTION_TEXEL_SETS, include/ptx.h): the materials of the stress scene get normal and metallic-roughness
    maps of their base texture's odd little size (4..6 x 3..7 texels, uv far outside [0, 1]: every edge entry of a set is
    read) in all four wrap / filter combinations -- all three planes, base + normal, normal + metallic-roughness without a
    base texture, base alone (bilinear: a one-load set; nearest: no set), sRGB and linear base texels -- and every fifth a
    metallic-roughness map of ANOTHER size (no set: the plain taps).  Sets on (default) == oracle == sets off, bit for bit,
    with the same ray, hit and occlusion-ray counts; toggling the option on a live context takes effect on the next frame."""
    from path_tracer_amd import capi
    ing = stress_scene(n_textures=40, n_confetti=6000)
    rng = np.random.default_rng(23)
    n_mat = len(ing.materials)
    modes = [(2 if t % 2 else 0) | (4 if t % 3 else 0) for t in range(n_mat)]
    ing.textures = [(rgba, flags | modes[t]) for t, (rgba, flags) in enumerate(ing.textures)]
    for t in range(n_mat):
        h, w = ing.textures[t][0].shape[:2]
        ing.materials[t]["metallic"] = np.float32([0.0, 1.0, 0.5, 0.3][t % 4])
        if t % 9 == 0:
            ing.material_ext[t]["emissive"] = rng.uniform(0.5, 4.0, 3).astype(np.float32)
        if t % 7 == 6:
            continue                                            # base texture alone
        if t % 3 != 1:
            ing.textures.append((rng.integers(0, 256, (h, w, 4), dtype=np.uint8), modes[t]))
            ing.material_ext[t]["normal_texture"] = len(ing.textures)
            if t % 2:
                ing.normal_scale[t] = np.float32(0.35 + 0.3 * (t % 5))      # glTF normalTexture.scale (ptx_set_material_normal_scale)
        if t % 4 != 2:
            shape = (h + 1, w, 4) if t % 5 == 4 else (h, w, 4)   # every fifth: another size, no set
            ing.textures.append((rng.integers(0, 256, shape, dtype=np.uint8), modes[t]))
            ing.material_ext[t]["mr_texture"] = len(ing.textures)
    opts = capi.SCENE_WORLD_NORMALS | capi.SCENE_TEXTURE_TIMES_FACTOR
    sc = oracle.scene_from_ingest(ing, options=opts, with_material_ext=True)
    cam = oracle.camera_matrices([0.0, 0.2, 4.5], oracle.yaw_quat(0), np.deg2rad(np.float32(60)), 1.0, 0.1, 1000.0)
    want = sc.render(cam, 144, 144, 2, SEED, params=oracle.default_params(max_depth=6, integrator=oracle.INTEGRATOR_LIT))
    assert want["stats"]["textured_hits"] > 0 and want["stats"]["shadow_rays"] > 0
    params = capi.default_params(max_depth=6, integrator=capi.INTEGRATOR_LIT)
    a, b = gpu_ctx_factory(), gpu_ctx_factory()
    assert a.get_option(capi.OPTION_TEXEL_SETS) == 1
    b.set_option(capi.OPTION_TEXEL_SETS, 0)
    with pytest.raises(capi.PtxError):
        b.set_option(capi.OPTION_TEXEL_SETS, 2)
    for ctx, label in ((a, "texel sets on"), (b, "texel sets off")):
        upload(ctx, ing, oracle, options=opts, with_material_ext=True)
        ctx.reset_stats()
        got = gpu_render(ctx, oracle, cam, 144, 144, 2, params=params)
        assert_images_bit_equal(got, want, "lit stress scene, " + label)
        st = ctx.stats()
        assert st["rays"] == want["stats"]["rays"] and st["hits"] == want["stats"]["hits"] and st["shadow_rays"] == want["stats"]["shadow_rays"]
    # a really built sets (bilinear 32 / 16 B and nearest 8 B per fetch position of these tiny textures), b none
    assert 2000 < a.get_option(capi.OPTION_TEXEL_SET_BYTES) < 200000 and b.get_option(capi.OPTION_TEXEL_SET_BYTES) == 0
    with pytest.raises(capi.PtxError):
        a.set_option(capi.OPTION_TEXEL_SET_BYTES, 0)
    # toggled on live contexts: a without, b with its sets from now on
    a.set_option(capi.OPTION_TEXEL_SETS, 0)
    b.set_option(capi.OPTION_TEXEL_SETS, 1)
    for ctx, label in ((a, "switched off"), (b, "switched on")):
        assert_images_bit_equal(gpu_render(ctx, oracle, cam, 144, 144, 2, params=params), want, "lit stress scene, texel sets " + label)
    assert b.get_option(capi.OPTION_TEXEL_SET_BYTES) == a.get_option(capi.OPTION_TEXEL_SET_BYTES) > 0
    # the reference integrator never builds or reads a set
    want_ref = sc.render(cam, 96, 96, 1, SEED, params=oracle.default_params(max_depth=4))
    assert_images_bit_equal(gpu_render(b, oracle, cam, 96, 96, 1, params=capi.default_params(max_depth=4)), want_ref, "reference integrator beside texel sets")


def test_lion_real_2k_maps_lit_host_vs_oracle(oracle):
    """The reference's scanned lion head with ITS OWN 2048 x 2048 maps -- base colour (sRGB JPEG), tangent-space normal map,
    occlusion / roughness / metallic map -- through the C++ host's no-ref-quirks glTF import and the lit integrator: one
    bilinear-or-nearest sampler for all three, so the material gets a texel set (real data at full size), against the
    oracle-side ingest of the same file (same decoder), and against the plain taps."""
    import os
    from helpers import ASSETS, ingest
    from path_tracer_amd import capi, host
    W, H, SPP = 288, 162, 2
    fov = np.deg2rad(np.float32(70.0))
    imgs = []
    for sets in (1, 0):
        r = host.Renderer(W, H, 0)
        r._ctx().set_option(capi.OPTION_TEXEL_SETS, sets)
        r.set_ref_quirks(False)
        r.import_gltf(os.path.join(ASSETS, "lion_head_2k/lion_head_2k.gltf"), (2, 0, 0))
        r.set_camera([2.0, 0.25, 0.6], host.yaw_quat(0.0), fov, 0.1, 1000.0)
        r.set_seed(SEED, fixed=True)
        r.set_params(capi.default_params(max_depth=6, integrator=capi.INTEGRATOR_LIT))
        r.record_commands(); r.record_commands()
        r.record_commands_batch(SPP)
        imgs.append({k: r.read_image(v) for k, v in (("albedo", capi.IMAGE_ALBEDO), ("normal", capi.IMAGE_NORMAL),
                                                     ("depth", capi.IMAGE_DEPTH), ("ids", capi.IMAGE_PRIMARY_ID))})
        set_bytes = r._ctx().get_option(capi.OPTION_TEXEL_SET_BYTES)
        assert (set_bytes > 100 << 20) if sets else (set_bytes == 0), set_bytes        # 2050 x 2050 entries x 32 B
        r.close()
    ing = ingest(("lion_head_2k/lion_head_2k.gltf", (2, 0, 0)), decode_image=host.decode_image, ref_quirks=False)
    assert ing.material_ext["normal_texture"][0] and ing.material_ext["mr_texture"][0] and ing.materials["has_base_texture"][0]
    sc = oracle.scene_from_ingest(ing, options=oracle.SCENE_WORLD_NORMALS | oracle.SCENE_TEXTURE_TIMES_FACTOR, with_material_ext=True)
    cam = oracle.camera_matrices([2.0, 0.25, 0.6], oracle.yaw_quat(0), fov, W / H, 0.1, 1000.0)
    want = sc.render(cam, W, H, SPP, SEED, params=oracle.default_params(max_depth=6, integrator=oracle.INTEGRATOR_LIT))
    assert want["stats"]["textured_hits"] > 10000
    assert_images_bit_equal(imgs[0], want, "lion, real 2k maps, texel set")
    assert_images_bit_equal(imgs[1], want, "lion, real 2k maps, plain taps")


def test_extension_error_paths(gpu_ctx_factory):
    from path_tracer_amd import capi
    ctx = gpu_ctx_factory()
    m = np.zeros(1, dtype=capi.MATERIAL_DTYPE)
    mi = ctx.add_material(m)
    ext = np.zeros(1, dtype=capi.MATERIAL_EXT_DTYPE)
    ext["ior"] = 1.5
    ctx.set_material_ext(mi, ext)
    with pytest.raises(capi.PtxError):
        ctx.set_material_ext(mi + 1, ext)
    bad = ext.copy(); bad["ior"] = 0.0
    with pytest.raises(capi.PtxError):
        ctx.set_material_ext(mi, bad)
    bad = ext.copy(); bad["transmission"] = 1.5
    with pytest.raises(capi.PtxError):
        ctx.set_material_ext(mi, bad)
    with pytest.raises(capi.PtxError):
        ctx.set_scene_options(4)
    with pytest.raises(capi.PtxError):
        ctx.add_texture(np.zeros((2, 2, 4), np.uint8), 8)
    ctx.set_material_normal_scale(mi, 0.5)
    with pytest.raises(capi.PtxError):
        ctx.set_material_normal_scale(mi + 1, 0.5)
    with pytest.raises(capi.PtxError):
        ctx.set_material_normal_scale(mi, float("nan"))


def test_host_lit_matches_oracle(oracle):
    """The C++ host mirror with set_ref_quirks(false) + the lit integrator: its glTF import (material
    extensions, sampler wrap, child nodes) feeds the same scene as the oracle-side ingest."""
    import os
    from helpers import ASSETS
    from path_tracer_amd import capi, host
    r = host.Renderer(96, 96, 0)
    r.set_ref_quirks(False)
    r.import_gltf(os.path.join(ASSETS, "cornell_lit", "cornell_lit.gltf"))
    fov = np.deg2rad(np.float32(40.0))
    r.set_camera([0.0, 1.0, 3.9], host.yaw_quat(0.0), fov, 0.1, 1000.0)
    r.set_seed(SEED, fixed=True)
    r.set_params(capi.default_params(max_depth=8, integrator=capi.INTEGRATOR_LIT))
    r.record_commands()          # dirty frame, overwritten (renderer.rs:988-1031)
    r.record_commands()
    r.record_commands_batch(4)
    got = {k: r.read_image(v) for k, v in (("albedo", capi.IMAGE_ALBEDO), ("normal", capi.IMAGE_NORMAL),
                                           ("depth", capi.IMAGE_DEPTH), ("ids", capi.IMAGE_PRIMARY_ID))}
    sc = oracle.scene_from_ingest(cornell_lit(), options=oracle.SCENE_WORLD_NORMALS | oracle.SCENE_TEXTURE_TIMES_FACTOR,
                                  with_material_ext=True)
    want = sc.render(cam_cornell(oracle), 96, 96, 4, SEED, params=oracle.default_params(max_depth=8, integrator=oracle.INTEGRATOR_LIT))
    assert_images_bit_equal(got, want, "host lit cornell")
    r.close()


def test_lit_scene_motion_device_refit_and_moving_emitter(oracle, monkeypatch):
    """Lit scene + instance motion: moving the tall box refits on the GPU (the light table is untouched), moving the
    emitter takes the host path (the light table follows it); both equal the oracle's render of the moved scene."""
    import os
    from helpers import ASSETS
    from path_tracer_amd import capi, host
    monkeypatch.setenv("PTX_REFIT_VERIFY", "1")
    r = host.Renderer(80, 80, 0)
    r.set_ref_quirks(False)
    r.import_gltf(os.path.join(ASSETS, "cornell_lit", "cornell_lit.gltf"))
    fov = np.deg2rad(np.float32(40.0))
    r.set_camera([0.0, 1.0, 3.9], host.yaw_quat(0.0), fov, 0.1, 1000.0)
    r.set_seed(SEED, fixed=True)
    r.set_params(capi.default_params(max_depth=6, integrator=capi.INTEGRATOR_LIT))
    r.record_commands(); r.record_commands()
    ing = cornell_lit()
    cam = cam_cornell(oracle)
    opts = oracle.SCENE_WORLD_NORMALS | oracle.SCENE_TEXTURE_TIMES_FACTOR

    def move(entity, dpos):
        geom, (p, q, s), mat = ing.entities[entity]
        ing.entities[entity] = (geom, (p + np.float32(dpos), q, s), mat)
        r.set_entity_transform(entity, p + np.float32(dpos), q, s)
        r.record_commands()
        r.record_commands_batch(3)
        got = {k: r.read_image(v) for k, v in (("albedo", capi.IMAGE_ALBEDO), ("normal", capi.IMAGE_NORMAL),
                                               ("depth", capi.IMAGE_DEPTH), ("ids", capi.IMAGE_PRIMARY_ID))}
        sc = oracle.scene_from_ingest(ing, options=opts, with_material_ext=True)
        want = sc.render(cam, 80, 80, 3, SEED, params=oracle.default_params(max_depth=6, integrator=oracle.INTEGRATOR_LIT))
        assert_images_bit_equal(got, want, "lit cornell after moving entity %d" % entity)
        return r.stats()

    st = move(7, [0.15, 0.0, 0.2])          # tall box (metal): device refit
    assert st["bvh_refits"] == 1 and st["bvh_device_refits"] == 1 and st["bvh_builds"] == 1
    st = move(4, [0.2, -0.05, 0.1])         # the light quad: host refit, new light table
    assert st["bvh_refits"] == 2 and st["bvh_device_refits"] == 1 and st["bvh_builds"] == 1
    r.close()


def test_host_metallic_roughness_texture(oracle, tmp_path):
    """metallicRoughnessTexture through the C++ host's glTF import (linear texture, its own sampler) against the
    oracle-side ingest of the same file, lit integrator."""
    from helpers import write_mr_quad_gltf
    from oracle.gltf_ingest import IngestedScene
    from path_tracer_amd import capi, host
    path = write_mr_quad_gltf(str(tmp_path))
    r = host.Renderer(96, 96, 0)
    r.set_ref_quirks(False)
    r.import_gltf(path)
    fov = np.deg2rad(np.float32(55.0))
    r.set_camera([0.0, 1.2, 3.5], host.yaw_quat(0.0), fov, 0.1, 1000.0)
    r.set_seed(SEED, fixed=True)
    r.set_params(capi.default_params(max_depth=6, integrator=capi.INTEGRATOR_LIT))
    r.record_commands(); r.record_commands()
    r.record_commands_batch(4)
    got = {k: r.read_image(v) for k, v in (("albedo", capi.IMAGE_ALBEDO), ("normal", capi.IMAGE_NORMAL),
                                           ("depth", capi.IMAGE_DEPTH), ("ids", capi.IMAGE_PRIMARY_ID))}
    ing = IngestedScene(ref_quirks=False)
    ing.import_gltf(path)
    assert ing.material_ext["mr_texture"].tolist() == [2, 0]
    sc = oracle.scene_from_ingest(ing, options=oracle.SCENE_WORLD_NORMALS | oracle.SCENE_TEXTURE_TIMES_FACTOR, with_material_ext=True)
    cam = oracle.camera_matrices([0.0, 1.2, 3.5], oracle.yaw_quat(0), fov, 1.0, 0.1, 1000.0)
    want = sc.render(cam, 96, 96, 4, SEED, params=oracle.default_params(max_depth=6, integrator=oracle.INTEGRATOR_LIT))
    assert_images_bit_equal(got, want, "host mr texture")
    assert want["stats"]["shadow_rays"] > 0 and want["stats"]["textured_hits"] > 0
    r.close()
    ctx = capi.Context(0)
    m = np.zeros(1, dtype=capi.MATERIAL_DTYPE)
    ctx.add_material(m)
    e = np.zeros(1, dtype=capi.MATERIAL_EXT_DTYPE); e["ior"] = 1.5; e["mr_texture"] = 1
    with pytest.raises(capi.PtxError):
        ctx.set_material_ext(0, e)                       # no such texture yet
    ctx.add_texture(np.zeros((2, 2, 4), np.uint8), capi.TEXTURE_SRGB)
    with pytest.raises(capi.PtxError):
        ctx.set_material_ext(0, e)                       # sRGB texture refused
    ctx.close()


def test_material_sort_option_changes_nothing_but_the_order(oracle, monkeypatch):
    """PTX_MATERIAL_SORT=1: between traversal and shading the lit integrator groups the queue
    by material class (sky / opaque / glass) with ballot + block-scan ranks.  Only the order in which rays are shaded
    changes: images, ids and the ray / hit / occlusion-ray counts must equal the oracle's bit for bit -- on the lit
    Cornell box (glass + metal + emitter) and on the stress scene with hundreds of emitters."""
    from path_tracer_amd import capi
    monkeypatch.setenv("PTX_MATERIAL_SORT", "1")
    ing = cornell_lit()
    ctx = capi.Context(0)
    upload(ctx, ing, oracle, options=capi.SCENE_WORLD_NORMALS | capi.SCENE_TEXTURE_TIMES_FACTOR, with_material_ext=True)
    sc = oracle.scene_from_ingest(ing, options=oracle.SCENE_WORLD_NORMALS | oracle.SCENE_TEXTURE_TIMES_FACTOR, with_material_ext=True)
    cam = cam_cornell(oracle, 200 / 136)
    want = sc.render(cam, 200, 136, 5, SEED, params=oracle.default_params(integrator=oracle.INTEGRATOR_LIT))
    ctx.reset_stats()
    got = gpu_render(ctx, oracle, cam, 200, 136, 5, params=capi.default_params(integrator=capi.INTEGRATOR_LIT))
    assert_images_bit_equal(got, want, "lit cornell, material sort on")
    st = ctx.stats()
    assert st["rays"] == want["stats"]["rays"] and st["hits"] == want["stats"]["hits"] and st["shadow_rays"] == want["stats"]["shadow_rays"]
    # the reference integrator ignores the option
    want_ref = sc.render(cam, 200, 136, 2, SEED)
    got_ref = gpu_render(ctx, oracle, cam, 200, 136, 2)
    assert_images_bit_equal(got_ref, want_ref, "reference integrator, material sort env set")
    ctx.close()


def test_host_normal_and_emissive_textures(oracle, tmp_path):
    """normalTexture / emissiveTexture through the C++ host's glTF import against the oracle-side ingest of the same
    file (lit integrator): same bindless indices, bit-identical images; and the C ABI's checks on the new fields."""
    from helpers import write_mr_quad_gltf
    from oracle.gltf_ingest import IngestedScene
    from path_tracer_amd import capi, host
    path = write_mr_quad_gltf(str(tmp_path), with_maps=True)
    r = host.Renderer(96, 96, 0)
    r.set_ref_quirks(False)
    r.import_gltf(path)
    fov = np.deg2rad(np.float32(55.0))
    r.set_camera([0.0, 1.2, 3.5], host.yaw_quat(0.0), fov, 0.1, 1000.0)
    r.set_seed(SEED, fixed=True)
    r.set_params(capi.default_params(max_depth=6, integrator=capi.INTEGRATOR_LIT))
    r.record_commands(); r.record_commands()
    r.record_commands_batch(4)
    got = {k: r.read_image(v) for k, v in (("albedo", capi.IMAGE_ALBEDO), ("normal", capi.IMAGE_NORMAL),
                                           ("depth", capi.IMAGE_DEPTH), ("ids", capi.IMAGE_PRIMARY_ID))}
    ing = IngestedScene(ref_quirks=False)
    ing.import_gltf(path)
    assert ing.material_ext["normal_texture"].tolist() == [3, 0] and ing.material_ext["emissive_texture"].tolist() == [0, 4]
    sc = oracle.scene_from_ingest(ing, options=oracle.SCENE_WORLD_NORMALS | oracle.SCENE_TEXTURE_TIMES_FACTOR, with_material_ext=True)
    cam = oracle.camera_matrices([0.0, 1.2, 3.5], oracle.yaw_quat(0), fov, 1.0, 0.1, 1000.0)
    want = sc.render(cam, 96, 96, 4, SEED, params=oracle.default_params(max_depth=6, integrator=oracle.INTEGRATOR_LIT))
    assert_images_bit_equal(got, want, "host normal + emissive maps")
    floor = want["ids"][..., 0] == 0
    assert np.abs(want["normal"][floor][:, :3] - np.float32([0, 1, 0])).max() > 0.2      # the bumps reach the G-buffer normal
    r.close()
    ctx = capi.Context(0)
    ctx.add_material(np.zeros(1, dtype=capi.MATERIAL_DTYPE))
    for field in ("normal_texture", "emissive_texture"):
        e = np.zeros(1, dtype=capi.MATERIAL_EXT_DTYPE); e["ior"] = 1.5; e[field] = 1
        with pytest.raises(capi.PtxError):
            ctx.set_material_ext(0, e)                   # no such texture yet
    ctx.add_texture(np.zeros((2, 2, 4), np.uint8), capi.TEXTURE_SRGB)
    e = np.zeros(1, dtype=capi.MATERIAL_EXT_DTYPE); e["ior"] = 1.5; e["normal_texture"] = 1
    with pytest.raises(capi.PtxError):
        ctx.set_material_ext(0, e)                       # a normal map must be linear
    e["normal_texture"] = 0; e["emissive_texture"] = 1
    ctx.set_material_ext(0, e)                           # an emissive map may be sRGB
    ctx.close()


def test_lit_fused_tail_equals_bounce_by_bounce(oracle, monkeypatch):
    """The lit integrator's fused tail (k_tail_lit: a lane per path -- closest-hit walk, lit shading, the occlusion query of its
    light sample -- from the first bounce whose expected queue is short) against the oracle and against the wavefront schedule
    (PTX_TAIL_RAYS=0): image, ids, and the ray / hit / occlusion-ray counts, on the lit Cornell box (emitter, glass, metal)
    and on the stress scene with hundreds of emitters, glass, metallic-roughness, normal and emissive maps."""
    from path_tracer_amd import capi
    stress = stress_scene(n_confetti=6000)
    rng = np.random.default_rng(3)
    for i in range(len(stress.materials)):
        stress.materials[i]["metallic"] = np.float32([0.0, 1.0, 0.5, 0.0][i % 4])
        if i % 7 == 0:
            stress.material_ext[i]["emissive"] = rng.uniform(0.5, 4.0, 3).astype(np.float32)
        if i % 5 == 1:
            stress.material_ext[i]["transmission"] = np.float32(0.7)
            stress.material_ext[i]["ior"] = np.float32(1.33)
        if i % 3 == 0:
            stress.material_ext[i]["mr_texture"] = (2 * i) % 70 + 1
        if i % 2 == 1:
            stress.material_ext[i]["normal_texture"] = (2 * i + 4) % 70 + 1
        if i % 14 == 0:
            stress.material_ext[i]["emissive_texture"] = (3 * i + 1) % 70 + 1
    cases = ((cornell_lit(), oracle.SCENE_WORLD_NORMALS | oracle.SCENE_TEXTURE_TIMES_FACTOR, [0, 1.0, 3.9], 40, (160, 120), 12),
             (stress, oracle.SCENE_WORLD_NORMALS, [0.0, 0.2, 4.5], 60, (128, 96), 8))
    for ing, opts, pos, fov, (w, h), depth in cases:
        sc = oracle.scene_from_ingest(ing, options=opts, with_material_ext=True)
        cam = oracle.camera_matrices(pos, oracle.yaw_quat(0), np.deg2rad(np.float32(fov)), w / h, 0.1, 1000.0)
        want = sc.render(cam, w, h, 6, SEED, params=oracle.default_params(max_depth=depth, integrator=oracle.INTEGRATOR_LIT))
        P = capi.default_params(max_depth=depth, integrator=capi.INTEGRATOR_LIT)
        outs = {}
        for tail in (1 << 20, 0):
            monkeypatch.setenv("PTX_TAIL_RAYS", str(tail))
            ctx = capi.Context(0)
            upload(ctx, ing, oracle, options=opts, with_material_ext=True)
            gpu_render(ctx, oracle, cam, w, h, 2, params=P)       # learns the queue lengths
            ctx.synchronize()
            ctx.reset_stats()
            got = gpu_render(ctx, oracle, cam, w, h, 6, params=P, batch=(tail == 0))
            st = ctx.stats()
            assert_images_bit_equal(got, want, "lit tail %d" % tail)
            assert st["rays"] == want["stats"]["rays"] and st["hits"] == want["stats"]["hits"] and st["samples"] == want["stats"]["samples"]
            assert st["shadow_rays"] == want["stats"]["shadow_rays"] > 0
            assert (st["tail_launches"] > 0) == (tail > 0), st["tail_launches"]
            outs[tail] = got
            ctx.close()
        assert_images_bit_equal(outs[1 << 20], outs[0], "lit tail vs bounce by bounce")


def test_lit_many_lights_two_level_cdf(oracle, gpu_ctx_factory):
    """More emissive triangles than k_shade_lit's LDS light table holds (4096): the light pick goes through the table of
    bucket maxima and finishes with a short binary search in the global CDF -- the same index as the oracle's plain search."""
    from path_tracer_amd import capi
    ing = stress_scene(n_confetti=42000)
    rng = np.random.default_rng(5)
    for i in range(len(ing.materials)):
        if i % 2 == 0:
            ing.material_ext[i]["emissive"] = rng.uniform(0.2, 3.0, 3).astype(np.float32)
    ctx = gpu_ctx_factory()
    upload(ctx, ing, oracle, options=capi.SCENE_WORLD_NORMALS, with_material_ext=True)
    sc = oracle.scene_from_ingest(ing, options=oracle.SCENE_WORLD_NORMALS, with_material_ext=True)
    assert len(sc.lights()) > 3 * 4096
    cam = oracle.camera_matrices([0.0, 0.2, 4.5], oracle.yaw_quat(0), np.deg2rad(np.float32(60)), 1.0, 0.1, 1000.0)
    ctx.reset_stats()
    want = sc.render(cam, 96, 96, 3, SEED, params=oracle.default_params(max_depth=5, integrator=oracle.INTEGRATOR_LIT))
    got = gpu_render(ctx, oracle, cam, 96, 96, 3, params=capi.default_params(max_depth=5, integrator=capi.INTEGRATOR_LIT))
    assert_images_bit_equal(got, want, "lit, > 12 k lights")
    st = ctx.stats()
    assert st["lights"] == len(sc.lights()) and st["shadow_rays"] == want["stats"]["shadow_rays"] > 0
