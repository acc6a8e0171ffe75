"""GPU parity at the BASELINE.json configuration sizes on the stand-in geometry, plus the
multi-GPU entry points on one device."""
import os

import numpy as np
import pytest

from helpers import ASSETS, SEED, assert_images_bit_equal, copy_struct, cornell, gpu_render, ingest, upload

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def standins_ready():
    from path_tracer_amd import standins
    standins.ensure_standins(ASSETS)


def _big_limits():
    from path_tracer_amd import capi
    return capi.default_limits(max_vertices=10 ** 7, max_indices=3 * 10 ** 7)


def test_c2_spheres_720p_bit_exact(oracle, gpu_ctx_factory, standins_ready):
    """983 064 vertices: the reference's 2.5 M-vertex capacity holds them (scene/mod.rs:50); roughness 1 / 0 / 0.5 materials."""
    from path_tracer_amd import capi
    ing = ingest(("spheres/spheres.gltf", (0, 0, 0)))
    ctx = gpu_ctx_factory()                      # reference capacities on purpose
    upload(ctx, ing, oracle)
    sc = oracle.scene_from_ingest(ing)
    cam = oracle.camera_matrices([7.0, 2.0, 0.0], oracle.yaw_quat(90), np.deg2rad(np.float32(70)), 1280 / 720, 0.1, 1000.0)
    spp = 24                                     # config: 256 spp; 24 is what the oracle does in a few seconds on the box
    want = sc.render(cam, 1280, 720, spp, SEED)
    got = gpu_render(ctx, oracle, cam, 1280, 720, spp)
    assert_images_bit_equal(got, want, "spheres 1280x720x%d" % spp)
    st = ctx.stats()
    assert st["bvh_triangles"] == 12 + 3 * 491520
    assert st["rays"] == want["stats"]["rays"] and st["hits"] == want["stats"]["hits"]


def test_c3_sponza_1080p_bit_exact_and_c5_4k_primary_ids(oracle, gpu_ctx_factory, standins_ready):
    from path_tracer_amd import capi, host
    ing = ingest(("sponza/Sponza.gltf", (0, 0, 0)), decode_image=host.decode_image)
    ctx = gpu_ctx_factory(limits=_big_limits())
    upload(ctx, ing, oracle)
    sc = oracle.scene_from_ingest(ing)
    cam = oracle.camera_matrices([-8.0, 1.6, 0.0], oracle.yaw_quat(-90), np.deg2rad(np.float32(70)), 16 / 9, 0.1, 1000.0)
    want = sc.render(cam, 1920, 1080, 1, SEED)
    got = gpu_render(ctx, oracle, cam, 1920, 1080, 1)
    assert_images_bit_equal(got, want, "sponza 1920x1080x1")
    assert want["stats"]["textured_hits"] == want["stats"]["hits"] > 0
    # C5: 3840x2160 -- the oracle needs minutes for the full frame, so:
    #  * 2 spp in one batch == 2 single renders (chunking at 8.3 M pixels), and
    #  * three row bands of the frame -- the top 16 rows, 64 rows through the middle of the atrium, the bottom 16 rows --
    #    equal the oracle's render of the same rows bit for bit (a pixel depends on nothing but its own launch id,
    #    rgen.glsl:20-141, so orc_render_rows of a band IS the full render's band).
    a = gpu_render(ctx, oracle, cam, 3840, 2160, 2, batch=True)
    b = gpu_render(ctx, oracle, cam, 3840, 2160, 2, batch=False)
    assert_images_bit_equal(a, b, "4K batch vs repeated")
    assert np.all(a["albedo"][..., 3] == 1.0) and np.isfinite(a["albedo"]).all()
    st = ctx.stats()
    assert st["samples"] >= 2 * 2 * 3840 * 2160
    for y0, y1 in ((0, 16), (1048, 1112), (2144, 2160)):
        band = sc.render(cam, 3840, 2160, 2, SEED, rows=(y0, y1))
        assert band["stats"]["samples"] == 2 * 3840 * (y1 - y0)
        assert_images_bit_equal({k: a[k][y0:y1] for k in ("ids", "normal", "depth", "albedo")},
                                {k: band[k][y0:y1] for k in ("ids", "normal", "depth", "albedo")}, "4K rows %d..%d vs oracle" % (y0, y1))
    mid = sc.render(cam, 3840, 2160, 1, SEED, rows=(1048, 1112))
    assert (mid["ids"][1048:1112, :, 0] != 0xFFFFFFFF).mean() > 0.5          # the band really looks at geometry


def test_c3_sponza_1080p_lit_nee_bit_exact(oracle, gpu_ctx_factory, standins_ready):
    """BASELINE.json config 3 names next-event estimation: the lit integrator on the Sponza stand-in at 1920x1080 --
    bench.py's `--integrator lit` workload (the emitters of bench.lit_emitters, the scene's real metallic-roughness and
    normal maps, world normals, texel x factor, sampler wrap / filter modes) -- one sample per pixel through the C ABI
    against oracle/pt_oracle.c raygen_pixel_lit: ids, normal, depth and RGBA32F radiance bitwise, and the ray, hit and
    occlusion-ray counts."""
    import bench
    from path_tracer_amd import capi, host
    ing = ingest(("sponza/Sponza.gltf", (0, 0, 0)), decode_image=host.decode_image, ref_quirks=False)
    n_emit = 0
    for row, ext in bench.lit_emitters("sponza_1080p", ing.materials):
        ing.material_ext[row] = ext[0]
        n_emit += 1
    assert n_emit > 0
    assert any(int(e["mr_texture"]) for e in ing.material_ext) and any(int(e["normal_texture"]) for e in ing.material_ext)
    opts = capi.SCENE_WORLD_NORMALS | capi.SCENE_TEXTURE_TIMES_FACTOR
    ctx = gpu_ctx_factory(limits=_big_limits())
    upload(ctx, ing, oracle, options=opts, with_material_ext=True)
    sc = oracle.scene_from_ingest(ing, options=oracle.SCENE_WORLD_NORMALS | oracle.SCENE_TEXTURE_TIMES_FACTOR, with_material_ext=True)
    cam = oracle.camera_matrices([-8.0, 1.6, 0.0], oracle.yaw_quat(-90), np.deg2rad(np.float32(70)), 16 / 9, 0.1, 1000.0)
    want = sc.render(cam, 1920, 1080, 1, SEED, params=oracle.default_params(integrator=oracle.INTEGRATOR_LIT))
    ctx.reset_stats()
    got = gpu_render(ctx, oracle, cam, 1920, 1080, 1, params=capi.default_params(integrator=capi.INTEGRATOR_LIT))
    assert_images_bit_equal(got, want, "lit sponza 1920x1080x1")
    st = ctx.stats()
    assert st["lights"] == len(sc.lights()) > 1000
    assert st["samples"] == want["stats"]["samples"] == 1920 * 1080
    assert st["rays"] == want["stats"]["rays"] and st["hits"] == want["stats"]["hits"]
    assert st["shadow_rays"] == want["stats"]["shadow_rays"] > 1920 * 1080 // 2
    assert np.isfinite(got["albedo"]).all() and got["albedo"][..., :3].max() > 1.0


def test_accumulation_resume_and_device_export(oracle, gpu_ctx_factory):
    """Dump / reload of the accumulation buffer (SURVEY.md 5 checkpoint row) and the device-pointer export used by the
    one-process-per-GPU reduce; ptx_reduce_accumulation over a single context is the identity."""
    import torch
    from path_tracer_amd import capi
    from helpers import cornell
    ing = cornell()
    ctx = gpu_ctx_factory()
    upload(ctx, ing, oracle)
    cam = oracle.camera_matrices([0, 1.0, 3.9], oracle.yaw_quat(0), np.deg2rad(np.float32(40)), 1.0, 0.1, 1000.0)
    full = gpu_render(ctx, oracle, cam, 128, 128, 8)["albedo"]
    half = gpu_render(ctx, oracle, cam, 128, 128, 5)["albedo"]
    # resume: new images, reload the 5-sample mean, continue with samples 5..7
    ctx.resize(128, 128)
    ctx.write_albedo(half)
    u = capi.Uniforms(1, SEED, 5, 0.1, 100.0, (0, 0, 0))
    ctx.render_batch(u, copy_struct(capi.Camera, cam), 3)
    resumed = ctx.read_image(capi.IMAGE_ALBEDO)
    assert np.array_equal(resumed.view(np.uint32), full.view(np.uint32))
    # device export (scaled) / import round trip through a torch tensor on the same GPU
    t = torch.empty((128, 128, 4), dtype=torch.float32, device="cuda:0")
    ctx.export_albedo_device(t.data_ptr(), t.numel() * 4, 0.5)
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy()[..., :3], full[..., :3] * np.float32(0.5))
    ctx.import_albedo_device(t.data_ptr(), t.numel() * 4, 2.0)
    assert np.array_equal(ctx.read_image(capi.IMAGE_ALBEDO).view(np.uint32), full.view(np.uint32))
    capi.reduce_accumulation([ctx], root=0)
    assert np.array_equal(ctx.read_image(capi.IMAGE_ALBEDO).view(np.uint32), full.view(np.uint32))


def test_error_paths(oracle, gpu_ctx_factory):
    from path_tracer_amd import capi
    from helpers import cornell
    ctx = gpu_ctx_factory(limits=capi.default_limits(max_vertices=10, max_indices=12, max_materials=1, max_textures=1))
    cam = copy_struct(capi.Camera, oracle.camera_matrices([0, 1, 3.9], oracle.yaw_quat(0), 0.7, 1.0, 0.1, 1000.0))
    u = capi.Uniforms(1, 1, 0, 0.1, 100.0, (0, 0, 0))
    with pytest.raises(capi.PtxError) as e:
        ctx.render(u, cam)
    assert e.value.code == capi.PTX_ERR_STATE                        # no instances yet
    ing = cornell()
    with pytest.raises(capi.PtxError) as e:
        ctx.add_vertices(ing.vertices)
    assert e.value.code == capi.PTX_ERR_CAPACITY                     # scene/mod.rs:281-283 panics here
    ctx.add_vertices(ing.vertices[:4]); ctx.add_indices(ing.indices[:6])
    ctx.add_material(ing.materials[0])
    with pytest.raises(capi.PtxError) as e:
        ctx.add_material(ing.materials[1])
    assert e.value.code == capi.PTX_ERR_CAPACITY
    with pytest.raises(capi.PtxError) as e:
        ctx.create_geometry(0, 3, 0, 6)                              # index 3 refers past 3 vertices
    assert e.value.code == capi.PTX_ERR_INVALID
    g = ctx.create_geometry(0, 4, 0, 6)
    inst = np.zeros(1, capi.INSTANCE_DTYPE)
    inst[0]["xform"] = np.eye(4, dtype=np.float32)[:3]
    inst[0]["geometry"] = g
    inst[0]["off"] = (0, 4, 7, 0, 0, 6)                              # material 7 does not exist
    with pytest.raises(capi.PtxError) as e:
        ctx.set_instances(inst)
    assert e.value.code == capi.PTX_ERR_INVALID
    inst[0]["off"] = (0, 4, 0, 0, 0, 6)
    ctx.set_instances(inst)
    with pytest.raises(capi.PtxError) as e:
        ctx.render(u, cam)
    assert e.value.code == capi.PTX_ERR_STATE                        # no image yet
    ctx.resize(16, 16)
    ctx.render(u, cam)
    # the traversal orders hits by the bits of t and tests boxes in t_min-shifted time: 0 <= t_min < t_max or nothing
    for bad in (dict(t_min=-0.5), dict(t_min=float("nan")), dict(t_min=5.0, t_max=5.0), dict(t_max=float("nan")), dict(t_max=-1.0)):
        for call in (ctx.render, ctx.render_static):
            with pytest.raises(capi.PtxError) as e:
                call(u, cam, capi.default_params(**bad))
            assert e.value.code == capi.PTX_ERR_INVALID, bad
    ctx.render(u, cam, capi.default_params(t_min=0.0))                # zero is allowed
    with pytest.raises(capi.PtxError) as e:
        ctx._check(ctx._L.ptx_read_image(ctx.handle, capi.IMAGE_ALBEDO, np.zeros(4, np.float32).ctypes.data, 16))
    assert e.value.code == capi.PTX_ERR_INVALID
    # empty scene renders the sky
    ctx.set_instances(inst[:0])
    ctx.resize(16, 16)
    ctx.render(u, cam)
    img = ctx.read_image(capi.IMAGE_ALBEDO)
    assert np.all(ctx.read_image(capi.IMAGE_PRIMARY_ID) == 0xFFFFFFFF) and img[0, 0, 2] > 0.5


def test_frames_in_flight_equal_serial_frames(oracle, monkeypatch):
    """consecutive ptx_render calls overlap on two streams (two path sets, event-chained resolves).
    The images must equal those of strictly serial frames (PTX_FRAMES_IN_FLIGHT=1) bit for bit, through camera
    moves with and without an accumulation restart, batches after single frames, and a resize."""
    from helpers import copy_struct
    from path_tracer_amd import capi
    ing = cornell()

    def run(frames_in_flight):
        monkeypatch.setenv("PTX_FRAMES_IN_FLIGHT", str(frames_in_flight))
        ctx = capi.Context(0)
        upload(ctx, ing, oracle)
        ctx.resize(160, 120)
        cams = [copy_struct(capi.Camera, oracle.camera_matrices(p, oracle.yaw_quat(y), np.deg2rad(np.float32(40)), 160 / 120, 0.1, 1000.0))
                for p, y in (([0, 1.0, 3.9], 0.0), ([0.4, 1.1, 3.5], 8.0), ([-0.5, 0.8, 3.0], -12.0))]
        p = capi.default_params(max_depth=8)
        snaps = []

        def frame(cam, acc, n=1):
            u = capi.Uniforms(1, SEED, acc, 0.1, 100.0, (0, 0, 0))
            if n == 1:
                ctx.render(u, cam, p)
            else:
                ctx.render_batch(u, cam, n, p)
        for acc in range(5):
            frame(cams[0], acc)                     # clean accumulating frames: these overlap
        frame(cams[1], 0)                           # camera moved, accumulation restarts: waits for everything before it
        frame(cams[1], 1); frame(cams[1], 2)
        snaps.append({k: ctx.read_image(v) for k, v in (("albedo", capi.IMAGE_ALBEDO), ("normal", capi.IMAGE_NORMAL),
                                                        ("depth", capi.IMAGE_DEPTH), ("ids", capi.IMAGE_PRIMARY_ID))})
        frame(cams[2], 3)                           # a caller that moves the camera WITHOUT restarting: still ordered
        frame(cams[2], 4, n=3)
        frame(cams[2], 7)
        snaps.append({k: ctx.read_image(v) for k, v in (("albedo", capi.IMAGE_ALBEDO), ("normal", capi.IMAGE_NORMAL),
                                                        ("depth", capi.IMAGE_DEPTH), ("ids", capi.IMAGE_PRIMARY_ID))})
        ctx.resize(96, 64)
        for acc in range(4):
            frame(cams[0], acc)
        snaps.append({k: ctx.read_image(v) for k, v in (("albedo", capi.IMAGE_ALBEDO), ("normal", capi.IMAGE_NORMAL),
                                                        ("depth", capi.IMAGE_DEPTH), ("ids", capi.IMAGE_PRIMARY_ID))})
        ctx.close()
        return snaps

    serial, two, four = run(1), run(2), run(4)
    for i, (a, b, c) in enumerate(zip(serial, two, four)):
        assert_images_bit_equal(b, a, "2 frames in flight vs serial, snapshot %d" % i)
        assert_images_bit_equal(c, a, "4 frames in flight vs serial, snapshot %d" % i)
    # and the serial schedule equals the oracle for the first camera sequence's restart
    sc = oracle.scene_from_ingest(ing)
    cam1 = oracle.camera_matrices([0.4, 1.1, 3.5], oracle.yaw_quat(8.0), np.deg2rad(np.float32(40)), 160 / 120, 0.1, 1000.0)
    want = sc.render(cam1, 160, 120, 3, SEED, params=oracle.default_params(max_depth=8))
    assert_images_bit_equal(serial[0], want, "serial frames vs oracle")


def test_options_are_per_context_and_change_no_bit(oracle, gpu_ctx_factory):
    """ptx_set_option: two contexts of one process with different scheduling knobs (frames in flight, chunk size, fused
    tail, traversal grid, shading grid) give the same bits; values read back; bad keys / values are refused."""
    from path_tracer_amd import capi
    ing = cornell()
    cam = oracle.camera_matrices([0, 1.0, 3.9], oracle.yaw_quat(0), np.deg2rad(np.float32(40)), 160 / 120, 0.1, 1000.0)
    a, b = gpu_ctx_factory(), gpu_ctx_factory()
    assert a.get_option(capi.OPTION_FRAMES_IN_FLIGHT) == 2 and a.get_option(capi.OPTION_TAIL_RAYS) == 65536
    assert a.get_option(capi.OPTION_TARGET_PATHS) == 64 << 20 and a.get_option(capi.OPTION_SHADE_GRID) == 8192
    for key, value in ((capi.OPTION_FRAMES_IN_FLIGHT, 4), (capi.OPTION_TARGET_PATHS, 160 * 120 * 3), (capi.OPTION_TAIL_RAYS, 0),
                       (capi.OPTION_TAIL_SHIFT, 3), (capi.OPTION_EXTEND_WAVES, 640), (capi.OPTION_SHADE_GRID, 64)):
        b.set_option(key, value)
        assert b.get_option(key) == value
        assert a.get_option(key) != value                             # the other context keeps its own
    b.set_option(capi.OPTION_EXTEND_WAVES, 0)
    assert b.get_option(capi.OPTION_EXTEND_WAVES) == a.get_option(capi.OPTION_EXTEND_WAVES)   # 0 = what the device holds
    b.set_option(capi.OPTION_EXTEND_WAVES, 640)
    for key, value in ((capi.OPTION_FRAMES_IN_FLIGHT, 0), (capi.OPTION_FRAMES_IN_FLIGHT, 5), (capi.OPTION_TAIL_SHIFT, 32),
                       (capi.OPTION_TARGET_PATHS, 100), (capi.OPTION_MATERIAL_SORT, 2), (0, 1), (99, 1)):
        with pytest.raises(capi.PtxError) as e:
            b.set_option(key, value)
        assert e.value.code == capi.PTX_ERR_INVALID
    with pytest.raises(capi.PtxError):
        b.get_option(99)
    outs = []
    for ctx in (a, b):
        upload(ctx, ing, oracle)
        gpu_render(ctx, oracle, cam, 160, 120, 2, params=capi.default_params(max_depth=8))      # a learns its queue lengths (fused tail)
        outs.append(gpu_render(ctx, oracle, cam, 160, 120, 7, params=capi.default_params(max_depth=8)))
    assert_images_bit_equal(outs[1], outs[0], "context with other options")
    assert a.stats()["tail_launches"] > 0 and b.stats()["tail_launches"] == 0
    want = oracle.scene_from_ingest(ing).render(cam, 160, 120, 7, SEED, params=oracle.default_params(max_depth=8))
    assert_images_bit_equal(outs[1], want, "options vs oracle")


def test_c5_converged_4096spp_rel_mse(oracle, gpu_ctx_factory, standins_ready):
    """BASELINE.json: converged images must match the reference's render within rel-MSE <= 1e-3 at 4096 spp.  The
    'reference' is the CPU oracle (the reference cannot run here); its 4096-spp render of the Sponza stand-in at
    192x108 is the committed fixture tests/golden/sponza_standin_192x108_4096spp.npz (make_golden_converged.py).
    Tolerance: rel-MSE = mean((a-b)^2 / (b^2 + 1e-2)) <= 1e-3 -- and because the build pins every fp32 expression
    and applies samples in accumulated_count order, the single-GPU image is in fact bit-identical.  The 8-way sample
    partition (mean of 8 means of 512 samples, the multi-GPU combine) differs only by fp32 summation order."""
    from helpers import rel_mse
    from path_tracer_amd import capi, host
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sponza_standin_192x108_4096spp.npz"))
    want = g["albedo"]
    W, H, SPP = 192, 108, int(g["spp"])
    ing = ingest(("sponza/Sponza.gltf", (0, 0, 0)), decode_image=host.decode_image)
    ctx = gpu_ctx_factory(limits=_big_limits())
    upload(ctx, ing, oracle)
    cam = oracle.camera_matrices([-8.0, 1.6, 0.0], oracle.yaw_quat(-90.0), np.deg2rad(np.float32(70.0)), W / H, 0.1, 1000.0)
    got = gpu_render(ctx, oracle, cam, W, H, SPP)
    err = rel_mse(got["albedo"], want)
    assert err <= 1e-3, err
    assert np.array_equal(got["albedo"][..., :3].view(np.uint32), want.view(np.uint32)), "expected bit-identical, rel-MSE %g" % err
    st = ctx.stats()
    # 8-way spp partition on one device: slice g renders samples [512 g, 512 g + 512) into its own running mean
    acc = np.zeros((H, W, 4), np.float64)
    for part in range(8):
        p = capi.default_params(rng_sample_offset=part * (SPP // 8))
        acc += gpu_render(ctx, oracle, cam, W, H, SPP // 8, params=p)["albedo"]
    combined = (acc / 8.0).astype(np.float32)
    err8 = rel_mse(combined, want)
    assert err8 <= 1e-3 and err8 < 1e-9, err8
    assert st["samples"] >= W * H * SPP


def test_odd_extents_depth_extremes_and_bad_triangles(oracle, gpu_ctx_factory):
    """Edges of the input space: 1 x 1 and other extents that are no multiple of a wave or a block, max_depth 1 and 32,
    one sample; and a scene that carries degenerate, non-finite and duplicated triangles (they hit nothing / tie)."""
    from path_tracer_amd import capi
    from oracle.gltf_ingest import IngestedScene, MATERIAL_DTYPE, VERTEX_DTYPE, calculate_normals
    ing = cornell()
    ctx = gpu_ctx_factory()
    upload(ctx, ing, oracle)
    sc = oracle.scene_from_ingest(ing)
    for (w, h), depth, spp in (((1, 1), 16, 3), ((3, 17), 1, 2), ((65, 33), 32, 2), ((257, 1), 4, 1)):
        cam = oracle.camera_matrices([0, 1.0, 3.9], oracle.yaw_quat(7.0), np.deg2rad(np.float32(40)), w / h, 0.1, 1000.0)
        want = sc.render(cam, w, h, spp, SEED, params=oracle.default_params(max_depth=depth))
        got = gpu_render(ctx, oracle, cam, w, h, spp, params=capi.default_params(max_depth=depth))
        assert_images_bit_equal(got, want, "cornell %dx%d depth %d" % (w, h, depth))

    bad = IngestedScene()
    m = np.zeros(1, dtype=MATERIAL_DTYPE)
    m["base_color"] = [0.8, 0.6, 0.4, 1.0]; m["roughness"] = 0.7
    mi = bad.add_material(m)
    p = np.float32([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0],          # a good quad
                    [0, 0, 1], [0, 0, 1], [0, 0, 1],                          # a point
                    [-1, 0, 0.5], [0, 0, 0.5], [1, 0, 0.5],                   # collinear
                    [np.nan, 0, 0.2], [1, 0, 0.2], [0, 1, 0.2],               # NaN vertex
                    [np.inf, 0, 0.3], [1, 0, 0.3], [0, 1, 0.3],               # infinite vertex
                    [-0.5, -0.5, 0.25], [0.5, -0.5, 0.25], [0, 0.5, 0.25]])   # a good triangle in front, listed twice below
    idx = np.uint32([0, 1, 2, 0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 16, 17, 18])
    v = np.zeros(len(p), dtype=VERTEX_DTYPE)
    v["position"] = p
    v["normal"] = [0, 0, 1]
    geom = bad._geometry_create("bad", v, idx)
    bad.entities.append((geom, (np.float32([0, 0, 0]), np.float32([0, 0, 0, 1]), np.float32([1, 1, 1])), mi))
    ctx = gpu_ctx_factory()
    upload(ctx, bad, oracle)
    sc = oracle.scene_from_ingest(bad)
    cam = oracle.camera_matrices([0.1, 0.05, 3.0], oracle.yaw_quat(0), np.deg2rad(np.float32(50)), 1.0, 0.1, 1000.0)
    want = sc.render(cam, 96, 96, 4, SEED, params=oracle.default_params(max_depth=6))
    got = gpu_render(ctx, oracle, cam, 96, 96, 4, params=capi.default_params(max_depth=6))
    assert_images_bit_equal(got, want, "bad triangles")
    prims = set(np.unique(want["ids"][..., 1]).tolist())
    assert prims == {0, 1, 6, 0xFFFFFFFF}                 # the quad, the first copy of the front triangle, the sky


def test_fused_tail_launch_equals_bounce_by_bounce(oracle, monkeypatch):
    """From the second chunk on a context knows how short the late queues are and runs the last bounces as one fused
    launch (a lane per path: scalar traversal + the shading of k_shade).  Image, ids and the ray / hit counts must equal
    the oracle's and the bounce-by-bounce schedule's (PTX_TAIL_RAYS=0), for single frames, batches and a textured scene."""
    from helpers import lion_boulder
    from path_tracer_amd import capi
    for ing, pos, fov, size in ((cornell(), [0, 1.0, 3.9], 40, (160, 120)), (lion_boulder(), [-1.5, 1.0, 5.0], 70, (192, 108))):
        sc = oracle.scene_from_ingest(ing)
        w, h = size
        cam = oracle.camera_matrices(pos, oracle.yaw_quat(0), np.deg2rad(np.float32(fov)), w / h, 0.1, 1000.0)
        want = sc.render(cam, w, h, 6, SEED, params=oracle.default_params(max_depth=16))
        outs = {}
        for tail in (16384, 0):
            monkeypatch.setenv("PTX_TAIL_RAYS", str(tail))
            ctx = capi.Context(0)
            upload(ctx, ing, oracle)
            gpu_render(ctx, oracle, cam, w, h, 2, params=capi.default_params(max_depth=16))       # learns the queue lengths
            ctx.synchronize()
            ctx.reset_stats()
            got = gpu_render(ctx, oracle, cam, w, h, 6, params=capi.default_params(max_depth=16), batch=(tail == 0))
            st = ctx.stats()
            assert_images_bit_equal(got, want, "tail %d" % tail)
            assert st["rays"] == want["stats"]["rays"] and st["hits"] == want["stats"]["hits"] and st["samples"] == want["stats"]["samples"]
            assert (st["tail_launches"] > 0) == (tail > 0), st["tail_launches"]
            outs[tail] = got
            ctx.close()
        assert_images_bit_equal(outs[16384], outs[0], "tail vs bounce by bounce")


def test_reduce_accumulation_two_devices(oracle):
    """ptx_reduce_accumulation over two contexts on two GPUs (RCCL, cached communicator): GPU g renders sample
    slice g; the reduced image on the root must equal one context rendering all the samples up to fp32 summation
    order.  Called twice (the second call reuses the communicator) and the caller's device is left as it was.
    Skipped on a one-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from helpers import rel_mse
    from path_tracer_amd import capi
    ing = cornell()
    cam = oracle.camera_matrices([0, 1.0, 3.9], oracle.yaw_quat(0), np.deg2rad(np.float32(40)), 1.0, 0.1, 1000.0)
    ctxs = [capi.Context(d) for d in (0, 1)]
    for c in ctxs:
        upload(c, ing, oracle)
    torch.cuda.set_device(0)
    single = gpu_render(ctxs[0], oracle, cam, 128, 128, 32, params=capi.default_params(max_depth=8))["albedo"]
    for round_ in range(2):
        parts = [gpu_render(c, oracle, cam, 128, 128, 16, params=capi.default_params(max_depth=8, rng_sample_offset=16 * g))["albedo"]
                 for g, c in enumerate(ctxs)]
        capi.reduce_accumulation(ctxs, root=0)
        assert torch.cuda.current_device() == 0
        got = ctxs[0].read_image(capi.IMAGE_ALBEDO)
        host_mean = ((parts[0].astype(np.float64) + parts[1].astype(np.float64)) / 2.0).astype(np.float32)
        assert np.max(np.abs(got[..., :3] - host_mean[..., :3])) <= 2e-7, round_
        assert np.all(got[..., 3] == 1.0)
        assert rel_mse(got, single) < 1e-9
    for c in ctxs:
        c.close()


def test_bench_two_rank_rehearsal_reduce_leg():
    """bench.py's N > 1 path with the real renderer: `python bench.py --gpus 2` as typed (the script launches its own
    ranks), both ranks on this box's one GPU, the accumulation reduce host-staged over gloo (RCCL refuses two ranks on
    one device).  Exercises export -> reduce -> import and checks the combined image against one context rendering
    every rank's samples."""
    import json
    import subprocess
    import sys
    from helpers import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-gloo", "--workload", "cornell_512",
                        "--steps", "2", "--warmup", "1", "--spp-per-step", "2", "--max-depth", "8", "--cpu-baseline-spp", "0",
                        "--verify-reduce"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "weak"
    assert d["reduce_check"]["samples_per_rank"] == 4 and d["reduce_check"]["ranks"] == 2    # 2 dirty frames, then 4 accumulated
    assert d["reduce_check"]["rel_mse_vs_single_context"] < 1e-9, d["reduce_check"]
    assert d["value"] > 0 and d["steps"] == 2
    # the line taken apart: per-rank render time, the reduce leg, per-rank load time
    assert len(d["render_ms"]["per_rank"]) == 2 and d["render_ms"]["max"] >= d["render_ms"]["min"] > 0
    assert d["reduce_ms"] > 0 and len(d["scene_load_s_per_rank"]) == 2 and d["spp_per_rank"] == [4, 4]
    assert d["time_to_image_ms"] >= d["render_ms"]["max"]


def test_bench_one_rank_through_rccl():
    """bench.py --gpus 1 --force-dist: the process group of the N > 1 path brought up over RCCL for ONE rank, so that the
    collectives the multi-GPU line depends on -- barrier, the reduce of the 1080p-sized accumulation tensor (the untimed warm-up
    call), the gathers of the per-rank timings, the max / sum all-reduces -- really execute in librccl on this box's GPU
    (two ranks on one device is what RCCL refuses, one rank it takes).  The line must equal a plain N = 1 run's shape."""
    import json
    import subprocess
    import sys
    from helpers import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--workload", "cornell_512",
                        "--steps", "2", "--warmup", "1", "--spp-per-step", "2", "--max-depth", "8", "--cpu-baseline-spp", "0",
                        "--verify-reduce"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["collective_backend"] == "nccl (RCCL)"
    assert d["reduce_check"]["rel_mse_vs_single_context"] < 1e-12, d["reduce_check"]
    assert len(d["render_ms"]["per_rank"]) == 1 and d["reduce_ms"] >= 0 and d["value"] > 0


def test_bench_fixed_budget_mode_two_ranks_and_one(tmp_path):
    """bench.py --total-spp S (BASELINE.json configs 3 / 5: a fixed sample budget partitioned over the GPUs): S = 10 spp
    split over two ranks (5 each, as batches of 3 + 2), time-to-image including the reduce, "scaling": "strong"; the
    combined image against one context rendering both slices.  Then the same budget at N = 1 through the same code.
    Also the stale-profile guard: counter summaries stamped with another source hash are not quoted."""
    import json
    import subprocess
    import sys
    from helpers import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    stale = tmp_path / "traffic.json"
    stale.write_text(json.dumps({"workload": "cornell_512", "csrc_sha16": "0000000000000000", "hbm_bytes_per_ray": 1.0, "tcc_hit_rate": 0.5}))
    (tmp_path / "valu.json").write_text(json.dumps({"workload": "cornell_512", "csrc_sha16": "0000000000000000", "kernels": {}}))
    lines = {}
    for gpus in (2, 1):
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--workload", "cornell_512", "--total-spp", "10",
               "--warmup", "1", "--spp-per-step", "3", "--max-depth", "8", "--cpu-baseline-spp", "0", "--traffic-json", str(stale)]
        if gpus > 1:
            cmd += ["--rehearse-gloo", "--verify-reduce"]
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        out = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(out) == 1, p.stdout
        lines[gpus] = json.loads(out[0])
    two, one = lines[2], lines[1]
    assert two["scaling"] == one["scaling"] == "strong" and two["total_spp"] == one["total_spp"] == 10
    assert two["spp_per_rank"] == [5, 5] and two["steps"] == 2 and one["spp_per_rank"] == [10] and one["steps"] == 4
    # the job's samples: the budget x pixels, whatever the number of ranks
    for d in (two, one):
        assert abs(d["value"] * d["time_to_image_ms"] * 1e-3 / (10 * 512 * 512) - 1.0) < 1e-3, (d["value"], d["time_to_image_ms"])   # (both rounded in the line)
        assert d["roofline"]["profile_stale"] is True and d["roofline"]["traffic"] is None and d["roofline"]["valu_issue"] is None
    assert two["reduce_ms"] > 0 and one["reduce_ms"] == 0.0
    assert two["reduce_check"]["samples_per_rank"] == 5 and two["reduce_check"]["rel_mse_vs_single_context"] < 1e-9, two["reduce_check"]
