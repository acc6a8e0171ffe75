"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on identical inputs.
Bar: bit-exact for the primary-hit (instance, primitive) ids AND, because the build pins every
fp32 expression (no FMA contraction, shared sin/cos polynomial), for the RGBA32F accumulation,
normal and depth images as well."""
import numpy as np
import pytest

from helpers import SEED, assert_images_bit_equal, cornell, gpu_render, lion_boulder, rel_mse, stress_scene, upload

pytestmark = pytest.mark.gpu


def cornell_cam(oracle, aspect=1.0):
    # SURVEY.md 8d, config C1
    return oracle.camera_matrices([0, 1.0, 3.9], oracle.yaw_quat(0), np.deg2rad(np.float32(40)), aspect, 0.1, 1000.0)


@pytest.fixture(scope="module")
def cornell_pair(oracle, gpu_ctx_factory):
    ing = cornell()
    ctx = gpu_ctx_factory()
    upload(ctx, ing, oracle)
    return oracle.scene_from_ingest(ing), ctx


def test_cornell_64x64_16spp_bit_exact(oracle, cornell_pair):
    sc, ctx = cornell_pair
    cam = cornell_cam(oracle)
    want = sc.render(cam, 64, 64, 16, SEED, params=oracle.default_params(max_depth=8, flags=1))
    from path_tracer_amd import capi
    got = gpu_render(ctx, oracle, cam, 64, 64, 16, params=capi.default_params(max_depth=8))
    assert_images_bit_equal(got, want, "cornell 64x64x16")
    st = ctx.stats()
    assert st["samples"] >= 64 * 64 * 16


def test_render_batch_equals_repeated_render(oracle, cornell_pair):
    sc, ctx = cornell_pair
    cam = cornell_cam(oracle)
    a = gpu_render(ctx, oracle, cam, 96, 64, 5, batch=True)
    b = gpu_render(ctx, oracle, cam, 96, 64, 5, batch=False)
    assert_images_bit_equal(a, b, "batch vs 5x render")


def test_c1_cornell_512_64spp_depth8_bit_exact(oracle, cornell_pair):
    """BASELINE.json config 1 at its real size: 512 x 512, 64 spp, max depth 8, fixed seed (16.8 M pixel-samples;
    the oracle takes a few seconds on the box's host cores)."""
    sc, ctx = cornell_pair
    cam = cornell_cam(oracle)
    from path_tracer_amd import capi
    want = sc.render(cam, 512, 512, 64, SEED, params=oracle.default_params(max_depth=8))
    ctx.synchronize()
    ctx.reset_stats()
    got = gpu_render(ctx, oracle, cam, 512, 512, 64, params=capi.default_params(max_depth=8))
    assert_images_bit_equal(got, want, "cornell 512x512x64")
    st = ctx.stats()
    assert st["samples"] == 512 * 512 * 64 == want["stats"]["samples"]
    assert st["rays"] == want["stats"]["rays"] and st["hits"] == want["stats"]["hits"]


def test_lion_boulder_textured_bit_exact(oracle, gpu_ctx_factory):
    ing = lion_boulder()
    ctx = gpu_ctx_factory()
    upload(ctx, ing, oracle)
    sc = oracle.scene_from_ingest(ing)
    # SURVEY.md 8d C4 close-up on the lion (the app's default camera at render_window.rs:436 sees both as specks)
    cam = oracle.camera_matrices([2.0, 0.25, 0.6], oracle.yaw_quat(0), np.deg2rad(np.float32(70)), 320 / 180, 0.1, 1000.0)
    want = sc.render(cam, 320, 180, 4, SEED)
    got = gpu_render(ctx, oracle, cam, 320, 180, 4)
    assert_images_bit_equal(got, want, "lion+boulder 320x180x4")
    assert want["stats"]["textured_hits"] > 0
    assert rel_mse(got["albedo"], want["albedo"]) == 0.0


def test_c4_lion_boulder_1080p_both_cameras_bit_exact(oracle, gpu_ctx_factory):
    """BASELINE.json config 4 at its real size: the scanned lion head (x = +2) and boulder (x = -2) placed as
    render_window.rs:412-426 does, 1920 x 1080, from both SURVEY.md 8d cameras -- the app's default view
    (render_window.rs:436) and the close-up that fills the frame with the lion (BVH depth / divergence stress).
    ids, normal, depth and RGBA32F accumulation bitwise against the oracle; ray / hit counts equal."""
    ing = lion_boulder()
    ctx = gpu_ctx_factory()
    upload(ctx, ing, oracle)
    sc = oracle.scene_from_ingest(ing)
    for pos, spp in (([-1.5, 1.0, 5.0], 4), ([2.0, 0.25, 0.6], 4)):
        cam = oracle.camera_matrices(pos, oracle.yaw_quat(0), np.deg2rad(np.float32(70)), 1920 / 1080, 0.1, 1000.0)
        want = sc.render(cam, 1920, 1080, spp, SEED)
        ctx.synchronize()
        ctx.reset_stats()
        got = gpu_render(ctx, oracle, cam, 1920, 1080, spp)
        assert_images_bit_equal(got, want, "lion+boulder 1920x1080x%d from %s" % (spp, pos))
        st = ctx.stats()
        assert st["rays"] == want["stats"]["rays"] and st["hits"] == want["stats"]["hits"]
        assert want["stats"]["textured_hits"] == want["stats"]["hits"] > 0
    assert ctx.stats()["bvh_triangles"] == 47214 + 66122


def test_stress_scene_ties_queue_overflow_many_textures(oracle, gpu_ctx_factory):
    """Coincident stacks (equal-t ties), leaf-heavy nodes (the per-wave triangle queue overflows and the
    lanes retry), 70 textures (descriptors beyond the 64 staged in LDS), linear + sRGB texels, clamped uv."""
    ing = stress_scene()
    ctx = gpu_ctx_factory()
    upload(ctx, ing, oracle)
    sc = oracle.scene_from_ingest(ing)
    from path_tracer_amd import capi
    for pos, yaw in (([0.0, 0.2, 4.5], 0.0), ([0.3, -0.4, -1.0], 180.0)):
        cam = oracle.camera_matrices(pos, oracle.yaw_quat(yaw), np.deg2rad(np.float32(60)), 1.0, 0.1, 1000.0)
        want = sc.render(cam, 160, 160, 3, SEED, params=oracle.default_params(max_depth=6))
        got = gpu_render(ctx, oracle, cam, 160, 160, 3, params=capi.default_params(max_depth=6))
        assert_images_bit_equal(got, want, "stress scene from %s" % (pos,))
    assert want["stats"]["textured_hits"] > 0


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_random_soups_bit_exact(oracle, gpu_ctx_factory, seed):
    """Randomised inputs: triangle soups of random size and spread (slivers, big overlapping triangles, clusters far from the
    origin), random vertex normals and uv, random materials (roughness 0 .. 1, some textured with tiny random sRGB / linear
    textures), instances with random TRS, random cameras inside and outside the cloud, depths 1 .. 24.  Everything the
    reference path reads is exercised with values no asset has; the HIP path must still equal the oracle bit for bit."""
    from path_tracer_amd import capi
    from oracle.gltf_ingest import IngestedScene, MATERIAL_DTYPE, VERTEX_DTYPE
    rng = np.random.default_rng(seed)
    ing = IngestedScene()
    n_tex = int(rng.integers(0, 5))
    for t in range(n_tex):
        ing.textures.append((rng.integers(0, 256, size=(int(rng.integers(1, 9)), int(rng.integers(1, 9)), 4), dtype=np.uint8), int(rng.integers(0, 2))))
    mats = []
    for t in range(int(rng.integers(1, 7))):
        m = np.zeros(1, dtype=MATERIAL_DTYPE)
        m["base_color"] = rng.random(4, dtype=np.float32)
        m["has_base_texture"] = 1 if (n_tex and rng.random() < 0.5) else 0
        m["base_texture_indice"] = int(rng.integers(0, max(1, n_tex)))
        m["metallic"] = rng.random()
        m["roughness"] = np.float32(rng.choice([0.0, 1.0, rng.random()]))
        mats.append(ing.add_material(m))
    n_geom = int(rng.integers(1, 4))
    for g in range(n_geom):
        n_tri = int(rng.integers(1, 3000))
        centre = rng.normal(0, rng.choice([0.5, 3.0, 40.0]), size=(n_tri, 1, 3))
        size = rng.choice([0.01, 0.3, 2.0, 8.0], size=(n_tri, 1, 1), p=[0.2, 0.5, 0.25, 0.05])
        pos = (centre + rng.normal(0, 1, size=(n_tri, 3, 3)) * size).reshape(-1, 3).astype(np.float32)
        v = np.zeros(len(pos), dtype=VERTEX_DTYPE)
        v["position"] = pos
        nrm = rng.normal(size=(len(pos), 3)).astype(np.float32)
        v["normal"] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
        v["uv"] = rng.uniform(-2, 3, size=(len(pos), 2)).astype(np.float32)
        idx = np.arange(len(pos), dtype=np.uint32)
        if rng.random() < 0.5:
            idx = rng.permutation(idx.reshape(-1, 3)).reshape(-1)             # shuffled primitive order
        geom = ing._geometry_create("soup%d" % g, v, idx)
        for k in range(int(rng.integers(1, 4))):                              # instances with random TRS
            q = rng.normal(size=4); q /= np.linalg.norm(q)
            ing.entities.append((geom, (np.float32(rng.normal(0, 2.0, 3)), np.float32(q), np.float32(rng.uniform(0.3, 2.5, 3))),
                                 mats[int(rng.integers(0, len(mats)))]))
    ctx = gpu_ctx_factory()
    upload(ctx, ing, oracle)
    sc = oracle.scene_from_ingest(ing)
    for k in range(3):
        w, h = int(rng.integers(17, 120)), int(rng.integers(9, 90))
        pos = rng.normal(0, rng.choice([1.0, 10.0, 60.0]), 3).tolist()
        cam = oracle.camera_matrices(pos, oracle.yaw_quat(float(rng.uniform(-180, 180))), np.deg2rad(np.float32(rng.uniform(20, 100))), w / h, 0.1, 1000.0)
        depth, spp = int(rng.integers(1, 25)), int(rng.integers(1, 5))
        want = sc.render(cam, w, h, spp, SEED + seed, params=oracle.default_params(max_depth=depth))
        ctx.synchronize(); ctx.reset_stats()
        got = gpu_render(ctx, oracle, cam, w, h, spp, seed=SEED + seed, params=capi.default_params(max_depth=depth))
        assert_images_bit_equal(got, want, "random soup %d view %d (%dx%d depth %d spp %d)" % (seed, k, w, h, depth, spp))
        st = ctx.stats()
        assert st["rays"] == want["stats"]["rays"] and st["hits"] == want["stats"]["hits"]


@pytest.mark.gpu
@pytest.mark.parametrize("width", [2, 3, 5])
def test_closest_hits_do_not_depend_on_the_tree(oracle, gpu_ctx_factory, monkeypatch, width):
    """The image is a property of the scene, not of the acceleration structure: with the builder collapsing to nodes of at
    most 2 / 3 / 5 children (PTX_LAB_BVH_WIDTH, a developer switch; deeper trees, other groupings, other stack depths) the
    textured lion + boulder render must still equal the oracle bit for bit."""
    from path_tracer_amd import capi
    monkeypatch.setenv("PTX_LAB_BVH_WIDTH", str(width))
    ing = lion_boulder()
    ctx = gpu_ctx_factory()
    upload(ctx, ing, oracle)
    sc = oracle.scene_from_ingest(ing)
    cam = oracle.camera_matrices([2.0, 0.25, 0.6], oracle.yaw_quat(0), np.deg2rad(np.float32(70)), 16 / 9, 0.1, 1000.0)
    want = sc.render(cam, 160, 90, 2, SEED, params=oracle.default_params(max_depth=6))
    got = gpu_render(ctx, oracle, cam, 160, 90, 2, params=capi.default_params(max_depth=6))
    assert_images_bit_equal(got, want, "tree width %d" % width)
    assert ctx.stats()["bvh_nodes"] > (40000 if width == 2 else 15000)
