"""ptx_render_static -- the entry point that allocates nothing (include/ptx.h): its frames must equal ptx_render's
bit for bit.  Two runs in a row on one GPU, the second on a smaller image (the shared file-scope scratch keeps the
first run's contents: stale queue lengths, fetch heads or hit records would show up here), then the limits.
Collected last (test_zz_*): a failure here must not hide the other GPU tests under `pytest -x`."""
import numpy as np
import pytest
import torch

from helpers import SEED, assert_images_bit_equal, copy_struct, cornell, lion_boulder, upload

pytestmark = pytest.mark.gpu

KINDS = ("albedo", "normal", "depth", "ids")


def _images(ctx):
    from path_tracer_amd import capi
    return dict(albedo=ctx.read_image(capi.IMAGE_ALBEDO), normal=ctx.read_image(capi.IMAGE_NORMAL),
                depth=ctx.read_image(capi.IMAGE_DEPTH), ids=ctx.read_image(capi.IMAGE_PRIMARY_ID))


def _frames(ctx, cam, w, h, n, params, static):
    from path_tracer_amd import capi
    ctx.resize(w, h)
    for s in range(n):
        u = capi.Uniforms(1, SEED, s, 0.1, 100.0, (0, 0, 0))
        (ctx.render_static if static else ctx.render)(u, cam, params)
    return _images(ctx)


def test_render_static_equals_render_twice_second_smaller(oracle, gpu_ctx_factory):
    from path_tracer_amd import capi
    ing = lion_boulder()
    a, b = gpu_ctx_factory(), gpu_ctx_factory()
    upload(a, ing, oracle)
    upload(b, ing, oracle)
    params = capi.default_params(max_depth=16)
    # one frame first: the HIP runtime sizes the queue's private-segment memory on a stream's first launch of a
    # kernel that needs it -- its business, not an allocation of the entry point; everything after is measured
    b.resize(64, 64)
    b.render_static(capi.Uniforms(1, SEED, 0, 0.1, 100.0, (0, 0, 0)),
                    copy_struct(capi.Camera, oracle.camera_matrices([2.0, 0.25, 0.6], oracle.yaw_quat(0), np.deg2rad(np.float32(70)), 1.0, 0.1, 1000.0)), params)
    b.synchronize()
    # first call: 448 x 320 close-up of the lion (textured, deep paths); second: 200 x 120 from further away
    for (w, h), pos, n in (((448, 320), [2.0, 0.25, 0.6], 3), ((200, 120), [1.2, 0.6, 2.2], 4)):
        cam = copy_struct(capi.Camera, oracle.camera_matrices(pos, oracle.yaw_quat(0), np.deg2rad(np.float32(70)), w / h, 0.1, 1000.0))
        want = _frames(a, cam, w, h, n, params, static=False)
        b.resize(w, h)
        b.synchronize()
        free_before = torch.cuda.mem_get_info(0)[0]
        got = _frames_static_only(b, cam, n, params)
        free_after = torch.cuda.mem_get_info(0)[0]
        assert_images_bit_equal(got, want, "render_static %dx%d" % (w, h))
        assert free_after >= free_before, "device memory in use grew by %d bytes across ptx_render_static calls" % (free_before - free_after)
    # mixed with the allocating path on ONE context: frames 0-1 static, 2-3 ptx_render, 4 static == five ptx_render frames
    w, h = 160, 120
    cam = copy_struct(capi.Camera, oracle.camera_matrices([2.0, 0.25, 0.6], oracle.yaw_quat(0), np.deg2rad(np.float32(70)), w / h, 0.1, 1000.0))
    want = _frames(a, cam, w, h, 5, params, static=False)
    b.resize(w, h)
    for s, static in enumerate((True, True, False, False, True)):
        u = capi.Uniforms(1, SEED, s, 0.1, 100.0, (0, 0, 0))
        (b.render_static if static else b.render)(u, cam, params)
    assert_images_bit_equal(_images(b), want, "static and allocating frames interleaved")
    sa, sb = a.stats(), b.stats()
    assert sb["samples"] > 0 and sb["rays"] > sb["samples"]


def _frames_static_only(ctx, cam, n, params):
    from path_tracer_amd import capi
    for s in range(n):
        u = capi.Uniforms(1, SEED, s, 0.1, 100.0, (0, 0, 0))
        ctx.render_static(u, cam, params)
    ctx.synchronize()
    return _images(ctx)


def test_render_static_matches_oracle_and_rejects_what_it_cannot_hold(oracle, gpu_ctx_factory):
    from path_tracer_amd import capi
    ing = cornell()
    ctx = gpu_ctx_factory()
    upload(ctx, ing, oracle)
    sc = oracle.scene_from_ingest(ing)
    ocam = oracle.camera_matrices([0, 1.0, 3.9], oracle.yaw_quat(0), np.deg2rad(np.float32(40)), 1.0, 0.1, 1000.0)
    cam = copy_struct(capi.Camera, ocam)
    ctx.resize(512, 512)                                # the largest image the static scratch holds
    got = _frames_static_only(ctx, cam, 2, capi.default_params(max_depth=8))
    want = sc.render(ocam, 512, 512, 2, SEED, params=oracle.default_params(max_depth=8))
    assert_images_bit_equal(got, want, "render_static cornell 512x512x2 vs oracle")
    u = capi.Uniforms(1, SEED, 0, 0.1, 100.0, (0, 0, 0))
    ctx.resize(513, 512)
    with pytest.raises(capi.PtxError) as e:
        ctx.render_static(u, cam)
    assert e.value.code == capi.PTX_ERR_CAPACITY and "262144" in str(e.value)
    ctx.resize(64, 64)
    with pytest.raises(capi.PtxError) as e:
        ctx.render_static(u, cam, capi.default_params(integrator=capi.INTEGRATOR_LIT))
    assert e.value.code == capi.PTX_ERR_INVALID
    ctx.render_static(u, cam)                           # and still works afterwards
    ctx.synchronize()


def test_render_static_two_contexts_back_to_back_without_synchronize(oracle, gpu_ctx_factory):
    """Two contexts of one GPU share that GPU's scratch array and run on different streams.  Their static frames, issued
    alternately from this one thread with NO ptx_synchronize in between, are ordered on the device by the per-device
    event of ptx_render_static; each context's image must equal its own ptx_render frames."""
    from path_tracer_amd import capi
    ing = lion_boulder()
    a, b, ref = gpu_ctx_factory(), gpu_ctx_factory(), gpu_ctx_factory()
    for c in (a, b, ref):
        upload(c, ing, oracle)
    params = capi.default_params(max_depth=16)
    views = (((320, 240), [2.0, 0.25, 0.6]), ((256, 160), [1.2, 0.6, 2.2]))
    cams = [copy_struct(capi.Camera, oracle.camera_matrices(pos, oracle.yaw_quat(0), np.deg2rad(np.float32(70)), w / h, 0.1, 1000.0))
            for (w, h), pos in views]
    want = [_frames(ref, cam, w, h, 6, params, static=False) for ((w, h), _), cam in zip(views, cams)]
    for c, ((w, h), _) in zip((a, b), views):
        c.resize(w, h)
    for s in range(6):
        u = capi.Uniforms(1, SEED, s, 0.1, 100.0, (0, 0, 0))
        a.render_static(u, cams[0], params)
        b.render_static(u, cams[1], params)
    assert_images_bit_equal(_images(a), want[0], "context a, interleaved static frames")
    assert_images_bit_equal(_images(b), want[1], "context b, interleaved static frames")
