"""Shared test plumbing: scenes from the oracle-side ingest are fed to BOTH the oracle and the
C ABI, so the two sides see byte-identical inputs."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "assets")
SEED = 0x1234567


def ingest(*items, decode_image=None):
    """items: (gltf path relative to assets/, position offset)"""
    from oracle.gltf_ingest import IngestedScene
    ing = IngestedScene()
    for rel, off in items:
        ing.import_gltf(os.path.join(ASSETS, rel), off, decode_image=decode_image)
    return ing


def cornell():
    return ingest(("cornell/cornell.gltf", (0, 0, 0)))


def lion_boulder():
    # render_window.rs:412-426: lion at x=+2, boulder at x=-2
    return ingest(("lion_head_2k/lion_head_2k.gltf", (2, 0, 0)), ("boulder_01_2k/boulder_01_2k.gltf", (-2, 0, 0)))


def upload(ctx, ing, oracle):
    """Drives the C ABI the way the reference's loaders drive Scene/Renderer."""
    from path_tracer_amd import capi
    for rgba, srgb in ing.textures:
        ctx.add_texture(rgba, srgb)
    for m in ing.materials:
        ctx.add_material(m)
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
