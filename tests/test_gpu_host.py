"""GPU tests of the C++ host mirror (Renderer / Scene::import_gltf / codecs) against the oracle-side ingest."""
import os

import numpy as np
import pytest

from helpers import ASSETS, SEED, assert_images_bit_equal, ingest

pytestmark = pytest.mark.gpu


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


def test_import_gltf_matches_oracle_ingest(oracle):
    from path_tracer_amd import capi, host
    r = host.Renderer(64, 64, 0)
    r.import_gltf(os.path.join(ASSETS, "lion_head_2k/lion_head_2k.gltf"), (2, 0, 0))
    r.import_gltf(os.path.join(ASSETS, "boulder_01_2k/boulder_01_2k.gltf"), (-2, 0, 0))
    r.import_gltf(os.path.join(ASSETS, "cornell/cornell.gltf"))
    v, i, m, inst = r.scene_arrays()
    ing = ingest(("lion_head_2k/lion_head_2k.gltf", (2, 0, 0)), ("boulder_01_2k/boulder_01_2k.gltf", (-2, 0, 0)),
                 ("cornell/cornell.gltf", (0, 0, 0)), decode_image=host.decode_image)
    assert v.tobytes() == ing.vertices.tobytes()
    assert i.tobytes() == ing.indices.tobytes()
    assert m.tobytes() == ing.materials.tobytes()
    want = ing.instances(oracle.transform_matrix)
    assert np.array_equal(inst["xform"].view(np.uint32), want["xform"].view(np.uint32))
    assert inst["off"].tobytes() == want["off"].tobytes()
    r.close()


def test_renderer_frames_match_oracle_and_dirty_rules(oracle, tmp_path):
    """record_commands bookkeeping (renderer.rs:988-1031): a dirty frame renders with accumulated_count 0 and
    does not advance the counter; clean frames advance it; moving the camera restarts accumulation."""
    from path_tracer_amd import capi, host
    r = host.Renderer(96, 64, 0)
    r.import_gltf(os.path.join(ASSETS, "cornell/cornell.gltf"))
    fov = np.deg2rad(np.float32(40.0))
    r.set_camera([0.0, 1.0, 3.9], host.yaw_quat(0.0), fov)
    r.set_seed(SEED, fixed=True)
    r.record_commands()                  # camera-dirty; the || short-circuit leaves the scene flag set
    assert r.accumulated_count == 0
    r.record_commands()                  # scene-dirty
    assert r.accumulated_count == 0
    r.record_commands()
    assert r.accumulated_count == 1
    r.record_commands_batch(3)
    assert r.accumulated_count == 4
    ing = ingest(("cornell/cornell.gltf", (0, 0, 0)))
    sc = oracle.scene_from_ingest(ing)
    cam = oracle.camera_matrices([0.0, 1.0, 3.9], oracle.yaw_quat(0.0), fov, 96 / 64, 0.1, 1000.0)
    want = sc.render(cam, 96, 64, 4, SEED)
    got = dict(albedo=r.read_image(capi.IMAGE_ALBEDO), normal=r.read_image(capi.IMAGE_NORMAL),
               depth=r.read_image(capi.IMAGE_DEPTH), ids=r.read_image(capi.IMAGE_PRIMARY_ID))
    assert_images_bit_equal(got, want, "renderer frames")
    # camera move -> dirty -> restart
    r.set_camera([0.2, 1.0, 3.9], host.yaw_quat(5.0), fov)
    r.record_commands()
    assert r.accumulated_count == 0
    r.record_commands()
    assert r.accumulated_count == 1
    cam2 = oracle.camera_matrices([0.2, 1.0, 3.9], oracle.yaw_quat(5.0), fov, 96 / 64, 0.1, 1000.0)
    want2 = sc.render(cam2, 96, 64, 1, SEED)
    got2 = dict(albedo=r.read_image(capi.IMAGE_ALBEDO), normal=r.read_image(capi.IMAGE_NORMAL),
                depth=r.read_image(capi.IMAGE_DEPTH), ids=r.read_image(capi.IMAGE_PRIMARY_ID))
    assert_images_bit_equal(got2, want2, "after camera move")
    # accumulation limit: the counter stops advancing
    r.set_accumulation_limit(2)
    r.record_commands(); r.record_commands(); r.record_commands()
    assert r.accumulated_count == 2
    # save: linear clamp x255, refuses to overwrite (renderer.rs:752-754)
    out = str(tmp_path / "frame.png")
    r.save_render_texture_to_file(out)
    from PIL import Image
    png = np.asarray(Image.open(out))
    alb = r.read_image(capi.IMAGE_ALBEDO)
    assert np.array_equal(png, np.floor(np.clip(alb[..., :3], 0, 1) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8))
    with pytest.raises(host.HostError, match="File already exists"):
        r.save_render_texture_to_file(out)
    r.close()


def read_pfm(path):
    with open(path, "rb") as f:
        kind = f.readline().strip()
        w, h = map(int, f.readline().split())
        scale = float(f.readline())
        assert scale < 0                                          # little endian
        ch = 3 if kind == b"PF" else 1
        a = np.frombuffer(f.read(), dtype="<f4").reshape(h, w, ch)
    return a[::-1]                                                # PFM stores the bottom row first


def test_headless_cli_matches_the_host_renderer(tmp_path):
    """ptx-render writes the same accumulation / normal / depth images (PFM) and the same PNG as the
    in-process host mirror driven with the same arguments."""
    import subprocess
    from path_tracer_amd import build, capi, host
    cli = build.build_cli()
    gltf = os.path.join(ASSETS, "cornell/cornell.gltf")
    out = str(tmp_path / "cli")
    r = subprocess.run([cli, gltf, "--res", "96x64", "--spp", "5", "--seed", "19088743", "--depth", "8",
                        "--camera", "0,1.0,3.9", "--yaw", "0", "--fov", "40", "--out", out],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    import json
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["samples"] == 96 * 64 * 7 and rep["rays"] > rep["samples"]      # 2 settle frames + 5 accumulated

    R = host.Renderer(96, 64)
    R.import_gltf(gltf)
    R.set_camera([0, 1.0, 3.9], host.yaw_quat(0.0), np.deg2rad(np.float32(40.0)), 0.1, 1000.0)
    R.set_seed(19088743, True)
    R.set_params(capi.default_params(max_depth=8))
    R.record_commands()
    R.record_commands()
    R.record_commands_batch(5)
    assert R.accumulated_count == 5
    alb, nrm, dep = R.read_image(capi.IMAGE_ALBEDO), R.read_image(capi.IMAGE_NORMAL), R.read_image(capi.IMAGE_DEPTH)
    assert np.array_equal(read_pfm(out + "_albedo.pfm").view(np.uint32), alb[..., :3].view(np.uint32))
    assert np.array_equal(read_pfm(out + "_normal.pfm").view(np.uint32), nrm[..., :3].view(np.uint32))
    assert np.array_equal(read_pfm(out + "_depth.pfm")[..., 0].view(np.uint32), dep.reshape(64, 96).view(np.uint32))
    R.save_render_texture_to_file(str(tmp_path / "host.png"))
    assert open(out + ".png", "rb").read() == open(tmp_path / "host.png", "rb").read()
    R.close()
    # an existing output is not overwritten (renderer.rs:752-754)
    r = subprocess.run([cli, gltf, "--res", "8x8", "--spp", "1", "--out", out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "exists" in r.stderr.lower()


def test_headless_cli_lit_matches_oracle(oracle, tmp_path):
    """ptx-render --lit (implies --no-ref-quirks): the PFM radiance equals the oracle's lit render of the same file."""
    import subprocess
    from helpers import cornell_lit
    from path_tracer_amd import build
    cli = build.build_cli()
    gltf = os.path.join(ASSETS, "cornell_lit/cornell_lit.gltf")
    out = str(tmp_path / "lit")
    r = subprocess.run([cli, gltf, "--lit", "--res", "64x64", "--spp", "4", "--seed", "19088743", "--depth", "8",
                        "--camera", "0,1.0,3.9", "--yaw", "0", "--fov", "40", "--out", out],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    sc = oracle.scene_from_ingest(cornell_lit(), options=oracle.SCENE_WORLD_NORMALS | oracle.SCENE_TEXTURE_TIMES_FACTOR,
                                  with_material_ext=True)
    cam = oracle.camera_matrices([0, 1.0, 3.9], oracle.yaw_quat(0.0), np.deg2rad(np.float32(40.0)), 1.0, 0.1, 1000.0)
    want = sc.render(cam, 64, 64, 4, SEED, params=oracle.default_params(max_depth=8, integrator=oracle.INTEGRATOR_LIT))
    assert np.array_equal(read_pfm(out + "_albedo.pfm").view(np.uint32), want["albedo"][..., :3].view(np.uint32))
    assert np.array_equal(read_pfm(out + "_normal.pfm").view(np.uint32), want["normal"][..., :3].view(np.uint32))


@pytest.mark.parametrize("where", ["device", "host"])
def test_entity_motion_refits_and_matches_oracle(oracle, monkeypatch, where):
    """todo.md:3 / SURVEY 8f.4: moving an entity refits the acceleration structure (no rebuild) -- on the GPU
    (ptx_refit.hip; PTX_REFIT_VERIFY makes the library compare its nodes and triangles with the host refit byte for
    byte) or on the host (PTX_HOST_REFIT) -- and the image equals the oracle's render of the moved scene bit for bit;
    a changed entity list rebuilds."""
    import os
    monkeypatch.setenv("PTX_REFIT_VERIFY", "1")
    if where == "host":
        monkeypatch.setenv("PTX_HOST_REFIT", "1")
    from helpers import ASSETS, SEED, assert_images_bit_equal, cornell
    from path_tracer_amd import capi, host
    r = host.Renderer(96, 96, 0)
    r.import_gltf(os.path.join(ASSETS, "cornell", "cornell.gltf"))
    fov = np.deg2rad(np.float32(40.0))
    r.set_camera([0.0, 1.0, 3.9], host.yaw_quat(0.0), fov, 0.1, 1000.0)
    r.set_seed(SEED, fixed=True)
    r.set_params(capi.default_params(max_depth=8))
    r.record_commands(); r.record_commands()
    st = r.stats()
    assert st["bvh_builds"] == 1 and st["bvh_refits"] == 0
    ing = cornell()
    geom, (p, q, s), mat = ing.entities[6]
    new_p = p + np.float32([0.3, 0.25, 0.35]); new_s = s * np.float32([1.0, 1.3, 1.0])
    ing.entities[6] = (geom, (new_p, q, new_s), mat)
    r.set_entity_transform(6, new_p, q, new_s)
    r.record_commands()          # scene-dirty frame: refit, accumulated_count restarts
    r.record_commands_batch(3)
    st = r.stats()
    assert st["bvh_builds"] == 1 and st["bvh_refits"] == 1
    assert st["bvh_device_refits"] == (1 if where == "device" else 0)
    got = {k: r.read_image(v) for k, v in (("albedo", capi.IMAGE_ALBEDO), ("normal", capi.IMAGE_NORMAL),
                                           ("depth", capi.IMAGE_DEPTH), ("ids", capi.IMAGE_PRIMARY_ID))}
    cam = oracle.camera_matrices([0.0, 1.0, 3.9], oracle.yaw_quat(0.0), fov, 1.0, 0.1, 1000.0)
    want = oracle.scene_from_ingest(ing).render(cam, 96, 96, 3, SEED, params=oracle.default_params(max_depth=8))
    assert_images_bit_equal(got, want, "cornell after moving the short box (refit)")
    # a far move loosens the tree beyond the limit -> rebuild; same image rule
    far_p = new_p + np.float32([0.0, 40.0, 0.0])
    r.set_entity_transform(6, far_p, q, new_s)
    r.record_commands()
    st = r.stats()
    assert st["bvh_builds"] + st["bvh_refits"] == 3
    # importing another scene changes the entity list -> rebuild
    r.import_gltf(os.path.join(ASSETS, "cornell", "cornell.gltf"), (3.0, 0.0, 0.0))
    r.record_commands()
    st2 = r.stats()
    assert st2["bvh_builds"] == st["bvh_builds"] + 1
    r.close()
