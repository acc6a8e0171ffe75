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
