"""CPU tests of the product's host side: the libraries load and export what include/ptx.h declares,
no CPU fallback exists, codecs, host math, BVH builder + traversal logic, stand-in generator."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import ASSETS, ROOT, cornell, lion_boulder


def test_c_abi_exports_every_declared_symbol():
    from path_tracer_amd import capi, host
    header = open(os.path.join(ROOT, "include", "ptx.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char \*)\s*(ptx_[a-z_]+)\s*\(", header, re.M))
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    L = capi.load()
    for name in sorted(declared):
        assert hasattr(L, name), name
    H = host.load()
    for name in host.EXPORTS:
        assert hasattr(H, name), name


def test_option_keys_match_the_header():
    """ptx_set_option / ptx_get_option: the Python binding's keys are the header's enum, and NULL contexts are refused
    without touching a GPU."""
    from path_tracer_amd import capi
    header = open(os.path.join(ROOT, "include", "ptx.h")).read()
    enum = dict((k, int(v)) for k, v in re.findall(r"(PTX_OPTION_[A-Z_]+) = (\d+)", header))
    assert len(enum) == 11
    for name, value in enum.items():
        assert getattr(capi, name[4:]) == value, name
    L = capi.load()
    assert L.ptx_set_option(None, 1, 2) == capi.PTX_ERR_INVALID
    assert L.ptx_get_option(None, 1, None) == capi.PTX_ERR_INVALID


def test_struct_layouts_match_the_reference_glsl():
    from path_tracer_amd import capi
    assert capi.VERTEX_DTYPE.itemsize == 48 and capi.VERTEX_DTYPE.fields["normal"][1] == 16 and capi.VERTEX_DTYPE.fields["uv"][1] == 32
    assert capi.MATERIAL_DTYPE.itemsize == 32 and capi.MATERIAL_DTYPE.fields["roughness"][1] == 28
    assert capi.OFFSETS_DTYPE.itemsize == 24 and capi.INSTANCE_DTYPE.itemsize == 76
    assert C.sizeof(capi.Camera) == 192 and C.sizeof(capi.Uniforms) == 32 and C.sizeof(capi.RenderParams) == 32
    p = capi.default_params()
    assert (p.max_depth, p.rng_sample_offset) == (16, 0)
    assert abs(p.t_min - 0.001) < 1e-9 and p.t_max == 10000.0 and abs(p.normal_eps - 0.001) < 1e-9   # rgen.glsl:18,37-38; rchit.glsl:242
    l = capi.default_limits()
    assert (l.max_vertices, l.max_indices, l.max_materials, l.max_offsets, l.max_textures) == (2500000, 5000000, 10000, 10000, 5000)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from path_tracer_amd import capi, host
    with pytest.raises(capi.PtxError) as e:
        capi.Context(0)
    assert e.value.code == capi.PTX_ERR_HIP
    with pytest.raises(host.HostError):
        host.Renderer(64, 64, 0)


def test_image_codecs_against_pil(tmp_path):
    from PIL import Image
    from path_tracer_amd import host
    for rel, tol in (("sponza/16275776544635328252.png", 0), ("sponza/white.png", 0),
                     ("sponza/11474523244911310074.jpg", 3), ("lion_head_2k/textures/lion_head_diff_2k.jpg", 3)):
        path = os.path.join(ASSETS, rel)
        a = host.decode_image(path)
        b = np.asarray(Image.open(path).convert("RGBA" if a.shape[2] == 4 else "RGB"))
        d = np.abs(a.astype(int) - b.astype(int))
        assert a.shape == b.shape and d.max() <= tol and d.mean() < 0.06, (rel, d.max(), d.mean())
    rgb = np.random.default_rng(0).integers(0, 256, (37, 53, 3), dtype=np.uint8)
    out = str(tmp_path / "x.png")
    host.encode_png_rgb8(out, rgb)
    assert np.array_equal(np.asarray(Image.open(out)), rgb) and np.array_equal(host.decode_image(out), rgb)
    with pytest.raises(host.HostError):
        host.decode_image(os.path.join(ASSETS, "cornell", "cornell.bin"))


def test_host_camera_matrices_equal_the_oracle(oracle):
    from path_tracer_amd import host
    for pos, yaw, fov, asp in (([0, 1.0, 3.9], 0, 40, 1.0), ([-8, 1.6, 0], -90, 70, 16 / 9), ([7, 2, 0], 90, 70, 16 / 9), ([2, 0.25, 0.6], 33, 55, 1.5)):
        a = host.camera_matrices(pos, host.yaw_quat(yaw), np.deg2rad(np.float32(fov)), asp, 0.1, 1000.0)
        b = oracle.camera_matrices(pos, oracle.yaw_quat(yaw), np.deg2rad(np.float32(fov)), asp, 0.1, 1000.0)
        assert bytes(a) == bytes(b)


@pytest.fixture(scope="module")
def bvh_check():
    """g++ build of the product's BVH builder + traversal headers for the host (test-only harness)."""
    out = os.path.join(ROOT, "tests", "_build", "libbvh_check.so")
    srcs = [os.path.join(ROOT, "tests", "native", "bvh_check.cpp"), os.path.join(ROOT, "path_tracer_amd", "csrc", "bvh_build.cpp")]
    deps = srcs + [os.path.join(ROOT, "path_tracer_amd", "csrc", f) for f in ("traverse.h", "bvh8.h", "pt_math.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread"] + srcs + ["-o", out])
    L = C.CDLL(out)
    L.bvhcheck_create.restype = C.c_void_p
    L.bvhcheck_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
    L.bvhcheck_destroy.argtypes = [C.c_void_p]
    L.bvhcheck_built_sah.argtypes = [C.c_void_p]
    L.bvhcheck_built_sah.restype = C.c_double
    L.bvhcheck_hash.argtypes = [C.c_void_p]
    L.bvhcheck_hash.restype = C.c_uint64
    L.bvhcheck_refit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_double)]
    L.bvhcheck_info.argtypes = [C.c_void_p] * 4
    L.bvhcheck_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_float] + [C.c_void_p] * 5
    return L


def _world_tris(ing, oracle):
    inst = ing.instances(oracle.transform_matrix)
    vs, ids = [], []
    for i, it in enumerate(inst):
        off = it["off"]
        idx = ing.indices[off["index_offset"]:off["index_offset"] + off["index_count"]].reshape(-1, 3)
        p = ing.vertices["position"][off["vertex_offset"] + idx]
        m = it["xform"]
        w = np.empty_like(p)
        for r in range(3):
            w[..., r] = ((m[r, 0] * p[..., 0] + m[r, 1] * p[..., 1]) + m[r, 2] * p[..., 2]) + m[r, 3]
        vs.append(w.reshape(-1, 9))
        ids.append(np.stack([np.full(len(idx), i, np.uint32), np.arange(len(idx), dtype=np.uint32)], 1))
    return np.ascontiguousarray(np.concatenate(vs), np.float32), np.ascontiguousarray(np.concatenate(ids), np.uint32)


@pytest.mark.parametrize("scene,center,radius,n", [("cornell", (0, 1, 0), 1.0, 6000), ("lion", (0, 0.2, 0), 0.3, 1500)])
def test_bvh8_traversal_equals_oracle_brute_force(oracle, bvh_check, scene, center, radius, n):
    if scene == "cornell":
        ing = cornell()
    else:
        from oracle.gltf_ingest import IngestedScene
        ing = IngestedScene()
        ing.import_gltf(os.path.join(ASSETS, "lion_head_2k/lion_head_2k.gltf"), decode_image=lambda p: np.zeros((4, 4, 4), np.uint8))
    sc = oracle.scene_from_ingest(ing)
    v, ids = _world_tris(ing, oracle)
    h = bvh_check.bvhcheck_create(v.ctypes.data, ids.ctypes.data, len(v), 4)
    rng = np.random.default_rng(1)
    o = (rng.normal(size=(n, 3)) * radius + center).astype(np.float32)
    d = ((rng.normal(size=(n, 3)) * radius * 0.5 + center).astype(np.float32) - o).astype(np.float32)
    d[::97, 0] = 0.0                                  # axis-parallel rays
    d[::89] *= 1e-3                                   # unnormalised directions
    rays = np.ascontiguousarray(np.concatenate([o, d], 1), np.float32)
    tuv = np.zeros((n, 3), np.float32); hid = np.zeros((n, 2), np.uint32)
    ms = C.c_uint32()
    bvh_check.bvhcheck_trace(h, rays.ctypes.data, n, 1e-3, 1e4, tuv.ctypes.data, hid.ctypes.data, None, None, C.byref(ms))
    nodes, tris, depth = C.c_uint64(), C.c_uint64(), C.c_uint32()
    bvh_check.bvhcheck_info(h, C.byref(nodes), C.byref(tris), C.byref(depth))
    assert tris.value == len(v) and ms.value < depth.value <= 32
    hits = 0
    for i in range(n):
        r = sc.trace(o[i], d[i], flags=1)
        if r is None:
            assert hid[i, 0] == 0xFFFFFFFF
        else:
            hits += 1
            assert (np.float32(r[0]), np.float32(r[1]), np.float32(r[2]), r[3], r[4]) == (tuv[i, 0], tuv[i, 1], tuv[i, 2], hid[i, 0], hid[i, 1])
    assert hits > n // 10
    bvh_check.bvhcheck_destroy(h)


def test_bvh8_refit_after_instance_motion(oracle, bvh_check):
    """ptx_update_instances' refit (todo.md:3): the tree built for the Cornell box still answers closest-hit
    queries exactly after the two boxes have been moved, rotated and scaled -- checked against the oracle's
    brute force on the moved scene -- and its cost figure grows with the motion."""
    ing = cornell()
    v0, ids = _world_tris(ing, oracle)
    h = bvh_check.bvhcheck_create(v0.ctypes.data, ids.ctypes.data, len(v0), 2)
    sah0 = C.c_double()
    assert bvh_check.bvhcheck_refit(h, v0.ctypes.data, ids.ctypes.data, len(v0), C.byref(sah0)) == 1
    # move entities 6 and 7 (short and tall box)
    for e, (pos, yaw, sc) in ((6, ((0.35, 0.3, 0.4), 40.0, (1.0, 1.0, 1.0))), (7, ((-0.3, 0.0, -0.2), -25.0, (0.8, 1.2, 0.8)))):
        geom, (p, q, s), mat = ing.entities[e]
        ing.entities[e] = (geom, (p + np.float32(pos), _quat_mul(oracle.yaw_quat(yaw), q), s * np.float32(sc)), mat)
    v1, ids1 = _world_tris(ing, oracle)
    assert np.array_equal(ids, ids1) and not np.array_equal(v0, v1)
    sah1 = C.c_double()
    assert bvh_check.bvhcheck_refit(h, v1.ctypes.data, ids1.ctypes.data, len(v1), C.byref(sah1)) == 1
    assert sah1.value != sah0.value
    moved = oracle.scene_from_ingest(ing)
    n = 4000
    rng = np.random.default_rng(3)
    o = (rng.normal(size=(n, 3)) + (0, 1, 0)).astype(np.float32)
    d = ((rng.normal(size=(n, 3)) * 0.5 + (0, 1, 0)).astype(np.float32) - o).astype(np.float32)
    rays = np.ascontiguousarray(np.concatenate([o, d], 1), np.float32)
    tuv = np.zeros((n, 3), np.float32); hid = np.zeros((n, 2), np.uint32)
    bvh_check.bvhcheck_trace(h, rays.ctypes.data, n, 1e-3, 1e4, tuv.ctypes.data, hid.ctypes.data, None, None, None)
    hits = 0
    for i in range(n):
        r = moved.trace(o[i], d[i], flags=1)
        if r is None:
            assert hid[i, 0] == 0xFFFFFFFF
        else:
            hits += 1
            assert (np.float32(r[0]), r[3], r[4]) == (tuv[i, 0], hid[i, 0], hid[i, 1])
    assert hits > n // 10
    # a triangle that stops being finite cannot be refitted (the caller rebuilds)
    bad = v1.copy(); bad[5, 0] = np.nan
    assert bvh_check.bvhcheck_refit(h, bad.ctypes.data, ids1.ctypes.data, len(bad), C.byref(sah1)) == 0
    bvh_check.bvhcheck_destroy(h)


def _quat_mul(a, b):
    ax, ay, az, aw = (np.float32(x) for x in a)
    bx, by, bz, bw = (np.float32(x) for x in b)
    return np.float32([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                       aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def test_bvh8_edge_cases(bvh_check):
    def build(v):
        v = np.ascontiguousarray(v, np.float32).reshape(-1, 9)
        ids = np.stack([np.zeros(len(v), np.uint32), np.arange(len(v), dtype=np.uint32)], 1)
        return bvh_check.bvhcheck_create(v.ctypes.data, np.ascontiguousarray(ids).ctypes.data, len(v), 1), len(v)

    def trace(h, o, d):
        rays = np.array([list(o) + list(d)], np.float32)
        tuv = np.zeros((1, 3), np.float32); hid = np.zeros((1, 2), np.uint32)
        bvh_check.bvhcheck_trace(h, rays.ctypes.data, 1, 1e-3, 1e4, tuv.ctypes.data, hid.ctypes.data, None, None, None)
        return tuv[0], hid[0]

    h, _ = build(np.zeros((0, 9)))                    # empty scene
    assert trace(h, (0, 0, 0), (0, 0, -1))[1][0] == 0xFFFFFFFF
    h, _ = build([[-1, -1, -2, 1, -1, -2, 0, 1, -2]])  # a single triangle: leaf root
    t, i = trace(h, (0, 0, 0), (0, 0, -1))
    assert t[0] == 2.0 and i[1] == 0
    # many coincident triangles: equal t everywhere, lowest primitive wins; plus a non-finite one that is skipped
    tri = [-1, -1, -2, 1, -1, -2, 0, 1, -2]
    h, _ = build([tri] * 40 + [[np.nan] * 9])
    t, i = trace(h, (0, 0, 0), (0, 0, -1))
    assert t[0] == 2.0 and i[1] == 0
    n = C.c_uint64(); tr = C.c_uint64(); dp = C.c_uint32()
    bvh_check.bvhcheck_info(h, C.byref(n), C.byref(tr), C.byref(dp))
    assert tr.value == 40


def test_standins_are_deterministic_and_match_the_declared_layout(tmp_path):
    import json
    from path_tracer_amd import standins
    a, b = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    gltf = os.path.join(ASSETS, "sponza", "Sponza.gltf")
    st = standins.make_sponza_bin(gltf, a)
    standins.make_sponza_bin(gltf, b)
    assert open(a, "rb").read() == open(b, "rb").read()
    doc = json.load(open(gltf))
    assert os.path.getsize(a) == doc["buffers"][0]["byteLength"] == 9528220
    assert st["triangles"] == 262267 and st["vertices"] == 192496 and st["degenerate"] < 0.05 * st["triangles"]
    # every primitive stays inside its declared accessor AABB
    from oracle.gltf_ingest import _accessor
    buf = [open(a, "rb").read()]
    for p in doc["meshes"][0]["primitives"]:
        acc = doc["accessors"][p["attributes"]["POSITION"]]
        pos = _accessor(doc, buf, p["attributes"]["POSITION"])
        assert np.all(pos >= np.array(acc["min"], np.float32) - 1e-3) and np.all(pos <= np.array(acc["max"], np.float32) + 1e-3)
        idx = _accessor(doc, buf, p["indices"])
        assert idx.max() < acc["count"]


def test_headless_cli_builds_and_validates_arguments(tmp_path):
    """ptx-render: the binary links against the two libraries and rejects bad command lines
    before touching a GPU; without a HIP device a render attempt fails loudly (no CPU fallback)."""
    from path_tracer_amd import build
    cli = build.build_cli()
    assert os.path.exists(cli)
    r = subprocess.run([cli, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "usage: ptx-render" in r.stderr
    r = subprocess.run([cli], capture_output=True, text=True)
    assert r.returncode == 2
    r = subprocess.run([cli, "x.gltf", "--bogus"], capture_output=True, text=True)
    assert r.returncode == 2 and "unknown option" in r.stderr
    r = subprocess.run([cli, "x.gltf", "--spp", "2", "--gpus", "3"], capture_output=True, text=True)
    assert r.returncode == 2                                     # more GPUs than samples
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([cli, os.path.join(ASSETS, "cornell/cornell.gltf"), "--res", "8x8", "--spp", "1",
                            "--out", str(tmp_path / "o")], capture_output=True, text=True)
        assert r.returncode == 1 and "ptx-render:" in r.stderr and not (tmp_path / "o.png").exists()


def test_native_host_code_under_asan_ubsan(tmp_path):
    """Host-side sanitizer run (GPU sanitizers are not available on this pool): the PNG / JPEG decoders on whole,
    truncated and corrupted files, the glTF JSON parser on broken documents, and BVH8 build / refit / traversal on
    random soups with degenerate, coincident and non-finite triangles, all under -fsanitize=address,undefined."""
    exe = os.path.join(ROOT, "tests", "_build", "sanitize_main")
    srcs = [os.path.join(ROOT, "tests", "native", "sanitize_main.cpp"), os.path.join(ROOT, "path_tracer_amd", "csrc", "bvh_build.cpp"),
            os.path.join(ROOT, "path_tracer_amd", "host", "pth_image.cpp")]
    deps = srcs + [os.path.join(ROOT, "path_tracer_amd", "csrc", f) for f in ("traverse.h", "bvh8.h", "pt_math.h")] + \
        [os.path.join(ROOT, "path_tracer_amd", "host", f) for f in ("pth_image.h", "pth_json.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                               "-ffp-contract=off", "-pthread"] + srcs + ["-lz", "-o", exe])
    # the two smallest real textures of each codec keep the run to a few seconds
    sponza = os.path.join(ASSETS, "sponza")
    pngs = sorted((os.path.getsize(os.path.join(sponza, f)), f) for f in os.listdir(sponza) if f.endswith(".png"))
    jpgs = sorted((os.path.getsize(os.path.join(sponza, f)), f) for f in os.listdir(sponza) if f.endswith(".jpg"))
    imgs = [os.path.join(sponza, pngs[0][1]), os.path.join(sponza, jpgs[0][1])]
    # crafted files: a DHT whose code lengths over-subscribe the code space (three 1-bit codes: such a table indexed
    # past the decoder's 512-entry fast table), an SOS segment with an empty payload, and a directory as "file"
    jpg = open(imgs[1], "rb").read()
    dht = jpg.index(b"\xff\xc4")
    bad_dht = bytearray(jpg)
    bad_dht[dht + 5] = 3                                   # marker 2 + length 2 + Tc/Th 1 -> bits[1]
    sos = jpg.index(b"\xff\xda")
    bad_sos = jpg[:sos] + b"\xff\xda\x00\x02" + jpg[sos + 4:]
    for name, data in (("bad_dht.jpg", bytes(bad_dht)), ("bad_sos.jpg", bad_sos)):
        (tmp_path / name).write_bytes(data)
        imgs.append("bad:" + str(tmp_path / name))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, str(tmp_path / "img.tmp")] + imgs, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "images decoded" in r.stdout


def test_threaded_build_equals_serial_build(oracle, bvh_check):
    """The builder's parallel parts (subtree tasks, level-parallel collapse) only divide the work: the tree is the
    serial build's, byte for byte, for any thread count -- on a mesh large enough to spawn tasks and to split levels."""
    from oracle.gltf_ingest import IngestedScene
    ing = IngestedScene()
    ing.import_gltf(os.path.join(ASSETS, "lion_head_2k/lion_head_2k.gltf"), decode_image=lambda p: np.zeros((4, 4, 4), np.uint8))
    ing.import_gltf(os.path.join(ASSETS, "boulder_01_2k/boulder_01_2k.gltf"), (-2, 0, 0), decode_image=lambda p: np.zeros((4, 4, 4), np.uint8))
    v, ids = _world_tris(ing, oracle)
    assert len(v) > 100000
    hashes = []
    for threads in (1, 3, 16):
        h = bvh_check.bvhcheck_create(v.ctypes.data, ids.ctypes.data, len(v), threads)
        hashes.append(bvh_check.bvhcheck_hash(h))
        # the build's own cost figure is what a refit of the unmoved triangles reports (the rebuild-or-refit rule compares the two)
        sah = C.c_double()
        built = bvh_check.bvhcheck_built_sah(h)
        assert bvh_check.bvhcheck_refit(h, v.ctypes.data, ids.ctypes.data, len(v), C.byref(sah)) == 1
        assert built > 1.0 and sah.value == built
        assert bvh_check.bvhcheck_hash(h) == hashes[-1]          # and that refit reproduces the built nodes
        bvh_check.bvhcheck_destroy(h)
    assert hashes[0] == hashes[1] == hashes[2]
