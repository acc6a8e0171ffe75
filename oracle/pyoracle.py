"""ctypes binding of oracle/_build/libpt_oracle.so -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Never imported from path_tracer_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .gltf_ingest import INSTANCE_DTYPE, MATERIAL_DTYPE, MATERIAL_EXT_DTYPE, VERTEX_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libpt_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, "pt_oracle.c"), os.path.join(_HERE, "pt_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _LIB_PATH


class Camera(C.Structure):
    _fields_ = [("view_proj", C.c_float * 16), ("view_inverse", C.c_float * 16), ("proj_inverse", C.c_float * 16)]


class Uniforms(C.Structure):
    _fields_ = [("samples_per_pixel", C.c_uint32), ("seed", C.c_uint32), ("accumulated_count", C.c_uint32),
                ("z_near", C.c_float), ("z_far", C.c_float), ("padding", C.c_uint32 * 3)]


class Params(C.Structure):
    _fields_ = [("max_depth", C.c_uint32), ("t_min", C.c_float), ("t_max", C.c_float), ("normal_eps", C.c_float),
                ("rng_sample_offset", C.c_uint32), ("flags", C.c_uint32), ("integrator", C.c_uint32), ("_pad", C.c_uint32)]


INTEGRATOR_REFERENCE, INTEGRATOR_LIT = 0, 1
SCENE_WORLD_NORMALS, SCENE_TEXTURE_TIMES_FACTOR = 1, 2


class Texture(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("srgb", C.c_uint32),
                ("_pad", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("n_vertices", C.c_uint64),
                ("indices", C.c_void_p), ("n_indices", C.c_uint64),
                ("materials", C.c_void_p), ("n_materials", C.c_uint64),
                ("instances", C.c_void_p), ("n_instances", C.c_uint64),
                ("textures", C.c_void_p), ("n_textures", C.c_uint64),
                ("material_ext", C.c_void_p), ("options", C.c_uint64), ("normal_scale", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("hits", C.c_uint64), ("samples", C.c_uint64), ("textured_hits", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("shadow_rays", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()          # no-op unless the library is missing or older than its sources (a stale one would silently ignore new
                         # fields of the scene description)
        L = C.CDLL(_LIB_PATH)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.POINTER(SceneDesc)]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_triangle_count.restype = C.c_uint64
        L.orc_scene_triangle_count.argtypes = [C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Uniforms), C.POINTER(Params),
                                 C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.POINTER(Stats), C.c_int]
        L.orc_render_rows.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Uniforms), C.POINTER(Params),
                                      C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.POINTER(Stats), C.c_int]
        L.orc_scene_light_count.restype = C.c_uint64
        L.orc_scene_light_count.argtypes = [C.c_void_p]
        L.orc_scene_light.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                      C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_trace.restype = C.c_int
        L.orc_trace.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float,
                                C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_hash.restype = C.c_uint32
        L.orc_hash.argtypes = [C.c_uint32]
        L.orc_hash2.restype = C.c_uint32
        L.orc_hash2.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_rand.restype = C.c_float
        L.orc_rand.argtypes = [C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_srgb_to_linear.restype = C.c_float
        L.orc_srgb_to_linear.argtypes = [C.c_uint32]
        f3 = C.POINTER(C.c_float)
        L.orc_create_basis.argtypes = [f3, f3, f3]
        L.orc_cosine_hemisphere_sample.argtypes = [C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, f3, f3]
        L.orc_sincos.argtypes = [C.c_float, f3, f3]
        L.orc_sky_color.argtypes = [f3, f3]
        L.orc_transform_matrix.argtypes = [f3, f3, f3, f3]
        L.orc_camera_matrices.argtypes = [f3, f3, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(Camera)]
        L.orc_default_params.argtypes = [C.POINTER(Params)]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def default_params(**kw):
    p = Params()
    lib().orc_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def transform_matrix(pos, quat, scale):
    out = np.zeros((3, 4), dtype=np.float32)
    lib().orc_transform_matrix(_fp(np.asarray(pos, np.float32)), _fp(np.asarray(quat, np.float32)),
                               _fp(np.asarray(scale, np.float32)), _fp(out))
    return out


def camera_matrices(pos, quat, fov_y, aspect, z_near, z_far):
    cam = Camera()
    lib().orc_camera_matrices(_fp(np.asarray(pos, np.float32)), _fp(np.asarray(quat, np.float32)),
                              np.float32(fov_y), np.float32(aspect), np.float32(z_near), np.float32(z_far),
                              C.byref(cam))
    return cam


def yaw_quat(yaw_deg):
    """glam Quat::from_rotation_y."""
    h = np.float32(np.deg2rad(np.float32(yaw_deg))) * np.float32(0.5)
    return np.array([0.0, np.sin(h), 0.0, np.cos(h)], dtype=np.float32)


class OracleScene:
    """Owns the numpy arrays an orc_scene points into."""

    def __init__(self, vertices, indices, materials, instances, textures=(), material_ext=None, options=0, normal_scale=None):
        self.vertices = np.ascontiguousarray(vertices, dtype=VERTEX_DTYPE)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        self.instances = np.ascontiguousarray(instances, dtype=INSTANCE_DTYPE)
        self.tex_arrays = [np.ascontiguousarray(t[0], dtype=np.uint8) for t in textures]
        self.tex_structs = (Texture * max(1, len(textures)))()
        for i, (rgba, srgb) in enumerate(textures):
            self.tex_structs[i] = Texture(self.tex_arrays[i].ctypes.data, rgba.shape[1], rgba.shape[0], int(srgb), 0)
        self.material_ext = None
        if material_ext is not None:
            self.material_ext = np.ascontiguousarray(material_ext, dtype=MATERIAL_EXT_DTYPE)
            assert len(self.material_ext) == len(self.materials)
        self.normal_scale = None
        if normal_scale is not None:
            self.normal_scale = np.ascontiguousarray(normal_scale, dtype=np.float32)
            assert len(self.normal_scale) == len(self.materials)
        d = SceneDesc(self.vertices.ctypes.data, len(self.vertices), self.indices.ctypes.data, len(self.indices),
                      self.materials.ctypes.data, len(self.materials), self.instances.ctypes.data,
                      len(self.instances), C.addressof(self.tex_structs), len(textures),
                      self.material_ext.ctypes.data if self.material_ext is not None else None, options,
                      self.normal_scale.ctypes.data if self.normal_scale is not None else None)
        self.handle = lib().orc_scene_create(C.byref(d))

    def __del__(self):
        if getattr(self, "handle", None):
            lib().orc_scene_destroy(self.handle)
            self.handle = None

    @property
    def triangle_count(self):
        return lib().orc_scene_triangle_count(self.handle)

    def render(self, cam, width, height, n_samples, seed, first_sample=0, z_near=0.1, z_far=100.0, params=None,
               albedo=None, threads=None, want_ids=True, rows=None):
        """Returns dict(albedo HxWx4 f32, normal HxWx4, depth HxW, ids HxWx2 u32, stats).
        rows=(y0, y1): only that band of the frame is rendered (the images keep the full extent; the other rows stay
        as initialised) -- a pixel depends on nothing but its own launch id, so the band equals the full render's rows."""
        params = params or default_params()
        u = Uniforms(1, seed, first_sample, z_near, z_far, (C.c_uint32 * 3)(0, 0, 0))
        if albedo is None:
            albedo = np.zeros((height, width, 4), dtype=np.float32)
        normal = np.zeros((height, width, 4), dtype=np.float32)
        depth = np.zeros((height, width), dtype=np.float32)
        ids = np.full((height, width, 2), 0xFFFFFFFF, dtype=np.uint32) if want_ids else None
        st = Stats()
        threads = threads or min(os.cpu_count() or 1, 16)
        y0, y1 = rows if rows is not None else (0, height)
        lib().orc_render_rows(self.handle, C.byref(cam), C.byref(u), C.byref(params), width, height, y0, y1, n_samples,
                              albedo.ctypes.data, normal.ctypes.data, depth.ctypes.data,
                              ids.ctypes.data if ids is not None else None, C.byref(st), threads)
        return dict(albedo=albedo, normal=normal, depth=depth, ids=ids,
                    stats={k: getattr(st, k) for k, _ in Stats._fields_})

    def lights(self):
        """The lit integrator's light table: list of (verts 3x3, le 3, cdf, pdf_area)."""
        out = []
        for i in range(lib().orc_scene_light_count(self.handle)):
            v = np.zeros(9, np.float32)
            le = np.zeros(3, np.float32)
            cdf, pdf = C.c_float(), C.c_float()
            lib().orc_scene_light(self.handle, i, _fp(v), _fp(le), C.byref(cdf), C.byref(pdf))
            out.append((v.reshape(3, 3), le, cdf.value, pdf.value))
        return out

    def trace(self, org, dirv, t_min=0.001, t_max=10000.0, flags=0):
        t, u, v = C.c_float(), C.c_float(), C.c_float()
        inst, prim = C.c_uint32(), C.c_uint32()
        hit = lib().orc_trace(self.handle, _fp(np.asarray(org, np.float32)), _fp(np.asarray(dirv, np.float32)),
                              t_min, t_max, flags, C.byref(t), C.byref(u), C.byref(v), C.byref(inst), C.byref(prim))
        return (t.value, u.value, v.value, inst.value, prim.value) if hit else None


def scene_from_ingest(ing, options=0, with_material_ext=False):
    """IngestedScene -> OracleScene with the reference's per-entity instance list."""
    inst = ing.instances(transform_matrix)
    return OracleScene(ing.vertices, ing.indices, ing.materials, inst, ing.textures,
                       material_ext=ing.material_ext if with_material_ext else None, options=options,
                       normal_scale=getattr(ing, "normal_scale", None) if with_material_ext else None)
