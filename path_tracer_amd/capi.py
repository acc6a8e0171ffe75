"""ctypes binding of libptx_amd.so (include/ptx.h).  Thin: one method per C entry point.

There is no CPU fallback: a missing library or a failed ptx_create raises.
"""
import atexit
import ctypes as C
import os
import weakref

import numpy as np

# Contexts (and host Renderers) still open at interpreter exit are closed by an atexit hook, while the HIP runtime
# is certainly alive: a __del__ that runs during module teardown can come after the runtime has begun to unload,
# and a GPU call there ends the process with an abort.
_live = weakref.WeakSet()


@atexit.register
def _close_live():
    for obj in list(_live):
        try:
            obj.close()
        except Exception:
            pass

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libptx_amd.so")

PTX_OK, PTX_ERR_INVALID, PTX_ERR_CAPACITY, PTX_ERR_HIP, PTX_ERR_STATE = range(5)
IMAGE_ALBEDO, IMAGE_NORMAL, IMAGE_DEPTH, IMAGE_PRIMARY_ID = range(4)
INTEGRATOR_REFERENCE, INTEGRATOR_LIT = 0, 1
SCENE_WORLD_NORMALS, SCENE_TEXTURE_TIMES_FACTOR = 1, 2
TEXTURE_SRGB, TEXTURE_REPEAT, TEXTURE_LINEAR = 1, 2, 4

VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("_p0", "<f4"), ("normal", "<f4", 3), ("_p1", "<f4"),
                         ("uv", "<f4", 2), ("padding", "<f4", 2)])
MATERIAL_DTYPE = np.dtype([("base_color", "<f4", 4), ("has_base_texture", "<u4"), ("base_texture_indice", "<u4"),
                           ("metallic", "<f4"), ("roughness", "<f4")])
OFFSETS_DTYPE = np.dtype([("vertex_offset", "<u4"), ("vertex_count", "<u4"), ("material_offset", "<u4"),
                          ("padding", "<u4"), ("index_offset", "<u4"), ("index_count", "<u4")])
INSTANCE_DTYPE = np.dtype([("xform", "<f4", (3, 4)), ("geometry", "<u4"), ("off", OFFSETS_DTYPE)])
MATERIAL_EXT_DTYPE = np.dtype([("emissive", "<f4", 3), ("transmission", "<f4"), ("ior", "<f4"), ("mr_texture", "<u4"),
                               ("normal_texture", "<u4"), ("emissive_texture", "<u4")])


class Camera(C.Structure):
    _fields_ = [("view_proj", C.c_float * 16), ("view_inverse", C.c_float * 16), ("proj_inverse", C.c_float * 16)]


class Uniforms(C.Structure):
    _fields_ = [("samples_per_pixel", C.c_uint32), ("seed", C.c_uint32), ("accumulated_count", C.c_uint32),
                ("z_near", C.c_float), ("z_far", C.c_float), ("padding", C.c_uint32 * 3)]


class RenderParams(C.Structure):
    _fields_ = [("max_depth", C.c_uint32), ("t_min", C.c_float), ("t_max", C.c_float), ("normal_eps", C.c_float),
                ("rng_sample_offset", C.c_uint32), ("flags", C.c_uint32), ("integrator", C.c_uint32), ("_pad", C.c_uint32)]


class Limits(C.Structure):
    _fields_ = [("max_vertices", C.c_uint64), ("max_indices", C.c_uint64), ("max_materials", C.c_uint32),
                ("max_offsets", C.c_uint32), ("max_textures", C.c_uint32), ("_pad", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("hits", C.c_uint64), ("samples", C.c_uint64),
                ("extend_launches", C.c_uint64), ("extend_ms", C.c_double), ("shade_ms", C.c_double),
                ("total_ms", C.c_double), ("bvh_nodes", C.c_uint64), ("bvh_triangles", C.c_uint64),
                ("bvh_bytes", C.c_uint64), ("bvh_depth", C.c_uint32), ("_pad", C.c_uint32),
                ("nodes_fetched", C.c_uint64), ("tris_tested", C.c_uint64), ("wave_sched", C.c_uint64 * 6),
                ("shadow_rays", C.c_uint64), ("lights", C.c_uint64), ("bvh_builds", C.c_uint64), ("bvh_refits", C.c_uint64),
                ("bvh_device_refits", C.c_uint64), ("tail_launches", C.c_uint64), ("extend_phase_cycles", C.c_uint64 * 8),
                ("extend_append", C.c_uint64 * 3)]


EXPORTS = [
    "ptx_create", "ptx_create_with_limits", "ptx_destroy", "ptx_default_limits", "ptx_default_params",
    "ptx_add_vertices", "ptx_add_indices", "ptx_add_material", "ptx_add_texture", "ptx_create_geometry",
    "ptx_set_instances", "ptx_resize", "ptx_render", "ptx_render_batch", "ptx_read_image", "ptx_write_albedo",
    "ptx_export_albedo_device", "ptx_import_albedo_device", "ptx_reduce_accumulation", "ptx_synchronize",
    "ptx_set_profiling", "ptx_get_stats", "ptx_reset_stats", "ptx_last_error",
    "ptx_set_material_ext", "ptx_set_material_normal_scale", "ptx_set_scene_options", "ptx_update_instances", "ptx_render_static",
    "ptx_set_option", "ptx_get_option",
]
# ptx_option (include/ptx.h)
(OPTION_FRAMES_IN_FLIGHT, OPTION_TAIL_RAYS, OPTION_TAIL_SHIFT, OPTION_TARGET_PATHS, OPTION_MATERIAL_SORT,
 OPTION_EXTEND_WAVES, OPTION_SHADE_GRID, OPTION_HOST_REFIT, OPTION_REFIT_VERIFY, OPTION_TEXEL_SETS,
 OPTION_TEXEL_SET_BYTES) = range(1, 12)

_lib = None


class PtxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ptx error %d: %s" % (code, msg))
        self.code = code


def load():
    """Loads libptx_amd.so.  Raises if it has not been built -- the product never falls back to a CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    # developer overrides: an alternative build of the same library (PTX_LIB), or a directory holding alternative
    # builds of this library and of libpt_host.so (PTX_LIB_DIR, tools/ab.sh) -- the product files are never overwritten
    path = os.environ.get("PTX_LIB") or (os.path.join(os.environ["PTX_LIB_DIR"], "libptx_amd.so") if os.environ.get("PTX_LIB_DIR") else LIB_PATH)
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: run `python -m path_tracer_amd.build` (hipcc, gfx950)" % path)
    L = C.CDLL(path)
    vp = C.c_void_p
    L.ptx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.ptx_create_with_limits.argtypes = [C.c_int, C.POINTER(Limits), C.POINTER(vp)]
    L.ptx_destroy.argtypes = [vp]
    L.ptx_default_limits.argtypes = [C.POINTER(Limits)]
    L.ptx_default_limits.restype = None
    L.ptx_default_params.argtypes = [C.POINTER(RenderParams)]
    L.ptx_default_params.restype = None
    L.ptx_add_vertices.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.ptx_add_indices.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    L.ptx_add_material.argtypes = [vp, vp, C.POINTER(C.c_uint32)]
    L.ptx_add_texture.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
    L.ptx_create_geometry.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]
    L.ptx_set_instances.argtypes = [vp, vp, C.c_uint32]
    L.ptx_update_instances.argtypes = [vp, vp, C.c_uint32]
    L.ptx_resize.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.ptx_render.argtypes = [vp, C.POINTER(Uniforms), C.POINTER(Camera), C.POINTER(RenderParams)]
    L.ptx_render_batch.argtypes = [vp, C.POINTER(Uniforms), C.POINTER(Camera), C.POINTER(RenderParams), C.c_uint32]
    L.ptx_render_static.argtypes = [vp, C.POINTER(Uniforms), C.POINTER(Camera), C.POINTER(RenderParams)]
    L.ptx_read_image.argtypes = [vp, C.c_int, vp, C.c_uint64]
    L.ptx_write_albedo.argtypes = [vp, vp, C.c_uint64]
    L.ptx_export_albedo_device.argtypes = [vp, vp, C.c_uint64, C.c_float]
    L.ptx_import_albedo_device.argtypes = [vp, vp, C.c_uint64, C.c_float]
    L.ptx_reduce_accumulation.argtypes = [C.POINTER(vp), C.c_int, C.c_int]
    L.ptx_synchronize.argtypes = [vp]
    L.ptx_set_profiling.argtypes = [vp, C.c_int]
    L.ptx_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.ptx_reset_stats.argtypes = [vp]
    L.ptx_last_error.argtypes = [vp]
    L.ptx_last_error.restype = C.c_char_p
    L.ptx_set_material_ext.argtypes = [vp, C.c_uint32, vp]
    L.ptx_set_material_normal_scale.argtypes = [vp, C.c_uint32, C.c_float]
    L.ptx_set_scene_options.argtypes = [vp, C.c_uint32]
    L.ptx_set_option.argtypes = [vp, C.c_int, C.c_uint64]
    L.ptx_get_option.argtypes = [vp, C.c_int, C.POINTER(C.c_uint64)]
    _lib = L
    return L


def default_params(**kw):
    p = RenderParams()
    load().ptx_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def default_limits(**kw):
    l = Limits()
    load().ptx_default_limits(C.byref(l))
    for k, v in kw.items():
        setattr(l, k, v)
    return l


class Context:
    """One ptx_context (one GPU)."""

    def __init__(self, device=0, limits=None):
        self._L = load()
        h = C.c_void_p()
        rc = (self._L.ptx_create_with_limits(device, C.byref(limits), C.byref(h)) if limits is not None
              else self._L.ptx_create(device, C.byref(h)))
        if rc:
            raise PtxError(rc, self._L.ptx_last_error(None).decode())
        self.handle = h
        self.width = self.height = 0
        _live.add(self)

    def close(self):
        if getattr(self, "handle", None):
            self._L.ptx_destroy(self.handle)
            self.handle = None

    __del__ = close

    def _check(self, rc):
        if rc:
            raise PtxError(rc, self._L.ptx_last_error(self.handle).decode())

    def add_vertices(self, v):
        v = np.ascontiguousarray(v, dtype=VERTEX_DTYPE)
        first = C.c_uint64()
        self._check(self._L.ptx_add_vertices(self.handle, v.ctypes.data, len(v), C.byref(first)))
        return first.value

    def add_indices(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        first = C.c_uint64()
        self._check(self._L.ptx_add_indices(self.handle, idx.ctypes.data, len(idx), C.byref(first)))
        return first.value

    def add_material(self, m):
        m = np.ascontiguousarray(m, dtype=MATERIAL_DTYPE).reshape(1)
        out = C.c_uint32()
        self._check(self._L.ptx_add_material(self.handle, m.ctypes.data, C.byref(out)))
        return out.value

    def add_texture(self, rgba8, srgb=True):
        """`srgb`: bool, or the flag word TEXTURE_SRGB | TEXTURE_REPEAT | TEXTURE_LINEAR."""
        a = np.ascontiguousarray(rgba8, dtype=np.uint8)
        assert a.ndim == 3 and a.shape[2] == 4
        out = C.c_uint32()
        self._check(self._L.ptx_add_texture(self.handle, a.ctypes.data, a.shape[1], a.shape[0], int(srgb),
                                            C.byref(out)))
        return out.value

    def set_material_ext(self, index, ext):
        e = np.ascontiguousarray(ext, dtype=MATERIAL_EXT_DTYPE).reshape(1)
        self._check(self._L.ptx_set_material_ext(self.handle, index, e.ctypes.data))

    def set_material_normal_scale(self, index, scale):
        self._check(self._L.ptx_set_material_normal_scale(self.handle, int(index), float(scale)))

    def set_scene_options(self, options):
        self._check(self._L.ptx_set_scene_options(self.handle, options))

    def set_option(self, key, value):
        self._check(self._L.ptx_set_option(self.handle, int(key), int(value)))

    def get_option(self, key):
        out = C.c_uint64()
        self._check(self._L.ptx_get_option(self.handle, int(key), C.byref(out)))
        return out.value

    def create_geometry(self, first_vertex, n_vertices, first_index, n_indices):
        out = C.c_uint32()
        self._check(self._L.ptx_create_geometry(self.handle, first_vertex, n_vertices, first_index, n_indices,
                                                C.byref(out)))
        return out.value

    def set_instances(self, inst):
        inst = np.ascontiguousarray(inst, dtype=INSTANCE_DTYPE)
        self._check(self._L.ptx_set_instances(self.handle, inst.ctypes.data, len(inst)))

    def update_instances(self, inst):
        inst = np.ascontiguousarray(inst, dtype=INSTANCE_DTYPE)
        self._check(self._L.ptx_update_instances(self.handle, inst.ctypes.data, len(inst)))

    def resize(self, width, height):
        self._check(self._L.ptx_resize(self.handle, width, height))
        self.width, self.height = width, height

    def render(self, uniforms, camera, params=None):
        self._check(self._L.ptx_render(self.handle, C.byref(uniforms), C.byref(camera),
                                       C.byref(params) if params is not None else None))

    def render_static(self, uniforms, camera, params=None):
        """ptx_render_static: one frame with the wavefront state in the library's file-scope device array (no allocation)."""
        self._check(self._L.ptx_render_static(self.handle, C.byref(uniforms), C.byref(camera),
                                              C.byref(params) if params is not None else None))

    def render_batch(self, uniforms, camera, n_samples, params=None):
        self._check(self._L.ptx_render_batch(self.handle, C.byref(uniforms), C.byref(camera),
                                             C.byref(params) if params is not None else None, n_samples))

    def read_image(self, kind):
        h, w = self.height, self.width
        if kind in (IMAGE_ALBEDO, IMAGE_NORMAL):
            out = np.empty((h, w, 4), dtype=np.float32)
        elif kind == IMAGE_DEPTH:
            out = np.empty((h, w), dtype=np.float32)
        else:
            out = np.empty((h, w, 2), dtype=np.uint32)
        self._check(self._L.ptx_read_image(self.handle, kind, out.ctypes.data, out.nbytes))
        return out

    def write_albedo(self, img):
        a = np.ascontiguousarray(img, dtype=np.float32)
        self._check(self._L.ptx_write_albedo(self.handle, a.ctypes.data, a.nbytes))

    def export_albedo_device(self, device_ptr, nbytes, scale=1.0):
        self._check(self._L.ptx_export_albedo_device(self.handle, device_ptr, nbytes, scale))

    def import_albedo_device(self, device_ptr, nbytes, scale=1.0):
        self._check(self._L.ptx_import_albedo_device(self.handle, device_ptr, nbytes, scale))

    def synchronize(self):
        self._check(self._L.ptx_synchronize(self.handle))

    def set_profiling(self, level):
        self._check(self._L.ptx_set_profiling(self.handle, int(level)))

    def stats(self):
        s = Stats()
        self._check(self._L.ptx_get_stats(self.handle, C.byref(s)))
        d = {k: getattr(s, k) for k, _ in Stats._fields_ if not k.startswith("_")}
        d["wave_sched"] = list(d["wave_sched"])
        d["extend_phase_cycles"] = list(d["extend_phase_cycles"])
        d["extend_append"] = list(d["extend_append"])
        return d

    def reset_stats(self):
        self._check(self._L.ptx_reset_stats(self.handle))


def reduce_accumulation(contexts, root=0):
    L = load()
    arr = (C.c_void_p * len(contexts))(*[c.handle for c in contexts])
    rc = L.ptx_reduce_accumulation(arr, len(contexts), root)
    if rc:
        raise PtxError(rc, L.ptx_last_error(contexts[root].handle).decode())
