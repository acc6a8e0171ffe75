"""ctypes binding of libpt_host.so: the C++ mirror of the reference's Renderer / Scene host
interface (path_tracer_amd/host/*.cpp).  Used by bench.py, the headless driver and the tests."""
import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpt_host.so")
EXPORTS = ["pth_last_error", "pth_renderer_create", "pth_renderer_destroy", "pth_renderer_context", "pth_import_gltf",
           "pth_set_camera", "pth_set_seed", "pth_set_params", "pth_set_accumulation_limit", "pth_accumulated_count",
           "pth_resize", "pth_record_commands", "pth_record_commands_batch", "pth_save_image", "pth_scene_counts",
           "pth_scene_copy", "pth_camera_matrices", "pth_decode_image", "pth_encode_png_rgb8", "pth_set_ref_quirks", "pth_set_entity_transform"]
_lib = None


class HostError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    capi.load()
    # developer override (tools/ab.sh): a directory holding an alternative build of BOTH libraries
    path = os.path.join(os.environ["PTX_LIB_DIR"], "libpt_host.so") if os.environ.get("PTX_LIB_DIR") else LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: run `python -m path_tracer_amd.build`" % path)
    L = C.CDLL(path)
    vp = C.c_void_p
    fp = C.POINTER(C.c_float)
    L.pth_last_error.restype = C.c_char_p
    L.pth_renderer_create.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.POINTER(capi.Limits), C.POINTER(vp)]
    L.pth_renderer_destroy.argtypes = [vp]
    L.pth_renderer_destroy.restype = None
    L.pth_renderer_context.argtypes = [vp]
    L.pth_renderer_context.restype = vp
    L.pth_import_gltf.argtypes = [vp, C.c_char_p, C.c_float, C.c_float, C.c_float]
    L.pth_set_ref_quirks.argtypes = [vp, C.c_int]
    L.pth_set_entity_transform.argtypes = [vp, C.c_uint64, fp, fp, fp]
    L.pth_set_camera.argtypes = [vp, fp, fp, C.c_float, C.c_float, C.c_float]
    L.pth_set_seed.argtypes = [vp, C.c_uint32, C.c_int]
    L.pth_set_params.argtypes = [vp, C.POINTER(capi.RenderParams)]
    L.pth_set_accumulation_limit.argtypes = [vp, C.c_uint32]
    L.pth_accumulated_count.argtypes = [vp]
    L.pth_accumulated_count.restype = C.c_uint32
    L.pth_resize.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.pth_record_commands.argtypes = [vp]
    L.pth_record_commands_batch.argtypes = [vp, C.c_uint32]
    L.pth_save_image.argtypes = [vp, C.c_char_p]
    L.pth_scene_counts.argtypes = [vp] + [C.POINTER(C.c_uint64)] * 4
    L.pth_scene_copy.argtypes = [vp, vp, vp, vp, vp]
    L.pth_camera_matrices.argtypes = [fp, fp, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(capi.Camera)]
    L.pth_decode_image.argtypes = [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int), vp, C.c_uint64]
    L.pth_encode_png_rgb8.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, vp]
    _lib = L
    return L


def _check(rc):
    if rc:
        raise HostError(load().pth_last_error().decode())


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def decode_image(path):
    """PNG / baseline JPEG -> HxWxC uint8 (C = 3 or 4), by the C++ host's own codecs."""
    L = load()
    w, h, ch = C.c_uint32(), C.c_uint32(), C.c_int()
    _check(L.pth_decode_image(path.encode(), C.byref(w), C.byref(h), C.byref(ch), None, 0))
    out = np.empty((h.value, w.value, ch.value), dtype=np.uint8)
    _check(L.pth_decode_image(path.encode(), C.byref(w), C.byref(h), C.byref(ch), out.ctypes.data, out.nbytes))
    return out


def encode_png_rgb8(path, rgb):
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    _check(load().pth_encode_png_rgb8(path.encode(), a.shape[1], a.shape[0], a.ctypes.data))


def camera_matrices(pos, quat, fov_y, aspect, z_near, z_far):
    cam = capi.Camera()
    load().pth_camera_matrices(_fp(np.asarray(pos, np.float32)), _fp(np.asarray(quat, np.float32)),
                               np.float32(fov_y), np.float32(aspect), np.float32(z_near), np.float32(z_far), C.byref(cam))
    return cam


def yaw_quat(yaw_deg):
    h = np.float32(np.deg2rad(np.float32(yaw_deg))) * np.float32(0.5)
    return np.array([0.0, np.sin(h), 0.0, np.cos(h)], dtype=np.float32)


class Renderer:
    """Headless mirror of the reference's Renderer (renderer.rs) + Scene::import_gltf."""

    def __init__(self, width=1280, height=720, device=0, limits=None):
        self._L = load()
        h = C.c_void_p()
        _check(self._L.pth_renderer_create(width, height, device, C.byref(limits) if limits is not None else None,
                                           C.byref(h)))
        self.handle = h
        self.width, self.height = width, height
        capi._live.add(self)
        # a non-owning capi.Context view of the renderer's ptx context (stats, readback, multi-GPU export)
        self.ctx = capi.Context.__new__(capi.Context)
        self.ctx._L = capi.load()
        self.ctx.handle = None
        self._ctx_handle = C.c_void_p(self._L.pth_renderer_context(h))

    def _ctx(self):
        c = self.ctx
        c.handle = self._ctx_handle
        c.width, c.height = self.width, self.height
        return c

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.handle = None          # owned by the C++ Renderer
            self._L.pth_renderer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        self.close()

    def import_gltf(self, path, position_offset=(0.0, 0.0, 0.0)):
        _check(self._L.pth_import_gltf(self.handle, path.encode(), *[float(x) for x in position_offset]))

    def set_ref_quirks(self, on):
        """False = SURVEY 8f.2's --no-ref-quirks: world-space normals, texel x baseColorFactor, child nodes,
        sampler wrap modes.  Call before import_gltf."""
        _check(self._L.pth_set_ref_quirks(self.handle, int(bool(on))))

    def set_entity_transform(self, index, pos, quat=(0, 0, 0, 1), scale=(1, 1, 1)):
        """Moves an entity (the reference's UI edits Transform components in place); the acceleration
        structure is refitted, not rebuilt, on the next frame."""
        _check(self._L.pth_set_entity_transform(self.handle, index, _fp(np.asarray(pos, np.float32)),
                                                _fp(np.asarray(quat, np.float32)), _fp(np.asarray(scale, np.float32))))

    def set_camera(self, pos, quat=(0, 0, 0, 1), fov_y=np.deg2rad(np.float32(70.0)), z_near=0.1, z_far=1000.0):
        _check(self._L.pth_set_camera(self.handle, _fp(np.asarray(pos, np.float32)), _fp(np.asarray(quat, np.float32)),
                                      np.float32(fov_y), z_near, z_far))

    def set_seed(self, seed, fixed=True):
        _check(self._L.pth_set_seed(self.handle, seed, int(fixed)))

    def set_params(self, params):
        _check(self._L.pth_set_params(self.handle, C.byref(params)))

    def set_accumulation_limit(self, limit):
        _check(self._L.pth_set_accumulation_limit(self.handle, limit))

    @property
    def accumulated_count(self):
        return self._L.pth_accumulated_count(self.handle)

    def resize_render_buffer(self, width, height):
        _check(self._L.pth_resize(self.handle, width, height))
        self.width, self.height = width, height

    def record_commands(self):
        _check(self._L.pth_record_commands(self.handle))

    def record_commands_batch(self, n):
        _check(self._L.pth_record_commands_batch(self.handle, n))

    def save_render_texture_to_file(self, path):
        _check(self._L.pth_save_image(self.handle, path.encode()))

    def read_image(self, kind):
        return self._ctx().read_image(kind)

    def stats(self):
        return self._ctx().stats()

    def scene_arrays(self):
        nv, ni, nm, ne = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._L.pth_scene_counts(self.handle, C.byref(nv), C.byref(ni), C.byref(nm), C.byref(ne))
        v = np.zeros(nv.value, dtype=capi.VERTEX_DTYPE)
        i = np.zeros(ni.value, dtype=np.uint32)
        m = np.zeros(nm.value, dtype=capi.MATERIAL_DTYPE)
        inst = np.zeros(ne.value, dtype=capi.INSTANCE_DTYPE)
        self._L.pth_scene_copy(self.handle, v.ctypes.data, i.ctypes.data, m.ctypes.data, inst.ctypes.data)
        return v, i, m, inst
