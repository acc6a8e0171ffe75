"""MI355X (gfx950) wavefront path-tracing backend for AHL00/path_tracer's render loop.

The product is libptx_amd.so (HIP kernels behind the C ABI of include/ptx.h); this package is the
thin host-side binding plus the headless mirror of the reference's Renderer / Scene interface.
"""
from . import capi  # noqa: F401
