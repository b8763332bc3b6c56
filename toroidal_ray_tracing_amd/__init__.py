"""toroidal_ray_tracing_amd — MI355X-native hot path of raffaelecicellini/toroidal_ray_tracing.

Ray–torus intersection, normal/reflection bounce loop and Phong shading as hand-written
gfx950 HIP kernels behind the C ABI of ``include/trt.h``.  This package only holds what the
path needs: ``csrc/`` (kernels + C ABI), the ctypes binding (``lib``, ``abi``, ``tracer``),
camera/scene inputs (``camera``) and the multi-GPU row-band tiling (``distributed``).
"""
from . import abi  # noqa: F401
from .abi import (TRT_CAMERA_PINHOLE, TRT_CAMERA_TOROIDAL, TRT_SOLVE_F32, TRT_SOLVE_F64,  # noqa: F401
                  Scene, make_globals, make_push)

__all__ = ["abi", "Scene", "make_globals", "make_push", "TRT_CAMERA_PINHOLE", "TRT_CAMERA_TOROIDAL",
           "TRT_SOLVE_F32", "TRT_SOLVE_F64"]
