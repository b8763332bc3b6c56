"""Loader of the native library ``libtrt.so`` (C ABI of ``include/trt.h``).

The library is built in-tree by ``__graft_entry__.build()`` /
``toroidal_ray_tracing_amd/csrc/Makefile``.  There is no Python or CPU fallback: if the
shared object is missing or cannot be loaded, importing a symbol raises.
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRT_LIB") or os.path.join(_HERE, "libtrt.so")  # TRT_LIB: A/B builds (tools/)

#: every entry point ``include/trt.h`` declares: name -> (restype, argtypes)
SYMBOLS = {
    "trt_version": (C.c_int, []),
    "trt_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "trt_destroy": (None, [C.c_void_p]),
    "trt_last_error": (C.c_char_p, [C.c_void_p]),
    "trt_set_solver": (C.c_int, [C.c_void_p, C.c_int]),
    "trt_trace": (C.c_int, [C.c_void_p, C.POINTER(abi.trt_rays), C.POINTER(abi.trt_scene),
                            C.c_float, C.c_float, C.POINTER(abi.trt_hits)]),
    "trt_trace_dev": (C.c_int, [C.c_void_p, C.POINTER(abi.trt_rays), C.POINTER(abi.trt_scene),
                                C.c_float, C.c_float, C.POINTER(abi.trt_hits), C.c_void_p]),
    "trt_render": (C.c_int, [C.c_void_p, C.POINTER(abi.trt_globals), C.POINTER(abi.trt_push),
                             C.POINTER(abi.trt_scene), C.c_uint32, C.c_uint32, C.c_int,
                             C.c_void_p, C.POINTER(abi.trt_hits)]),
    "trt_render_dev": (C.c_int, [C.c_void_p, C.POINTER(abi.trt_globals), C.POINTER(abi.trt_push),
                                 C.POINTER(abi.trt_scene), C.c_uint32, C.c_uint32, C.c_uint32,
                                 C.c_uint32, C.c_int, C.c_void_p, C.POINTER(abi.trt_hits),
                                 C.c_void_p, C.c_void_p]),
    "trt_render_tiled_dev": (C.c_int, [C.c_void_p, C.POINTER(abi.trt_globals), C.POINTER(abi.trt_push),
                                       C.POINTER(abi.trt_scene), C.c_uint32, C.c_uint32,
                                       C.POINTER(abi.trt_tiling), C.c_int, C.c_void_p,
                                       C.POINTER(abi.trt_hits), C.c_void_p, C.c_void_p]),
    "trt_render_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(abi.trt_frame), C.c_uint32, C.POINTER(abi.trt_scene), C.c_uint32,
                                       C.c_uint32, C.POINTER(abi.trt_tiling), C.c_int, C.c_void_p]),
    "trt_tiling_rows": (C.c_uint32, [C.POINTER(abi.trt_tiling), C.c_uint32]),
    "trt_post_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "trt_splat_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, abi.f32p, C.c_uint32, C.c_uint32, abi.f32p,
                                C.c_float, C.c_void_p, C.c_void_p]),
    "trt_enable_stats": (C.c_int, [C.c_void_p, C.c_int]),
    "trt_get_stats": (C.c_int, [C.c_void_p, C.POINTER(abi.trt_stats)]),
    "trt_set_render_variant": (C.c_int, [C.c_void_p, C.c_char_p]),
    "trt_get_render_variant": (C.c_char_p, [C.c_void_p]),
    "trt_set_classification": (C.c_int, [C.c_void_p, C.c_int]),
}

_lib = None


def load():
    """Load ``libtrt.so`` and declare the prototypes.  Raises if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME
        # libamdhip64.so.7 — the same SONAME libtrt.so links against).  Two HIP runtimes in one
        # process cannot both own the GPU, so torch must be loaded FIRST: the dynamic loader
        # then resolves libtrt.so's dependency to the runtime torch already mapped, and device
        # pointers / streams are shared between the two.  Without torch installed libtrt.so
        # simply uses /opt/rocm's runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
