"""Imported first by the tools that turn TRT_* knobs: they load the -DTRT_TUNING build
(toroidal_ray_tracing_amd/libtrt_tuning.so) — the release library reads no environment variable —
and re-read the knobs after changing them in-process with reload(tracer)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TRT_LIB", os.path.join(ROOT, "toroidal_ray_tracing_amd", "libtrt_tuning.so"))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def reload(tracer):
    fn = getattr(tracer._L, "trt_debug_reload_tuning", None)
    if fn is None:
        raise RuntimeError("TRT_LIB is not a -DTRT_TUNING build: knobs are compiled out")
    fn.restype = None
    import ctypes
    fn.argtypes = [ctypes.c_void_p]
    fn(tracer._h)
