"""ctypes wrapper of the CPU oracle (``oracle/libtrt_oracle.so``) — TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import
this module.  See ``oracle/trt_oracle.c`` for what is restated and the parity status
("parity unpinned": the reference holds no golden vectors and no torus arithmetic).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from toroidal_ray_tracing_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtrt_oracle.so")
_lib = None


def build(force=False):
    """(Re)build the oracle with its Makefile if missing or stale."""
    srcs = [os.path.join(_HERE, f) for f in ("trt_oracle.c", "trt_solve.inc", "Makefile")]
    srcs.append(os.path.join(_HERE, "..", "include", "trt.h"))
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libtrt_oracle.so"], check=True,
                       capture_output=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.oracle_render.restype = C.c_int
        L.oracle_render.argtypes = [C.POINTER(abi.trt_globals), C.POINTER(abi.trt_push),
                                    C.POINTER(abi.trt_scene), C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                    C.POINTER(abi.trt_hits), C.c_void_p,
                                    C.POINTER(abi.trt_stats)]
        L.oracle_trace.restype = C.c_int
        L.oracle_trace.argtypes = [C.POINTER(abi.trt_rays), C.POINTER(abi.trt_scene), C.c_float,
                                   C.c_float, C.c_int, C.c_int, C.POINTER(abi.trt_hits),
                                   C.POINTER(abi.trt_stats)]
        L.oracle_raygen.restype = C.c_int
        L.oracle_raygen.argtypes = [C.POINTER(abi.trt_globals), C.POINTER(abi.trt_push),
                                    C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32,
                                    abi.f32p, abi.f32p]
        L.oracle_toroidal_frame.restype = C.c_int
        L.oracle_toroidal_frame.argtypes = [C.POINTER(abi.trt_globals), C.POINTER(abi.trt_push),
                                            abi.f32p]
        L.oracle_reflect.restype = None
        L.oracle_reflect.argtypes = [abi.f32p, abi.f32p, abi.f32p]
        L.oracle_torus_first_hit.restype = C.c_int
        L.oracle_torus_first_hit.argtypes = [C.POINTER(abi.trt_torus), abi.f32p, abi.f32p,
                                             C.c_float, C.c_float, C.c_int,
                                             C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.oracle_post.restype = C.c_int
        L.oracle_post.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.oracle_splat.restype = C.c_int
        L.oracle_splat.argtypes = [C.c_void_p, C.c_uint64, abi.f32p, C.c_uint32, C.c_uint32, abi.f32p, C.c_float,
                                   C.c_void_p]
        L.oracle_max_threads.restype = C.c_int
        L.oracle_set_enclosure_cull.restype = None
        L.oracle_set_enclosure_cull.argtypes = [C.c_int]
        _lib = L
    return _lib


def max_threads():
    return int(lib().oracle_max_threads())


def set_enclosure_cull(on):
    """Tests only: 0 makes every query test every torus (the semantics before the enclosure cull of T3, trt_oracle.c);
    1 (the default, what the library implements) skips the tori whose tube lies inside a tube the ray starts outside of."""
    lib().oracle_set_enclosure_cull(1 if on else 0)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed: {abi.ERROR_NAMES.get(rc, rc)}")


def trace(scene, o, d, tmin=0.001, tmax=10000.0, precision=abi.TRT_SOLVE_F32, nthreads=1):
    """o, d: (n,3) arrays.  Returns (hits dict of SoA arrays, stats dict)."""
    o = np.ascontiguousarray(np.asarray(o, np.float32).T)
    d = np.ascontiguousarray(np.asarray(d, np.float32).T)
    n = o.shape[1]
    rays = abi.rays_struct([o[0], o[1], o[2], d[0], d[1], d[2]], n)
    out = abi.alloc_hits(n)
    hs = abi.hits_struct(out)
    st = abi.trt_stats()
    _check(lib().oracle_trace(C.byref(rays), C.byref(scene.c), tmin, tmax, precision, nthreads,
                              C.byref(hs), C.byref(st)), "trace")
    return out, {k: int(getattr(st, k)) for k in abi.STAT_FIELDS}


def render(scene, g, pc, W, H, camera=abi.TRT_CAMERA_PINHOLE, rows=None,
           precision=abi.TRT_SOLVE_F32, nthreads=1, want_hits=True, want_rendered=False):
    """Full-frame buffers are returned; only rows [rows[0], rows[1]) are written."""
    r0, r1 = (0, H) if rows is None else rows
    rgba = np.zeros((H, W, 4), np.float32)
    hits = abi.alloc_hits(W * H) if want_hits else None
    if hits is not None:
        for k in hits:
            hits[k][...] = 0
    hs = abi.hits_struct(hits) if hits is not None else None
    rendered = np.zeros((W * H, 16), np.float32) if want_rendered else None
    st = abi.trt_stats()
    _check(lib().oracle_render(C.byref(g), C.byref(pc), C.byref(scene.c), W, H, r0, r1, camera,
                               precision, nthreads, abi.ptr(rgba),
                               C.byref(hs) if hs is not None else None, abi.ptr(rendered),
                               C.byref(st)), "render")
    stats = {k: int(getattr(st, k)) for k in abi.STAT_FIELDS}
    return rgba, hits, rendered, stats


def raygen(g, pc, W, H, camera, x, y):
    o = (C.c_float * 3)()
    d = (C.c_float * 3)()
    _check(lib().oracle_raygen(C.byref(g), C.byref(pc), W, H, camera, x, y, o, d), "raygen")
    return np.array(o[:], np.float32), np.array(d[:], np.float32)


def toroidal_frame(g, pc):
    buf = (C.c_float * 5)()
    _check(lib().oracle_toroidal_frame(C.byref(g), C.byref(pc), buf), "toroidal_frame")
    return {"omega": buf[0], "theta": buf[1], "eye": np.array(buf[2:5], np.float32)}


def reflect(i, n):
    a = (C.c_float * 3)(*[float(v) for v in i])
    b = (C.c_float * 3)(*[float(v) for v in n])
    r = (C.c_float * 3)()
    lib().oracle_reflect(a, b, r)
    return np.array(r[:], np.float32)


def torus_first_hit(torus, o, d, tmin=0.001, tmax=10000.0, precision=abi.TRT_SOLVE_F32):
    """torus: (center, R, r).  Returns (t or None, polynomial evaluations)."""
    T = abi.trt_torus()
    T.center[:] = [float(v) for v in torus[0]]
    T.R, T.r, T.matId = float(torus[1]), float(torus[2]), 0
    oo = (C.c_float * 3)(*[float(v) for v in o])
    dd = (C.c_float * 3)(*[float(v) for v in d])
    t = C.c_double()
    ne = C.c_int()
    hit = lib().oracle_torus_first_hit(C.byref(T), oo, dd, tmin, tmax, precision, C.byref(t),
                                       C.byref(ne))
    return (t.value if hit else None), ne.value


def post(rgba):
    """Tonemap pass: returns (float32 image, uint8 image) of the same shape as rgba (…,4)."""
    a = np.ascontiguousarray(rgba, np.float32)
    f = np.empty_like(a)
    u = np.empty(a.shape, np.uint8)
    _check(lib().oracle_post(abi.ptr(a), a.size // 4, abi.ptr(f), abi.ptr(u)), "post")
    return f, u


def splat(points, view_proj, W, H, clear=(0.8, 0.8, 0.8, 1.0), point_size=2.5):
    """points: (n, 8) float32 array of trt_point records (pos.xyzw, color.xyzw)."""
    pts = np.ascontiguousarray(points, np.float32)
    vp = (C.c_float * 16)(*np.asarray(view_proj, np.float32).T.reshape(-1).tolist())
    cc = (C.c_float * 4)(*[float(v) for v in clear])
    out = np.empty((H, W, 4), np.float32)
    _check(lib().oracle_splat(abi.ptr(pts), len(pts), vp, W, H, cc, point_size, abi.ptr(out)), "splat")
    return out
