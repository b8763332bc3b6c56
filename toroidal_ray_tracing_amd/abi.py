"""ctypes mirror of ``include/trt.h`` — the C ABI structs, enums and helpers.

Every struct here is byte-for-byte the C declaration of the same name; the layouts in turn
mirror the reference's host/device structs (vk_raytracing_tutorial_KHR/…):

* ``trt_globals``   = ``GlobalUniforms``    ray_tracing__before/shaders/host_device.h:69-75
* ``trt_push``      = ``PushConstantRay``   ray_tracing__before/shaders/host_device.h:90-98
* ``trt_material``  = ``WaveFrontMaterial`` ray_tracing_reflections/shaders/host_device.h:103-115
* ``trt_rendered_data`` = ``RenderedData``  ray_tracing__before/shaders/host_device.h:101-107
"""
import ctypes as C

import numpy as np

TRT_OK, TRT_E_INVALID, TRT_E_NO_DEVICE, TRT_E_HIP, TRT_E_SCENE, TRT_E_NOMEM = 0, -1, -2, -3, -4, -5
TRT_MAX_TORI = 8
TRT_MAX_MATERIALS = 8
TRT_MAX_BATCH = 8
TRT_CAMERA_PINHOLE, TRT_CAMERA_TOROIDAL = 0, 1
TRT_CLASSIFY_AUTO, TRT_CLASSIFY_MACRO, TRT_CLASSIFY_TILE = -1, 0, 1
TRT_SOLVE_F32, TRT_SOLVE_F64, TRT_SOLVE_DK_F32, TRT_SOLVE_DK_F64 = 0, 1, 2, 3
TRT_SOLVE_FERRARI_F32, TRT_SOLVE_FERRARI_F64 = 4, 5

ERROR_NAMES = {
    TRT_E_INVALID: "TRT_E_INVALID", TRT_E_NO_DEVICE: "TRT_E_NO_DEVICE", TRT_E_HIP: "TRT_E_HIP",
    TRT_E_SCENE: "TRT_E_SCENE", TRT_E_NOMEM: "TRT_E_NOMEM",
}

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)


class trt_globals(C.Structure):
    _fields_ = [("viewProj", C.c_float * 16), ("viewInverse", C.c_float * 16),
                ("projInverse", C.c_float * 16), ("center", C.c_float * 3)]


class trt_push(C.Structure):
    _fields_ = [("clearColor", C.c_float * 4), ("lightPosition", C.c_float * 3),
                ("lightIntensity", C.c_float), ("lightType", C.c_int32),
                ("maxDepth", C.c_int32), ("rho", C.c_float)]


class trt_material(C.Structure):
    _fields_ = [("ambient", C.c_float * 3), ("diffuse", C.c_float * 3),
                ("specular", C.c_float * 3), ("transmittance", C.c_float * 3),
                ("emission", C.c_float * 3), ("shininess", C.c_float), ("ior", C.c_float),
                ("dissolve", C.c_float), ("illum", C.c_int32), ("textureId", C.c_int32)]


class trt_torus(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("R", C.c_float), ("r", C.c_float),
                ("matId", C.c_int32)]


class trt_scene(C.Structure):
    _fields_ = [("tori", C.POINTER(trt_torus)), ("n_tori", C.c_uint32),
                ("materials", C.POINTER(trt_material)), ("n_materials", C.c_uint32)]


class trt_rendered_data(C.Structure):
    _fields_ = [("pos", C.c_float * 4), ("color", C.c_float * 4),
                ("rayOrigin", C.c_float * 4), ("rayDir", C.c_float * 4)]


class trt_rays(C.Structure):
    _fields_ = [("ox", C.c_void_p), ("oy", C.c_void_p), ("oz", C.c_void_p),
                ("dx", C.c_void_p), ("dy", C.c_void_p), ("dz", C.c_void_p),
                ("n", C.c_uint64)]


class trt_hits(C.Structure):
    _fields_ = [("t", C.c_void_p), ("px", C.c_void_p), ("py", C.c_void_p), ("pz", C.c_void_p),
                ("nx", C.c_void_p), ("ny", C.c_void_p), ("nz", C.c_void_p), ("id", C.c_void_p)]


class trt_point(C.Structure):
    _fields_ = [("pos", C.c_float * 4), ("color", C.c_float * 4)]


class trt_tiling(C.Structure):
    _fields_ = [("group_rows", C.c_uint32), ("n_parts", C.c_uint32), ("part", C.c_uint32),
                ("compact", C.c_uint32)]


class trt_frame(C.Structure):
    """One frame of a batch (trt_render_batch_dev): what differs from frame to frame in a frame loop."""
    _fields_ = [("g", C.POINTER(trt_globals)), ("pc", C.POINTER(trt_push)), ("rgba_dev", C.c_void_p),
                ("first_hit_dev", C.POINTER(trt_hits))]


class trt_stats(C.Structure):
    _fields_ = [("primary_tests", C.c_uint64), ("bounce_tests", C.c_uint64),
                ("shadow_tests", C.c_uint64), ("pixels", C.c_uint64),
                ("traced_tests", C.c_uint64), ("solved_tests", C.c_uint64),
                ("evaluations", C.c_uint64), ("reserved", C.c_uint64)]


STAT_FIELDS = ("primary_tests", "bounce_tests", "shadow_tests", "pixels", "traced_tests", "solved_tests", "evaluations")


assert C.sizeof(trt_globals) == 204 and C.sizeof(trt_push) == 44
assert C.sizeof(trt_material) == 80 and C.sizeof(trt_torus) == 24
assert C.sizeof(trt_rendered_data) == 64 and C.sizeof(trt_point) == 32

HIT_FIELDS = ("t", "px", "py", "pz", "nx", "ny", "nz", "id")
RAY_FIELDS = ("ox", "oy", "oz", "dx", "dy", "dz")


class Scene:
    """Owns the ctypes arrays behind a ``trt_scene`` (keeps them alive)."""

    def __init__(self, tori, materials):
        """tori: iterable of (center(3), R, r, matId); materials: iterable of dicts with the
        WaveFrontMaterial field names (missing fields default to 0, textureId to -1,
        dissolve/ior to 1)."""
        tori = list(tori)
        materials = list(materials)
        self._tori = (trt_torus * len(tori))()
        for dst, (c, R, r, mid) in zip(self._tori, tori):
            dst.center[:] = [float(v) for v in c]
            dst.R, dst.r, dst.matId = float(R), float(r), int(mid)
        self._mats = (trt_material * len(materials))()
        for dst, m in zip(self._mats, materials):
            for key in ("ambient", "diffuse", "specular", "transmittance", "emission"):
                getattr(dst, key)[:] = [float(v) for v in m.get(key, (0.0, 0.0, 0.0))]
            dst.shininess = float(m.get("shininess", 0.0))
            dst.ior = float(m.get("ior", 1.0))
            dst.dissolve = float(m.get("dissolve", 1.0))
            dst.illum = int(m.get("illum", 0))
            dst.textureId = int(m.get("textureId", -1))
        self.c = trt_scene(self._tori, len(tori), self._mats, len(materials))

    @property
    def n_tori(self):
        return int(self.c.n_tori)

    def tori_list(self):
        return [(tuple(t.center), t.R, t.r, t.matId) for t in self._tori]


def make_globals(view_inverse, proj_inverse, view_proj=None, center=(0.0, 0.0, 0.0)):
    """Build a ``trt_globals`` from 4x4 numpy matrices given in the usual math convention
    (``M[row, col]``); they are stored column-major like nvmath::mat4f."""
    g = trt_globals()
    vp = np.eye(4) if view_proj is None else view_proj
    g.viewProj[:] = np.asarray(vp, np.float32).T.reshape(-1).tolist()
    g.viewInverse[:] = np.asarray(view_inverse, np.float32).T.reshape(-1).tolist()
    g.projInverse[:] = np.asarray(proj_inverse, np.float32).T.reshape(-1).tolist()
    g.center[:] = [float(v) for v in center]
    return g


def make_push(clear=(1.0, 1.0, 1.0, 1.0), light_pos=(10.0, 15.0, 8.0), light_intensity=100.0,
              light_type=0, max_depth=10, rho=0.0):
    """``PushConstantRay`` with the reference defaults (REFL/hello_vulkan.h:74-80,157;
    REFL/main.cpp:212)."""
    p = trt_push()
    p.clearColor[:] = [float(v) for v in clear]
    p.lightPosition[:] = [float(v) for v in light_pos]
    p.lightIntensity = float(light_intensity)
    p.lightType = int(light_type)
    p.maxDepth = int(max_depth)
    p.rho = float(rho)
    return p


def ptr(a):
    """Address of a numpy array (or None) as c_void_p."""
    return None if a is None else C.c_void_p(a.ctypes.data)


def rays_struct(arrays, n):
    r = trt_rays()
    for k, a in zip(RAY_FIELDS, arrays):
        setattr(r, k, a if isinstance(a, int) else a.ctypes.data)
    r.n = int(n)
    return r


def hits_struct(arrays):
    """arrays: dict name -> numpy array | int address | None."""
    h = trt_hits()
    for k in HIT_FIELDS:
        a = arrays.get(k)
        setattr(h, k, None if a is None else (a if isinstance(a, int) else a.ctypes.data))
    return h


def alloc_hits(n):
    d = {k: np.empty(n, np.float32) for k in HIT_FIELDS[:-1]}
    d["id"] = np.empty(n, np.int32)
    return d
