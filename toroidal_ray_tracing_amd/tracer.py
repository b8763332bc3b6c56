"""Host-side Python mirror of the reference's ray-tracing entry point, over the C ABI.

``Tracer.raytrace`` plays the part of ``HelloVulkan::raytrace(cmdBuf, clearColor)``
(vk_raytracing_tutorial_KHR/ray_tracing_reflections/hello_vulkan.cpp:913-935,
ray_tracing__before/hello_vulkan.cpp:936-958): push constants are refreshed from the
light state and the clear colour, then a W×H launch is issued.  All ray work happens in
``libtrt.so`` on the GPU; this module only marshals arguments.
"""
import ctypes as C

import numpy as np

from . import abi, lib


class TrtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"{abi.ERROR_NAMES.get(code, code)}: {message}")
        self.code = code


class Tracer:
    """One context per device (not re-entrant), like one ``HelloVulkan`` instance."""

    def __init__(self, device=0):
        self._L = lib.load()
        h = C.c_void_p()
        rc = self._L.trt_create(int(device), C.byref(h))
        if rc != 0:
            raise TrtError(rc, (self._L.trt_last_error(None) or b"").decode())
        self._h = h
        self.device = int(device)

    # -- lifetime ----------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.trt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise TrtError(rc, (self._L.trt_last_error(self._h) or b"").decode())

    # -- configuration -----------------------------------------------------------------
    def set_solver(self, precision):
        self._check(self._L.trt_set_solver(self._h, int(precision)))

    def set_render_variant(self, name):
        self._check(self._L.trt_set_render_variant(self._h, name.encode()))

    def set_classification(self, level):
        """abi.TRT_CLASSIFY_AUTO (-1) | _MACRO (0) | _TILE (1): level of the tile classification."""
        self._check(self._L.trt_set_classification(self._h, int(level)))

    def render_variant(self):
        return self._L.trt_get_render_variant(self._h).decode()

    def enable_stats(self, on=True):
        self._check(self._L.trt_enable_stats(self._h, int(bool(on))))

    def stats(self):
        st = abi.trt_stats()
        self._check(self._L.trt_get_stats(self._h, C.byref(st)))
        return {k: int(getattr(st, k)) for k in abi.STAT_FIELDS}

    # -- trace(rays_in -> hits_out) ----------------------------------------------------
    def trace(self, scene, o, d, tmin=0.001, tmax=10000.0):
        """Host arrays: o, d of shape (n,3).  Returns a dict of SoA hit arrays."""
        o = np.ascontiguousarray(np.asarray(o, np.float32).T)
        d = np.ascontiguousarray(np.asarray(d, np.float32).T)
        n = o.shape[1]
        rays = abi.rays_struct([o[0], o[1], o[2], d[0], d[1], d[2]], n)
        out = abi.alloc_hits(n)
        hs = abi.hits_struct(out)
        self._check(self._L.trt_trace(self._h, C.byref(rays), C.byref(scene.c), tmin, tmax, C.byref(hs)))
        return out

    def trace_dev(self, scene, ray_ptrs, n, hit_ptrs, tmin=0.001, tmax=10000.0, stream=0):
        """Device pointers (ints): ray_ptrs = 6 addresses, hit_ptrs = dict name -> address."""
        rays = abi.rays_struct([int(p) for p in ray_ptrs], n)
        hs = abi.hits_struct({k: (int(v) if v else None) for k, v in hit_ptrs.items()})
        self._check(self._L.trt_trace_dev(self._h, C.byref(rays), C.byref(scene.c), tmin, tmax,
                                          C.byref(hs), C.c_void_p(int(stream))))

    # -- render -------------------------------------------------------------------------
    def render(self, scene, g, pc, W, H, camera=abi.TRT_CAMERA_PINHOLE, want_hits=True):
        """Host buffers.  Returns (rgba (H,W,4), hits dict | None)."""
        rgba = np.empty((H, W, 4), np.float32)
        hits = abi.alloc_hits(W * H) if want_hits else None
        hs = abi.hits_struct(hits) if hits is not None else None
        self._check(self._L.trt_render(self._h, C.byref(g), C.byref(pc), C.byref(scene.c), W, H,
                                       camera, abi.ptr(rgba), C.byref(hs) if hs is not None else None))
        return rgba, hits

    def render_dev(self, scene, g, pc, W, H, rgba_ptr, rows=None, camera=abi.TRT_CAMERA_PINHOLE,
                   hit_ptrs=None, rendered_ptr=0, stream=0):
        """Device buffers, asynchronous on ``stream``; rows = (begin, end) of the band to render."""
        r0, r1 = (0, H) if rows is None else rows
        hs = None
        if hit_ptrs:
            hs = abi.hits_struct({k: (int(v) if v else None) for k, v in hit_ptrs.items()})
        self._check(self._L.trt_render_dev(self._h, C.byref(g), C.byref(pc), C.byref(scene.c), W, H,
                                           r0, r1, camera, C.c_void_p(int(rgba_ptr) or None),
                                           C.byref(hs) if hs is not None else None,
                                           C.c_void_p(int(rendered_ptr) or None),
                                           C.c_void_p(int(stream) or None)))

    def render_tiled_dev(self, scene, g, pc, W, H, tiling, rgba_ptr, camera=abi.TRT_CAMERA_PINHOLE,
                         hit_ptrs=None, rendered_ptr=0, stream=0):
        """Rows owned by ``tiling.part`` only (multi-GPU tiling, include/trt.h ``trt_tiling``)."""
        hs = None
        if hit_ptrs:
            hs = abi.hits_struct({k: (int(v) if v else None) for k, v in hit_ptrs.items()})
        self._check(self._L.trt_render_tiled_dev(self._h, C.byref(g), C.byref(pc), C.byref(scene.c), W, H,
                                                 C.byref(tiling), camera, C.c_void_p(int(rgba_ptr) or None),
                                                 C.byref(hs) if hs is not None else None,
                                                 C.c_void_p(int(rendered_ptr) or None),
                                                 C.c_void_p(int(stream) or None)))

    def render_batch_dev(self, scene, frames, W, H, tiling=None, camera=abi.TRT_CAMERA_PINHOLE, stream=0):
        """Up to TRT_MAX_BATCH consecutive frames of a frame loop in ONE pair of launches (trt_render_batch_dev).
        frames: sequence of (g, pc, rgba_ptr, hit_ptrs | None); tiling None = whole frames."""
        n = len(frames)
        arr = (abi.trt_frame * n)()
        keep = []
        for dst, (g, pc, rgba_ptr, hit_ptrs) in zip(arr, frames):
            dst.g, dst.pc = C.pointer(g), C.pointer(pc)
            dst.rgba_dev = int(rgba_ptr) or None
            if hit_ptrs:
                hs = abi.hits_struct({k: (int(v) if v else None) for k, v in hit_ptrs.items()})
                keep.append(hs)
                dst.first_hit_dev = C.pointer(hs)
        self._check(self._L.trt_render_batch_dev(self._h, arr, n, C.byref(scene.c), W, H,
                                                 C.byref(tiling) if tiling is not None else None, camera,
                                                 C.c_void_p(int(stream) or None)))

    def tiling_rows(self, tiling, H):
        return int(self._L.trt_tiling_rows(C.byref(tiling), H))

    def post_dev(self, rgba_ptr, n_pixels, f32_out_ptr=0, unorm8_out_ptr=0, stream=0):
        """Tonemap pass of post.frag (pow(c, 1/2.2)) on device buffers."""
        self._check(self._L.trt_post_dev(self._h, C.c_void_p(int(rgba_ptr) or None), int(n_pixels),
                                         C.c_void_p(int(f32_out_ptr) or None),
                                         C.c_void_p(int(unorm8_out_ptr) or None),
                                         C.c_void_p(int(stream) or None)))

    def splat_dev(self, points_ptr, n_points, view_proj, W, H, rgba_ptr, clear=(0.8, 0.8, 0.8, 1.0),
                  point_size=2.5, stream=0):
        """Point-cloud re-projection of ray_tracing__before_second (trt_splat_dev).  view_proj:
        4x4 numpy matrix in math convention (M[row, col]); clear colour as SEC/main.cpp:178."""
        vp = (C.c_float * 16)(*np.asarray(view_proj, np.float32).T.reshape(-1).tolist())
        cc = (C.c_float * 4)(*[float(v) for v in clear])
        self._check(self._L.trt_splat_dev(self._h, C.c_void_p(int(points_ptr) or None), int(n_points), vp, W, H,
                                          cc, float(point_size), C.c_void_p(int(rgba_ptr) or None),
                                          C.c_void_p(int(stream) or None)))

    def raytrace(self, scene, g, light, max_depth, clear_color, W, H, rgba_ptr, camera=0, rho=0.0,
                 **kw):
        """Mirror of ``HelloVulkan::raytrace(cmdBuf, clearColor)``: fills PushConstantRay from
        the light state (``light`` = dict position/intensity/type, the m_pcRaster fields) and
        the clear colour, then launches W×H (hello_vulkan.cpp:917-931)."""
        pc = abi.make_push(clear=clear_color, light_pos=light["position"],
                           light_intensity=light["intensity"], light_type=light["type"],
                           max_depth=max_depth, rho=rho)
        self.render_dev(scene, g, pc, W, H, rgba_ptr, camera=camera, **kw)
        return pc
