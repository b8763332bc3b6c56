"""Camera matrices and the BASELINE scenes — inputs at the drop-in boundary.

The reference fills ``GlobalUniforms`` in ``HelloVulkan::updateUniformBuffer``
(vk_raytracing_tutorial_KHR/ray_tracing_reflections/hello_vulkan.cpp:57-98) with
``nvmath::perspectiveVK(fov, aspect, 0.1, 1000)``, the camera manipulator's view matrix and
their inverses; nvmath is not in the reference tree, so these matrices are *inputs* to the
path.  This module builds equivalent matrices with numpy float64 and
rounds once to float32.
"""
import numpy as np

from . import abi


def look_at(eye, center, up=(0.0, 1.0, 0.0)):
    """Right-handed view matrix (world → camera, camera looks down -z)."""
    eye, center, up = (np.asarray(v, np.float64) for v in (eye, center, up))
    f = center - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[:3, 3] = -m[:3, :3] @ eye
    return m


def perspective_vk(fov_deg, aspect, near=0.1, far=1000.0):
    """Vulkan-convention perspective (y down, depth 0..1), the shape of nvmath::perspectiveVK."""
    t = np.tan(np.radians(fov_deg) * 0.5)
    m = np.zeros((4, 4))
    m[0, 0] = 1.0 / (aspect * t)
    m[1, 1] = -1.0 / t
    m[2, 2] = far / (near - far)
    m[2, 3] = (far * near) / (near - far)
    m[3, 2] = -1.0
    return m


def globals_for(eye, center, W, H, fov_deg=60.0, up=(0.0, 1.0, 0.0)):
    """``trt_globals`` as updateUniformBuffer would fill it (hello_vulkan.cpp:60-70)."""
    view = look_at(eye, center, up)
    proj = perspective_vk(fov_deg, W / float(H))
    return abi.make_globals(np.linalg.inv(view), np.linalg.inv(proj), proj @ view, center)


#: reflective material of the reference's mirror scene (ray_tracing_reflections/README.md:30-38):
#: illum 3, Ks 0.95; diffuse/ambient 0, shininess 32
MIRROR = dict(ambient=(0.0, 0.0, 0.0), diffuse=(0.0, 0.0, 0.0), specular=(0.95, 0.95, 0.95),
              shininess=32.0, illum=3)
#: a plain Phong material (illum 2) used by the parity tests to exercise diffuse + specular
PLASTIC = dict(ambient=(0.05, 0.05, 0.05), diffuse=(0.7, 0.2, 0.2), specular=(0.5, 0.5, 0.5),
               shininess=24.0, illum=2)
MATTE = dict(ambient=(0.1, 0.1, 0.1), diffuse=(0.2, 0.6, 0.3), specular=(0.0, 0.0, 0.0),
             shininess=1.0, illum=1)
FLAT = dict(ambient=(0.3, 0.3, 0.3), diffuse=(0.4, 0.4, 0.8), specular=(0.0, 0.0, 0.0),
            shininess=1.0, illum=0)


def single_torus_scene(center=(0.0, 0.0, 0.0), R=1.0, r=0.25, material=MIRROR):
    """BASELINE configs 1-3, 5: single torus R=1.0, r=0.25."""
    return abi.Scene([(center, R, r, 0)], [material])


def nested_tori_scene(center=(0.0, 0.0, 0.0)):
    """BASELINE config 4: 8 nested tori ("tokamak shells"), same centre, R=1,
    r = 0.05…0.40 step 0.05; the inner shells are Phong, the outer ones mirrors
    (SURVEY.md §8d)."""
    tori = [(center, 1.0, 0.05 * (i + 1), 0 if i < 4 else 1) for i in range(8)]
    return abi.Scene(tori, [PLASTIC, MIRROR])


def baseline_camera(W, H):
    """Pinhole camera of SURVEY.md §8d: eye (0,1.5,-4) → origin, up +y, fov 60°."""
    return globals_for((0.0, 1.5, -4.0), (0.0, 0.0, 0.0), W, H)


def baseline_push(max_depth):
    """Light (10,15,8), intensity 100, point light; clear colour (1,1,1,1)
    (REFL/hello_vulkan.h:74-80, REFL/main.cpp:212)."""
    return abi.make_push(max_depth=max_depth)


def toroidal_camera(W, H, eye=(0.0, 0.0, 0.0), center=(10.0, 0.0, 0.0)):
    """The reference's toroidal set-up: eye at the origin looking at (10,0,0)
    (ray_tracing__before/main.cpp:124); rho is a push constant."""
    return globals_for(eye, center, W, H)
