"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs, against the committed golden fixtures, and — at BASELINE sizes — through
size-independent properties.

Bars (BASELINE.json north_star): hit/miss classification and torus id BIT-EXACT; here the
geometric outputs (t, P, N) are also required bit-exact because both sides execute the
same correctly-rounded operation DAG; colours within 1e-5 relative (+1e-6 absolute) because
pow()/cos()/sin() come from different maths libraries (OCML vs glibc)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, seeded_rays
from toroidal_ray_tracing_amd import abi, camera

pytestmark = pytest.mark.gpu

GEOM = ("t", "px", "py", "pz", "nx", "ny", "nz")
COLOR_RTOL, COLOR_ATOL = 1e-5, 1e-6


@pytest.fixture(scope="module")
def tr():
    from toroidal_ray_tracing_amd.tracer import Tracer
    t = Tracer(0)
    yield t
    t.close()


QUERY_KEYS = ("primary_tests", "bounce_tests", "shadow_tests", "pixels")


def q(stats):
    """The query counts of a stats dict: what must equal the oracle's whatever the kernel variant.
    (traced_tests / solved_tests / evaluations describe the WORK a variant did: tiles answered by the
    classification are not traced, and the GPU's walk books its evaluations differently.)"""
    return {k: stats[k] for k in QUERY_KEYS}


def assert_hits_equal(a, b, what=""):
    np.testing.assert_array_equal(a["id"], b["id"], err_msg=what + " id")
    for k in GEOM:
        np.testing.assert_array_equal(a[k].view(np.uint32), b[k].view(np.uint32), err_msg=f"{what} {k} bits")


SCENES = {
    "single": lambda: camera.single_torus_scene(),
    "thin_offset": lambda: camera.single_torus_scene(center=(0.3, -0.2, 0.5), R=2.0, r=0.1),
    "fat": lambda: camera.single_torus_scene(R=1.0, r=0.9),
    "nested8": lambda: camera.nested_tori_scene(),
}


@pytest.mark.parametrize("scene", list(SCENES))
@pytest.mark.parametrize("precision", [abi.TRT_SOLVE_F32, abi.TRT_SOLVE_F64, abi.TRT_SOLVE_DK_F32, abi.TRT_SOLVE_DK_F64,
                                       abi.TRT_SOLVE_FERRARI_F32, abi.TRT_SOLVE_FERRARI_F64],
                         ids=["f32", "f64", "dk32", "dk64", "ferrari32", "ferrari64"])
def test_trace_bit_exact_vs_oracle(tr, oracle, scene, precision):
    sc = SCENES[scene]()
    o, d = seeded_rays(100_003, 1234, center=sc.tori_list()[0][0], box=5.0, reach=2.4)
    tr.set_solver(precision)
    try:
        got = tr.trace(sc, o, d)
    finally:
        tr.set_solver(abi.TRT_SOLVE_F32)
    want, _ = oracle.trace(sc, o, d, precision=precision, nthreads=8)
    assert np.isfinite(want["t"]).mean() > 0.05
    assert_hits_equal(got, want, scene)


@pytest.mark.parametrize("name", ["rays_single", "rays_thin_offset", "rays_nested"])
def test_trace_vs_golden_fp64_truth(tr, name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    tori = [((c[0], c[1], c[2]), c[3], c[4]) for c in z["tori"]]
    sc = abi.Scene([(c, R, r, 0) for c, R, r in tori], [camera.MIRROR])
    got = tr.trace(sc, z["o"], z["d"])
    hit_g, hit_t, rob = np.isfinite(got["t"]), np.isfinite(z["t"]), z["robust"]
    assert not np.any((hit_g != hit_t) & rob)
    both = hit_g & hit_t & rob
    np.testing.assert_array_equal(got["id"][both], z["id"][both])
    assert (np.abs(got["t"][both] - z["t"][both]) / np.maximum(1, z["t"][both])).max() < 1e-5


@pytest.mark.parametrize("precision", [abi.TRT_SOLVE_F32, abi.TRT_SOLVE_F64], ids=["f32", "f64"])
@pytest.mark.parametrize("family", ["equatorial", "meridional", "axial"])
def test_trace_vs_closed_form_families(tr, family, precision):
    """The HIP path against closed-form truth that involves no quartic solver at all (oracle/truth.py::
    closed_form_family: rays that meet the torus in circles): classification equal on every non-tangent ray,
    t within 1e-5 relative (2e-6 with the FP64 solve), normals within 2e-5·R/r — the north_star tolerance,
    checked against arithmetic that shares nothing with the kernels or the C oracle."""
    from oracle import truth
    for C, R, r in (((0.0, 0.0, 0.0), 1.0, 0.25), ((0.5, -0.25, 1.0), 2.0, 0.125)):
        o, d, t, N, ok = truth.closed_form_family(family, 100_000, 78, C=C, R=R, r=r)
        sc = camera.single_torus_scene(center=C, R=R, r=r)
        tr.set_solver(precision)
        try:
            got = tr.trace(sc, o, d)
        finally:
            tr.set_solver(abi.TRT_SOLVE_F32)
        hit_t, hit_g = np.isfinite(t), np.isfinite(got["t"])
        assert not np.any((hit_t != hit_g) & ok)
        both = hit_t & hit_g & ok
        assert both.sum() > 5000
        rel = np.abs(got["t"][both] - t[both]) / np.maximum(1.0, t[both])
        assert rel.max() < (2e-6 if precision == abi.TRT_SOLVE_F64 else 1e-5), rel.max()
        Ng = np.stack([got["nx"], got["ny"], got["nz"]], 1)[both]
        assert np.abs(Ng - N[both]).max() < 2e-5 * max(1.0, R / r)


def test_trace_edge_cases(tr, oracle):
    sc = camera.single_torus_scene()
    # empty input
    assert all(len(v) == 0 for v in tr.trace(sc, np.zeros((0, 3)), np.zeros((0, 3))).values())
    # ragged sizes around the wavefront / block size
    for n in (1, 63, 64, 65, 255, 257, 1000):
        o, d = seeded_rays(n, n)
        assert_hits_equal(tr.trace(sc, o, d), oracle.trace(sc, o, d)[0], f"n={n}")
    # degenerate rays: zero direction, NaN, inf, huge origin, non-unit direction, origin on surface
    o = np.float32([[0, 0, 0], [np.nan, 0, 0], [-5, 0, 0], [1e6, 0, 0], [-5, 0.1, 0.05], [-1.25, 0, 0],
                    [-5, 0, 0], [0, 5, 0], [1, 5, 0]])
    d = np.float32([[0, 0, 0], [1, 0, 0], [np.inf, 0, 0], [-1, 0, 0], [2.5, 0, 0], [1, 0, 0],
                    [1, 0, 0], [0, -1, 0], [0, -1, 0]])
    got, want = tr.trace(sc, o, d), oracle.trace(sc, o, d)[0]
    assert_hits_equal(got, want, "degenerate")
    assert got["id"][0] == -1 and got["id"][1] == -1 and np.isposinf(got["t"][0])
    assert got["t"][6] == 3.75 and got["id"][7] == -1 and got["t"][8] == 4.75
    # tmin/tmax window
    o, d = seeded_rays(5000, 5)
    assert_hits_equal(tr.trace(sc, o, d, 2.0, 4.0), oracle.trace(sc, o, d, 2.0, 4.0)[0], "window")


def test_trace_null_streams(tr):
    """Any output stream may be NULL (include/trt.h)."""
    import ctypes as C
    sc = camera.single_torus_scene()
    o, d = seeded_rays(1000, 9)
    oo, dd = np.ascontiguousarray(o.T), np.ascontiguousarray(d.T)
    rays = abi.rays_struct([oo[0], oo[1], oo[2], dd[0], dd[1], dd[2]], 1000)
    t = np.empty(1000, np.float32)
    hs = abi.hits_struct({"t": t})
    assert tr._L.trt_trace(tr._h, C.byref(rays), C.byref(sc.c), 0.001, 1e4, C.byref(hs)) == 0
    np.testing.assert_array_equal(t, tr.trace(sc, o, d)["t"])


def test_error_codes(tr):
    from toroidal_ray_tracing_amd.tracer import TrtError
    o, d = seeded_rays(10, 1)
    for bad in (abi.Scene([((0, 0, 0), 1.0, 1.5, 0)], [camera.MIRROR]),
                abi.Scene([((0, 0, 0), 1.0, 0.2, 2)], [camera.MIRROR])):
        with pytest.raises(TrtError) as e:
            tr.trace(bad, o, d)
        assert e.value.code == abi.TRT_E_SCENE
    with pytest.raises(TrtError) as e:
        tr.render(camera.single_torus_scene(), camera.baseline_camera(8, 8), camera.baseline_push(1), 0, 8)
    assert e.value.code == abi.TRT_E_INVALID
    with pytest.raises(TrtError):
        tr.set_render_variant("nope")
    import torch
    img = torch.zeros(8 * 8 * 4 + 4, device="cuda:0")
    with pytest.raises(TrtError) as e:   # float4 stores need a 16-byte aligned image
        tr.render_dev(camera.single_torus_scene(), camera.baseline_camera(8, 8), camera.baseline_push(1), 8, 8,
                      img[1:].data_ptr())
    assert e.value.code == abi.TRT_E_INVALID


RENDERS = {
    "mirror_d1": lambda W, H: (camera.single_torus_scene(), camera.baseline_camera(W, H), camera.baseline_push(1), 0),
    "mirror_d5": lambda W, H: (camera.single_torus_scene(), camera.baseline_camera(W, H), camera.baseline_push(5), 0),
    "mirror_d10": lambda W, H: (camera.single_torus_scene(), camera.baseline_camera(W, H), camera.baseline_push(10), 0),
    "plastic_dir": lambda W, H: (camera.single_torus_scene(material=camera.PLASTIC), camera.baseline_camera(W, H),
                                 abi.make_push(max_depth=3, light_type=1), 0),
    "matte": lambda W, H: (camera.single_torus_scene(material=camera.MATTE), camera.baseline_camera(W, H),
                           camera.baseline_push(4), 0),
    "flat_lowlight": lambda W, H: (camera.single_torus_scene(material=camera.FLAT), camera.baseline_camera(W, H),
                                   abi.make_push(max_depth=2, light_pos=(0, -50, 0)), 0),
    "nested_d5": lambda W, H: (camera.nested_tori_scene(), camera.baseline_camera(W, H), camera.baseline_push(5), 0),
    "toroidal_interior": lambda W, H: (
        camera.single_torus_scene(R=6.0, r=1.5, material=camera.PLASTIC), camera.toroidal_camera(W, H),
        abi.make_push(max_depth=5, rho=4.0), 1),
    "toroidal_survey": lambda W, H: (   # SURVEY §8d secondary run: torus moved to (10,0,0)
        camera.single_torus_scene(center=(10.0, 0.0, 0.0), R=3.0, r=1.0), camera.toroidal_camera(W, H),
        abi.make_push(max_depth=5, rho=4.0), 1),
    "toroidal_tilted": lambda W, H: (   # eye.y != center.y exercises theta (BEF rgen:45-53)
        camera.single_torus_scene(R=6.0, r=1.5, material=camera.MIRROR),
        camera.toroidal_camera(W, H, eye=(0.5, 0.4, -0.3), center=(4.0, -1.0, 7.0)),
        abi.make_push(max_depth=4, rho=3.0), 1),
}


def check_render(tr, oracle, sc, g, pc, W, H, cam, precision=abi.TRT_SOLVE_F32):
    rgba, hits = tr.render(sc, g, pc, W, H, cam)
    wr, wh, _, wstats = oracle.render(sc, g, pc, W, H, cam, precision=precision, nthreads=8)
    assert_hits_equal(hits, wh, "first-hit record")
    np.testing.assert_allclose(rgba, wr, rtol=COLOR_RTOL, atol=COLOR_ATOL)
    return rgba, hits, wstats


@pytest.mark.parametrize("variant", ["static", "persistent", "listed"])
@pytest.mark.parametrize("name", list(RENDERS))
def test_render_parity(tr, oracle, name, variant):
    W, H = 200, 136   # not multiples of the 8x8 tile
    sc, g, pc, cam = RENDERS[name](W, H)
    tr.set_render_variant(variant)
    tr.enable_stats(True)
    try:
        _, hits, wstats = check_render(tr, oracle, sc, g, pc, W, H, cam)
        st = tr.stats()
    finally:
        tr.enable_stats(False)
        tr.set_render_variant("listed")
    # the same queries were executed: bounce/shadow decisions are bit-exact too
    assert q(st) == q(wstats)


def _nests():
    P, M = camera.PLASTIC, camera.MIRROR
    return [camera.nested_tori_scene(),
            abi.Scene([((0, 0, 0), 1.0, 0.4, 1), ((0, 0, 0), 1.05, 0.3, 0), ((0, 0.05, 0), 0.97, 0.2, 1), ((0, 0, 0), 1.0, 0.1, 0),
                       ((2.5, 0, 0), 0.6, 0.2, 1), ((2.5, 0, 0), 0.6, 0.1, 0)], [P, M]),      # a nest beside a nest
            abi.Scene([((0, 0, 0), 2.0, 1.2, 1), ((0, 0, 0), 2.0, 0.5, 1), ((0, 0, 0), 2.0, 0.2, 0)], [P, M])]   # room for an eye between shells


@pytest.mark.parametrize("variant", ["static", "persistent", "listed"])
@pytest.mark.parametrize("scene", [0, 1, 2])
def test_enclosure_cull_scenes(tr, oracle, scene, variant):
    """T3's enclosure cull on the GPU against the oracle (which implements the same rule, and is itself checked against
    "every torus tested" in tests/test_oracle.py): nests seen from outside, from close by, from BETWEEN two shells (the
    eye inside the outer tube: nothing may be skipped for primary rays there) and from above the hole, both cameras — the
    toroidal one with its ray origins on a circle around the eye —, first-hit records bit for bit, query counts equal,
    and fewer tests executed than queries x tori wherever something is enclosed."""
    sc = _nests()[scene]
    W, H = 120, 88
    tr.set_render_variant(variant)
    tr.enable_stats(True)
    culled = 0
    try:
        for ei, eye in enumerate([(0.0, 1.5, -4.0), (0.0, 0.2, -2.9), (2.9, 0.0, 0.0) if scene == 2 else (1.1, 0.0, 0.0), (0.3, 2.5, 0.4)]):
            for cam in (abi.TRT_CAMERA_PINHOLE, abi.TRT_CAMERA_TOROIDAL):
                g = camera.globals_for(eye, (0.0, 0.0, 0.0) if cam == abi.TRT_CAMERA_PINHOLE else (3.0, 0.1, 0.5), W, H)
                pc = abi.make_push(max_depth=4, rho=0.3 if cam == abi.TRT_CAMERA_TOROIDAL else 0.0, light_type=ei % 2)
                prec = abi.TRT_SOLVE_F64 if (ei + scene) % 2 else abi.TRT_SOLVE_F32
                tr.set_solver(prec)
                _, _, wstats = check_render(tr, oracle, sc, g, pc, W, H, cam, prec)
                st = tr.stats()
                assert q(st) == q(wstats), (scene, ei, cam)
                if variant == "static":   # every pixel traced: the tests a lane executed are the oracle's
                    all_tests = sum(q(st)[k] for k in ("primary_tests", "bounce_tests", "shadow_tests"))
                    assert st["traced_tests"] == wstats["traced_tests"] <= all_tests
                    culled += all_tests - st["traced_tests"]
        assert variant != "static" or culled > 0
    finally:
        tr.set_solver(abi.TRT_SOLVE_F32)
        tr.enable_stats(False)
        tr.set_render_variant("listed")


@pytest.mark.parametrize("variant", ["static", "persistent", "listed"])
def test_render_fp64_nested(tr, oracle, variant):
    W, H = 160, 120
    sc, g, pc = camera.nested_tori_scene(), camera.baseline_camera(W, H), camera.baseline_push(5)
    tr.set_solver(abi.TRT_SOLVE_F64)
    tr.set_render_variant(variant)
    try:
        check_render(tr, oracle, sc, g, pc, W, H, 0, abi.TRT_SOLVE_F64)
    finally:
        tr.set_solver(abi.TRT_SOLVE_F32)
        tr.set_render_variant("listed")


@pytest.mark.parametrize("variant", ["static", "listed"])
@pytest.mark.parametrize("precision", [abi.TRT_SOLVE_DK_F32, abi.TRT_SOLVE_DK_F64, abi.TRT_SOLVE_FERRARI_F32,
                                       abi.TRT_SOLVE_FERRARI_F64], ids=["dk32", "dk64", "ferrari32", "ferrari64"])
@pytest.mark.parametrize("name", ["mirror_d5", "nested_d5", "toroidal_interior"])
def test_render_durand_kerner(tr, oracle, name, precision, variant):
    """The two solvers north_star names for T2 (Durand–Kerner, Ferrari) behind the same render
    path: bit-exact first-hit record and query counts against the oracle's restatement of each."""
    W, H = 136, 104
    sc, g, pc, cam = RENDERS[name](W, H)
    tr.set_solver(precision)
    tr.set_render_variant(variant)
    tr.enable_stats(True)
    try:
        _, _, wstats = check_render(tr, oracle, sc, g, pc, W, H, cam, precision)
        assert q(tr.stats()) == q(wstats)
    finally:
        tr.enable_stats(False)
        tr.set_solver(abi.TRT_SOLVE_F32)
        tr.set_render_variant("listed")


def test_durand_kerner_not_in_persistent_variant(tr):
    from toroidal_ray_tracing_amd.tracer import TrtError
    tr.set_solver(abi.TRT_SOLVE_DK_F32)
    tr.set_render_variant("persistent")
    try:
        with pytest.raises(TrtError) as e:
            tr.render(camera.single_torus_scene(), camera.baseline_camera(16, 16), camera.baseline_push(2), 16, 16)
        assert e.value.code == abi.TRT_E_INVALID
    finally:
        tr.set_solver(abi.TRT_SOLVE_F32)
        tr.set_render_variant("listed")


@pytest.mark.parametrize("name,cam,prec", [("render_pinhole_mirror", 0, abi.TRT_SOLVE_F32),
                                           ("render_pinhole_nested_f64", 0, abi.TRT_SOLVE_F64),
                                           ("render_toroidal_plastic", 1, abi.TRT_SOLVE_F32)])
def test_render_vs_golden(tr, name, cam, prec):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    W = H = 64
    pc = camera.baseline_push(5)
    if name == "render_pinhole_mirror":
        sc, g = camera.single_torus_scene(), camera.baseline_camera(W, H)
    elif name == "render_pinhole_nested_f64":   # the BASELINE config 4 shape: 8 nested tori, FP64 solve
        sc, g = camera.nested_tori_scene(), camera.baseline_camera(W, H)
    else:
        sc = camera.single_torus_scene(R=6.0, r=1.5, material=camera.PLASTIC)
        g = camera.toroidal_camera(W, H)
        pc.rho = 4.0
    tr.set_solver(prec)
    try:
        rgba, hits = tr.render(sc, g, pc, W, H, cam)
    finally:
        tr.set_solver(abi.TRT_SOLVE_F32)
    np.testing.assert_allclose(rgba, z["rgba"], rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(np.isfinite(hits["t"]), np.isfinite(z["hit_t"]))
    np.testing.assert_array_equal(hits["id"], z["hit_id"])
    m = np.isfinite(z["hit_t"])
    np.testing.assert_allclose(hits["t"][m], z["hit_t"][m], rtol=1e-5)


@pytest.mark.parametrize("variant", ["static", "persistent", "listed"])
def test_render_dev_rows_and_rendered_data(tr, oracle, variant):
    """Device-pointer entry point: row bands tile the frame; RenderedData AoS at x*H+y."""
    import torch
    W, H = 96, 72
    sc, g, pc = camera.single_torus_scene(R=6.0, r=1.5, material=camera.PLASTIC), camera.toroidal_camera(W, H), \
        abi.make_push(max_depth=3, rho=4.0)
    dev = torch.device("cuda:0")
    rgba = torch.zeros(H, W, 4, device=dev)
    rend = torch.zeros(W * H, 16, device=dev)
    hit_t = torch.full((H * W,), -1.0, device=dev)
    tr.set_render_variant(variant)
    try:
        s = torch.cuda.current_stream().cuda_stream
        for rows in ((0, 17), (17, 50), (50, 72)):
            tr.render_dev(sc, g, pc, W, H, rgba.data_ptr(), rows=rows, camera=1,
                          hit_ptrs={"t": hit_t.data_ptr()}, rendered_ptr=rend.data_ptr(), stream=s)
        torch.cuda.synchronize()
    finally:
        tr.set_render_variant("listed")
    wr, wh, wrend, _ = oracle.render(sc, g, pc, W, H, 1, want_rendered=True, nthreads=8)
    np.testing.assert_allclose(rgba.cpu().numpy(), wr, rtol=COLOR_RTOL, atol=COLOR_ATOL)
    np.testing.assert_array_equal(hit_t.cpu().numpy().view(np.uint32), wh["t"].view(np.uint32))
    got = rend.cpu().numpy()
    np.testing.assert_array_equal(got[:, [0, 1, 2, 3, 8, 9, 10, 11, 12, 13, 14, 15]].view(np.uint32),
                                  wrend[:, [0, 1, 2, 3, 8, 9, 10, 11, 12, 13, 14, 15]].view(np.uint32))
    np.testing.assert_allclose(got[:, 4:8], wrend[:, 4:8], rtol=COLOR_RTOL, atol=COLOR_ATOL)
    # a band leaves the other rows untouched
    rgba.zero_()
    tr.render_dev(sc, g, pc, W, H, rgba.data_ptr(), rows=(10, 20), camera=1, stream=s)
    torch.cuda.synchronize()
    out = rgba.cpu().numpy()
    assert np.all(out[:10] == 0) and np.all(out[20:] == 0) and np.all(out[10:20, :, 3] == 1)


@pytest.mark.parametrize("variant", ["static", "persistent", "listed"])
def test_full_size_properties(tr, variant):
    """BASELINE config 3 (4096², maxDepth 5 = 4 bounces): properties that need no oracle.
    Left-right mirror symmetry of the scene is NOT used (the light breaks it); instead:
    every recorded hit point lies on the torus, normals are unit and point against the ray,
    t matches |P - eye|, misses carry the clear colour, and the static and persistent
    kernels agree bit for bit on the geometric record."""
    import torch
    W = H = 4096
    sc, g, pc = camera.single_torus_scene(), camera.baseline_camera(W, H), camera.baseline_push(5)
    dev = torch.device("cuda:0")
    rgba = torch.empty(H, W, 4, device=dev)
    hb = {k: torch.empty(H * W, device=dev) for k in GEOM}
    hid = torch.empty(H * W, dtype=torch.int32, device=dev)
    tr.set_render_variant(variant)
    try:
        tr.render_dev(sc, g, pc, W, H, rgba.data_ptr(),
                      hit_ptrs={**{k: v.data_ptr() for k, v in hb.items()}, "id": hid.data_ptr()},
                      stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    finally:
        tr.set_render_variant("listed")
    hit = hid >= 0
    assert 0.05 < hit.float().mean().item() < 0.5
    P = torch.stack([hb["px"], hb["py"], hb["pz"]], 1)[hit].double()
    N = torch.stack([hb["nx"], hb["ny"], hb["nz"]], 1)[hit].double()
    rho = torch.hypot(P[:, 0], P[:, 2])
    gval = (rho - 1.0) ** 2 + P[:, 1] ** 2 - 0.25 ** 2
    assert gval.abs().max().item() < 5e-6            # on the surface (|g| ~ 2 r dist)
    assert (N.norm(dim=1) - 1).abs().max().item() < 3e-7
    eye = torch.tensor([0.0, 1.5, -4.0], dtype=torch.float64, device=dev)
    D = P - eye
    assert ((D * N).sum(1) <= 1e-6).all()            # first hit faces the camera
    assert (D.norm(dim=1) - hb["t"][hit].double()).abs().max().item() < 2e-5
    miss = ~hit
    assert torch.isinf(hb["t"][miss]).all() and (hb["px"][miss] == 0).all()
    img = rgba.view(-1, 4)
    assert (img[miss][:, :3] == 0.8).all() and (img[:, 3] == 1).all()
    assert torch.isfinite(img).all() and (img[:, :3] >= 0).all()
    # checksum pinned across kernel variants (bit-exact geometry)
    csum = {k: int(v.view(torch.int32).long().sum().item()) for k, v in hb.items()}
    prev = test_full_size_properties.__dict__.setdefault("csum", csum)
    assert prev == csum


@pytest.mark.parametrize("variant", ["static", "persistent", "listed"])
def test_config4_full_size_properties(tr, variant):
    """BASELINE config 4 at full size (8 nested tori r = 0.05…0.40 around R = 1, 4096², maxDepth 5, FP64 root
    solve, FP32 I/O) through oracle-free properties: the first hit is always the OUTERMOST shell seen from outside
    (id 7), every hit point lies on the shell its id names, normals are unit and face the ray, t = |P - eye|,
    misses carry the miss record, and the three kernel variants agree bit for bit (checksum of every stream)."""
    import torch
    W = H = 4096
    sc, g, pc = camera.nested_tori_scene(), camera.baseline_camera(W, H), camera.baseline_push(5)
    dev = torch.device("cuda:0")
    rgba = torch.empty(H, W, 4, device=dev)
    hb = {k: torch.empty(H * W, device=dev) for k in GEOM}
    hid = torch.empty(H * W, dtype=torch.int32, device=dev)
    tr.set_render_variant(variant)
    tr.set_solver(abi.TRT_SOLVE_F64)
    tr.enable_stats(True)
    try:
        tr.render_dev(sc, g, pc, W, H, rgba.data_ptr(),
                      hit_ptrs={**{k: v.data_ptr() for k, v in hb.items()}, "id": hid.data_ptr()},
                      stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        st = tr.stats()
    finally:
        tr.enable_stats(False)
        tr.set_solver(abi.TRT_SOLVE_F32)
        tr.set_render_variant("listed")
    hit = hid >= 0
    assert 0.08 < hit.float().mean().item() < 0.5
    assert (hid[hit] == 7).all()                      # the camera is outside: the r = 0.40 shell hides the others
    P = torch.stack([hb["px"], hb["py"], hb["pz"]], 1)[hit].double()
    N = torch.stack([hb["nx"], hb["ny"], hb["nz"]], 1)[hit].double()
    rho = torch.hypot(P[:, 0], P[:, 2])
    r_id = 0.05 * (hid[hit].double() + 1.0)
    gval = (rho - 1.0) ** 2 + P[:, 1] ** 2 - (r_id.float().double()) ** 2
    assert gval.abs().max().item() < 5e-6             # FP32 P = o + t·d: |g| ≈ 2 r · 1e-6
    assert (N.norm(dim=1) - 1).abs().max().item() < 3e-7
    eye = torch.tensor([0.0, 1.5, -4.0], dtype=torch.float64, device=dev)
    D = P - eye
    assert ((D * N).sum(1) <= 1e-6).all()
    assert (D.norm(dim=1) - hb["t"][hit].double()).abs().max().item() < 2e-5
    miss = ~hit
    assert torch.isinf(hb["t"][miss]).all() and (hb["px"][miss] == 0).all() and (hb["nz"][miss] == 0).all()
    img = rgba.view(-1, 4)
    assert (img[miss][:, :3] == 0.8).all() and (img[:, 3] == 1).all() and torch.isfinite(img).all()
    assert st["primary_tests"] == 8 * W * H and st["pixels"] == W * H and st["bounce_tests"] > 0 and st["shadow_tests"] > 0
    csum = {k: int(v.view(torch.int32).long().sum().item()) for k, v in hb.items()}
    csum["rgba"] = int(rgba.view(torch.int32).long().sum().item()) if variant != "x" else 0
    csum["queries"] = (st["primary_tests"], st["bounce_tests"], st["shadow_tests"])
    prev = test_config4_full_size_properties.__dict__.setdefault("csum", csum)
    assert prev == csum


def test_config5_shape_tiled_on_one_gpu(tr):
    """BASELINE config 5's shape on ONE GPU: 8192², maxDepth 5, the 8 parts of `trt_tiling` (groups of 8 rows and the
    default groups of 64 rows) rendered one after the other through trt_render_tiled_dev into compact buffers —
    what the 8 ranks do.  The parts, put at their rows (the in-place gather of TiledFrame), equal the full-frame
    render bit for bit; plus the oracle-free properties of test_full_size_properties on the assembled frame."""
    import torch
    from toroidal_ray_tracing_amd import distributed as trtd
    W = H = 8192
    parts = 8
    sc, g, pc = camera.single_torus_scene(), camera.baseline_camera(W, H), camera.baseline_push(5)
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    full = torch.empty(H, W, 4, device=dev)
    t_full = torch.empty(H * W, device=dev)
    id_full = torch.empty(H * W, dtype=torch.int32, device=dev)
    tr.enable_stats(True)
    try:
        tr.render_dev(sc, g, pc, W, H, full.data_ptr(), hit_ptrs={"t": t_full.data_ptr(), "id": id_full.data_ptr()}, stream=s)
        torch.cuda.synchronize()
        st_full = tr.stats()
        for group in (8, trtd.default_group_rows(H, parts), trtd.default_group_rows(H, parts, 1)):
            assert H % (group * parts) == 0
            asm = torch.zeros(H, W, 4, device=dev)          # the frame as the in-place gathers assemble it
            t_asm = torch.zeros(H, W, device=dev)
            local = torch.empty(H // parts, W, 4, device=dev)
            t_loc = torch.empty(H // parts, W, device=dev)
            tot = {k: 0 for k in QUERY_KEYS}
            span = group * parts
            for p in range(parts):
                tiling = abi.trt_tiling(group, parts, p, 1)
                assert tr.tiling_rows(tiling, H) == H // parts
                tr.render_tiled_dev(sc, g, pc, W, H, tiling, local.data_ptr(), hit_ptrs={"t": t_loc.data_ptr()}, stream=s)
                torch.cuda.synchronize()
                for k, v in q(tr.stats()).items():
                    tot[k] += v
                # all_gather_into_tensor(frame[c·span:(c+1)·span], local[c·group:(c+1)·group]) puts rank p's slice at offset p·group
                asm.view(H // span, parts, group, W, 4)[:, p] = local.view(H // span, group, W, 4)
                t_asm.view(H // span, parts, group, W)[:, p] = t_loc.view(H // span, group, W)
            assert torch.equal(asm.view(torch.int32), full.view(torch.int32)), group
            assert torch.equal(t_asm.view(-1).view(torch.int32), t_full.view(torch.int32)), group
            assert tot == q(st_full), group
            del asm, t_asm, local, t_loc
    finally:
        tr.enable_stats(False)
    hit = id_full >= 0
    assert 0.05 < hit.float().mean().item() < 0.5 and st_full["primary_tests"] == W * H
    img = full.view(-1, 4)
    assert (img[~hit][:, :3] == 0.8).all() and (img[:, 3] == 1).all() and torch.isfinite(img).all()
    assert torch.isinf(t_full[~hit]).all() and (t_full[hit] > 2.0).all() and (t_full[hit] < 7.0).all()


def test_rccl_collective_path_world_1(tr):
    """The RCCL leg of TiledFrame on the one GPU there is: a world_size-1 `nccl` process group with the collective
    FORCED (an all_gather_into_tensor over one rank is a device copy issued by RCCL), both gather modes, the two-deep
    pipeline and finish().  Multi-rank correctness of the same class is covered on CPU over gloo
    (tests/test_distributed.py); this one proves that the calls bench.py makes with N > 1 are accepted by RCCL on
    device buffers of the shapes and dtypes used (contiguous row slices of the frame, fp32 and uint8)."""
    import torch
    import torch.distributed as dist
    from toroidal_ray_tracing_amd import distributed as trtd
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    dev = torch.device("cuda:0")
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        W = H = 512
        sc, g = camera.single_torus_scene(), camera.baseline_camera(W, H)
        want = torch.empty(H, W, 4, device=dev)
        stream = torch.cuda.current_stream()
        tr.render_dev(sc, g, camera.baseline_push(3), W, H, want.data_ptr(), stream=stream.cuda_stream)
        for mode in ("fp32", "rgba8"):
            frame = trtd.TiledFrame(tr, W, H, 1, 0, dev, gather=mode, force_collective=True)
            assert frame.gather and "all_gather_into_tensor" in frame.describe()
            for depth in (1, 2, 3):
                frame.render(sc, g, camera.baseline_push(depth), abi.TRT_CAMERA_PINHOLE, stream)
            full = frame.finish()
            torch.cuda.synchronize()
            if mode == "fp32":
                assert torch.equal(full.view(torch.int32), want.view(torch.int32))
            else:
                w8 = torch.empty(H, W, 4, dtype=torch.uint8, device=dev)
                tr.post_dev(want.data_ptr(), W * H, 0, w8.data_ptr(), stream=stream.cuda_stream)
                torch.cuda.synchronize()
                assert full.dtype == torch.uint8 and torch.equal(full, w8)
        # one gather per batch of three frames (what bench.py --gpus N does per step): frames 2 and 5 are gathered,
        # the last GATHERED frame is the one handed out
        frame = trtd.TiledFrame(tr, W, H, 1, 0, dev, gather="fp32", force_collective=True, gather_every=3)
        for depth in (1, 2, 5, 1, 2, 3, 1):
            frame.render(sc, g, camera.baseline_push(depth), abi.TRT_CAMERA_PINHOLE, stream)
        full = frame.finish()
        torch.cuda.synchronize()
        assert torch.equal(full.view(torch.int32), want.view(torch.int32))
        # three frames in flight on three HIP streams (three contexts, three output sets), one gather every 4 frames:
        # frames 3 and 7 are gathered (the second one is the maxDepth-3 frame), the caller's stream sees them after finish()
        from toroidal_ray_tracing_amd.tracer import Tracer
        extra = [Tracer(0), Tracer(0)]
        try:
            frame = trtd.TiledFrame([tr] + extra, W, H, 1, 0, dev, gather="fp32", force_collective=True, gather_every=4,
                                    want_hits=("t", "id"))
            for depth in (1, 2, 5, 2, 1, 2, 5, 3, 1, 2):
                frame.render(sc, g, camera.baseline_push(depth), abi.TRT_CAMERA_PINHOLE, stream)
            full = frame.finish()
            last3 = [frame.locals[k].clone() for k in range(3)]   # frames 9, 7, 8 (maxDepth 2, 3, 1) on sets 0, 1, 2
            torch.cuda.synchronize()
            assert torch.equal(full.view(torch.int32), want.view(torch.int32))
            assert torch.equal(last3[1].view(torch.int32), want.view(torch.int32))
            assert not torch.equal(last3[0].view(torch.int32), want.view(torch.int32))
            # the same three streams with a whole STEP captured into one hipGraph (what bench.py --gpus N replays): 6 frames
            # over 3 streams and 6 output sets per replay, the gather of the step's last frame issued eagerly behind it;
            # two replays, an eager frame loop in between, then the tiled part of an 8-rank job the same way (no gather)
            pc3 = camera.baseline_push(3)
            frame = trtd.TiledFrame([tr] + extra, W, H, 1, 0, dev, gather="fp32", force_collective=True, gather_every=6,
                                    want_hits=("t",), output_sets=6)
            assert frame.n_sets == 6
            for _ in range(6):
                frame.render(sc, g, pc3, abi.TRT_CAMERA_PINHOLE, stream)
            frame.restart()
            for buf in frame.locals:
                buf.fill_(-7.0)
            frame.capture_step(sc, g, pc3, abi.TRT_CAMERA_PINHOLE, stream, 6)
            frame.step(stream)
            frame.step(stream)
            full = frame.finish()
            torch.cuda.synchronize()
            assert torch.equal(full.view(torch.int32), want.view(torch.int32))
            for buf in frame.locals:
                assert torch.equal(buf.view(torch.int32), want.view(torch.int32))
            with pytest.raises(ValueError):
                frame.capture_step(sc, g, pc3, abi.TRT_CAMERA_PINHOLE, stream, 4)   # not whole rounds of the six output sets
            til = trtd.TiledFrame([tr] + extra, W, H, 8, 3, dev, gather="none", group_rows=8)
            for _ in range(3):
                til.render(sc, g, pc3, abi.TRT_CAMERA_PINHOLE, stream)
            til.restart()
            eager = [b.clone() for b in til.locals]
            for b in til.locals:
                b.zero_()
            til.capture_step(sc, g, pc3, abi.TRT_CAMERA_PINHOLE, stream, 9)
            til.step(stream)
            til.finish()
            torch.cuda.synchronize()
            rows = torch.tensor(trtd.owned_rows(H, 8, 8, 3), device=dev)
            for b, e in zip(til.locals, eager):
                assert torch.equal(b.view(torch.int32), e.view(torch.int32)) and torch.equal(b.view(torch.int32), want[rows].view(torch.int32))
        finally:
            for t in extra:
                t.close()
    finally:
        dist.destroy_process_group()


def test_render_batch_equals_single_frames(tr):
    """trt_render_batch_dev — up to 8 frames of a frame loop in ONE pair of launches — against the same frames rendered one
    by one: rgba, first-hit streams and the query counts bit for bit / equal.  Whole frames of a ragged size with a
    camera that moves from frame to frame, the tiled part of a 4-rank job, eight nested tori with the FP64 solve (cost
    feedback live: the batch is rendered twice), the toroidal camera with the per-tile classification and a rho sweep;
    and what a batch refuses."""
    import torch
    from toroidal_ray_tracing_amd.tracer import TrtError
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    keys = ("t", "px", "py", "pz", "nx", "ny", "nz", "id")

    def bufs(rows, W):
        rgba = torch.full((rows, W, 4), -5.0, device=dev)
        h = {k: torch.full((rows * W,), -5.0, device=dev) for k in keys[:-1]}
        h["id"] = torch.full((rows * W,), -5, dtype=torch.int32, device=dev)
        return rgba, h

    def same(a, b):
        return all(torch.equal(x.view(torch.int32), y.view(torch.int32)) for x, y in zip([a[0]] + [a[1][k] for k in keys], [b[0]] + [b[1][k] for k in keys]))

    def check(sc, frames, W, H, tiling=None, camera_model=0, repeat=1):
        rows = tr.tiling_rows(tiling, H) if tiling is not None else H
        single, want_stats = [], {k: 0 for k in abi.STAT_FIELDS}
        tr.enable_stats(True)
        for g, pc in frames:
            o = bufs(rows, W)
            hp = {k: v.data_ptr() for k, v in o[1].items()}
            if tiling is None:
                tr.render_dev(sc, g, pc, W, H, o[0].data_ptr(), camera=camera_model, hit_ptrs=hp, stream=s)
            else:
                tr.render_tiled_dev(sc, g, pc, W, H, tiling, o[0].data_ptr(), camera=camera_model, hit_ptrs=hp, stream=s)
            st = tr.stats()
            for k in abi.STAT_FIELDS:
                want_stats[k] += st[k]
            single.append(o)
        outs = [bufs(rows, W) for _ in frames]
        fl = [(g, pc, o[0].data_ptr(), {k: v.data_ptr() for k, v in o[1].items()}) for (g, pc), o in zip(frames, outs)]
        tr.render_batch_dev(sc, fl, W, H, tiling, camera=camera_model, stream=s)
        assert tr.stats() == want_stats
        tr.enable_stats(False)
        for _ in range(repeat):   # uncounted: the instantiations the timed loops run (cost feedback for scenes of >= 2 tori)
            for o in outs:
                o[0].fill_(-5.0)
            tr.render_batch_dev(sc, fl, W, H, tiling, camera=camera_model, stream=s)
            torch.cuda.synchronize()
            for i, (a, b) in enumerate(zip(outs, single)):
                assert same(a, b), i

    W, H = 200, 136
    sc1 = camera.single_torus_scene()
    moving = [(camera.globals_for((0.3 * i, 1.5 - 0.2 * i, -4.0 + 0.3 * i), (0.0, 0.0, 0.0), W, H), camera.baseline_push(1 + i % 5)) for i in range(5)]
    check(sc1, moving, W, H)
    check(sc1, moving[:1], W, H)                                         # a batch of one = the single-frame path
    W = H = 256
    til = abi.trt_tiling(8, 4, 1, 1)
    frames = [(camera.baseline_camera(W, H), camera.baseline_push(d)) for d in (5, 1, 3, 5, 2, 4, 5, 5)]
    check(sc1, frames, W, H, tiling=til)
    check(sc1, frames[:3], W, H, tiling=abi.trt_tiling(8, 2, 0, 1))
    tr.set_solver(abi.TRT_SOLVE_F64)
    try:
        check(camera.nested_tori_scene(), frames, W, H, repeat=2)
        check(camera.nested_tori_scene(), frames[:4], W, H, tiling=til, repeat=2)
    finally:
        tr.set_solver(abi.TRT_SOLVE_F32)
    # toroidal camera (per-tile classification), a rho sweep as in BEF/main.cpp:236-258: same eye and centre, other rho
    Wt, Ht = 256, 128
    sct, gt = camera.single_torus_scene(R=6.0, r=1.5, material=camera.PLASTIC), camera.toroidal_camera(Wt, Ht)
    sweep = []
    for i in range(4):
        pc = camera.baseline_push(3)
        pc.rho = 3.0 + 0.5 * i
        sweep.append((gt, pc))
    check(sct, sweep, Wt, Ht, camera_model=1)
    # refusals: frames whose toroidal tables differ, too many / no frames, another variant or solver
    o = bufs(Ht, Wt)
    other = camera.toroidal_camera(Wt, Ht, center=(10.0, 0.0, 3.0))
    with pytest.raises(TrtError) as e:
        tr.render_batch_dev(sct, [(gt, sweep[0][1], o[0].data_ptr(), None), (other, sweep[0][1], o[0].data_ptr(), None)], Wt, Ht, camera=1, stream=s)
    assert e.value.code == abi.TRT_E_INVALID and "one by one" in str(e.value)
    tr.render_dev(sct, gt, sweep[0][1], Wt, Ht, o[0].data_ptr(), camera=1, stream=s)   # the ctx still renders
    one = (moving[0][0], moving[0][1], o[0].data_ptr(), None)
    for bad in ([], [one] * 9):
        with pytest.raises(TrtError) as e:
            tr.render_batch_dev(sc1, bad, 64, 64, stream=s)
        assert e.value.code == abi.TRT_E_INVALID
    tr.set_render_variant("persistent")
    try:
        with pytest.raises(TrtError):
            tr.render_batch_dev(sc1, [one, one], 64, 64, stream=s)
        tr.render_batch_dev(sc1, [one], 64, 64, stream=s)                # one frame: every variant
    finally:
        tr.set_render_variant("listed")
    tr.set_solver(abi.TRT_SOLVE_FERRARI_F32)
    try:
        with pytest.raises(TrtError):
            tr.render_batch_dev(sc1, [one, one], 64, 64, stream=s)
    finally:
        tr.set_solver(abi.TRT_SOLVE_F32)
    torch.cuda.synchronize()


def test_bench_line_has_what_the_driver_parses():
    """`python bench.py` as the driver runs it (N = 1, small K, no CPU baseline / secondary passes to keep it short): ONE
    JSON line with the contract's keys, `roofline` priced on the pass into four alternating output sets with the
    single-image figure beside it, and — with the CPU baseline enabled on a small frame — a `cpu_baseline` object."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(*extra):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1", "--frames-per-step", "2", *extra],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=root)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, p.stdout[-2000:]
        return json.loads(lines[0])

    j = run("--no-cpu-baseline", "--no-secondary")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 1 and j["warmup"] == 1 and j["value"] > 2.0e9 and j["vs_baseline"] is None
    assert "BASELINE config 3" in j["config"]["workload"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] in ("hbm", "valu") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.05 < r["frac"] < 1.0
    assert r["output_sets"] == 4 and r["frac_cached"] is not None and r["kernel_ms_cached"] > 0
    assert abs(r["achieved"] - 44 * 4096 * 4096 / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert "cpu_baseline" not in j
    # CPU baseline leg on a frame small enough for a test (the oracle renders whole frames of the workload)
    j = run("--no-secondary", "--size", "512")
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["unit"] == j["unit"]


def test_trace_dev_full_size_matches_render(tr):
    """BASELINE config 2 shape (2048² primary rays, 0 bounces) through trace_dev: feeding the
    primary rays exported by the render (RenderedData.rayOrigin/rayDir) back into trace()
    reproduces the render's first-hit record bit for bit."""
    import torch
    W = H = 2048
    sc, g, pc = camera.single_torus_scene(), camera.baseline_camera(W, H), camera.baseline_push(1)
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    rend = torch.empty(W * H, 16, device=dev)
    t_r = torch.empty(H * W, device=dev)
    nx_r = torch.empty(H * W, device=dev)
    tr.render_dev(sc, g, pc, W, H, 0, hit_ptrs={"t": t_r.data_ptr(), "nx": nx_r.data_ptr()},
                  rendered_ptr=rend.data_ptr(), stream=s)
    r = rend.view(W, H, 16).permute(1, 0, 2).reshape(-1, 16)   # x*H+y -> y*W+x
    rays = [r[:, 8 + k].contiguous() for k in range(3)] + [r[:, 12 + k].contiguous() for k in range(3)]
    t_t = torch.empty(H * W, device=dev)
    nx_t = torch.empty(H * W, device=dev)
    tr.trace_dev(sc, [a.data_ptr() for a in rays], W * H, {"t": t_t.data_ptr(), "nx": nx_t.data_ptr()}, stream=s)
    torch.cuda.synchronize()
    assert torch.equal(t_r.view(torch.int32), t_t.view(torch.int32))
    assert torch.equal(nx_r.view(torch.int32), nx_t.view(torch.int32))
    assert 0.05 < torch.isfinite(t_t).float().mean().item() < 0.5


@pytest.mark.parametrize("variant", ["static", "persistent", "listed"])
@pytest.mark.parametrize("parts,group", [(2, 8), (4, 8), (3, 4), (8, 8)])
@pytest.mark.parametrize("scene_cam", ["pinhole", "toroidal"])
def test_tiled_render_matches_full(tr, variant, parts, group, scene_cam):
    """trt_render_tiled_dev: every part renders its interleaved row groups into a compact
    buffer; stacking + de-interleaving the parts reproduces the full-frame render bit for bit
    (this is what each rank does before the RCCL all-gather).  The toroidal case runs the
    per-tile classification level on interleaved rows."""
    import torch
    from toroidal_ray_tracing_amd import distributed as trtd
    W, H = 120, parts * group * 5
    if scene_cam == "pinhole":
        sc, g, pc, cam = camera.single_torus_scene(), camera.baseline_camera(W, H), camera.baseline_push(5), 0
    else:
        sc, g, pc, cam = RENDERS["toroidal_interior"](W, H)
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    tr.set_render_variant(variant)
    try:
        full = torch.zeros(H, W, 4, device=dev)
        t_full = torch.zeros(H * W, device=dev)
        tr.render_dev(sc, g, pc, W, H, full.data_ptr(), camera=cam, hit_ptrs={"t": t_full.data_ptr()}, stream=s)
        gathered = torch.zeros(parts, H // parts, W, 4, device=dev)
        t_parts = torch.zeros(parts, (H // parts) * W, device=dev)
        for p in range(parts):
            tiling = abi.trt_tiling(group, parts, p, 1)
            assert tr.tiling_rows(tiling, H) == H // parts
            tr.render_tiled_dev(sc, g, pc, W, H, tiling, gathered[p].data_ptr(), camera=cam,
                                hit_ptrs={"t": t_parts[p].data_ptr()}, stream=s)
        # non-compact: parts write straight into one full-frame buffer — and into one RenderedData buffer (x*H + y:
        # always full-frame; with groups that are not whole 8-row tile bands the tiles of a part are not contiguous rows)
        direct = torch.zeros(H, W, 4, device=dev)
        rd_parts = torch.zeros(W * H, 16, device=dev)
        for p in range(parts):
            tr.render_tiled_dev(sc, g, pc, W, H, abi.trt_tiling(group, parts, p, 0), direct.data_ptr(), camera=cam,
                                rendered_ptr=rd_parts.data_ptr(), stream=s)
        rd_full = torch.zeros(W * H, 16, device=dev)
        tr.render_dev(sc, g, pc, W, H, 0, camera=cam, rendered_ptr=rd_full.data_ptr(), stream=s)
        torch.cuda.synchronize()
        assert torch.equal(rd_parts.view(torch.int32), rd_full.view(torch.int32))
    finally:
        tr.set_render_variant("listed")
    assert torch.equal(trtd.deinterleave(gathered, H, W, group, parts), full)
    assert torch.equal(direct, full)
    tt = trtd.deinterleave(t_parts.view(parts, H // parts, W, 1), H, W, group, parts, channels=1).reshape(-1)
    assert torch.equal(tt.view(torch.int32), t_full.view(torch.int32))


CAMERAS = [
    # eye, center, fov, W, H  — the tile classification must stay conservative everywhere
    ((0.0, 1.5, -4.0), (0.0, 0.0, 0.0), 60.0, 200, 136),     # baseline
    ((0.0, 0.05, 0.0), (1.0, 0.0, 0.2), 90.0, 168, 104),     # inside the hole, in the torus plane
    ((0.9, 0.1, 0.3), (-1.0, 0.0, 0.0), 100.0, 136, 136),    # inside the bounding sphere, next to the tube
    ((0.0, 3.0, 0.01), (0.0, 0.0, 0.0), 45.0, 128, 160),     # straight down the axis (a = 0 rays)
    ((6.0, 0.0, 0.0), (0.0, 0.0, 0.0), 25.0, 264, 72),       # edge-on, narrow fov (dy = 0 rows)
    ((-2.0, 0.7, 2.5), (3.0, 0.0, -2.0), 120.0, 152, 88),    # wide angle, torus off-centre
    ((0.0, 1.5, -4.0), (0.0, 1.5, -9.0), 60.0, 96, 64),      # torus behind the camera
    ((0.3, 0.26, 1.0), (0.0, 0.0, 0.9), 70.0, 120, 120),     # skimming the top of the tube
    ((40.0, 12.0, -25.0), (0.0, 0.0, 0.0), 5.0, 232, 168),   # far away, tiny fov
]


@pytest.mark.parametrize("variant,fine", [("persistent", 0), ("persistent", 1), ("listed", 0), ("listed", 1)])
@pytest.mark.parametrize("cam", range(len(CAMERAS)))
@pytest.mark.parametrize("scene", ["single", "nested8", "thin_offset"])
def test_tile_classification_is_conservative(tr, oracle, cam, variant, fine, scene):
    """Frames from cameras around, inside and behind the tori: the CLEAR/LIVE classification —
    the macro-tile level and the finer per-tile level with the distance-function march (default
    for the toroidal camera only, forced here) — may never change a pixel: first-hit records
    bit-exact, query counts identical."""
    eye, center, fov, W, H = CAMERAS[cam]
    sc = SCENES[scene]()
    g = camera.globals_for(eye, center, W, H, fov_deg=fov)
    pc = camera.baseline_push(4)
    tr.set_render_variant(variant)
    tr.enable_stats(True)
    tr.set_classification(fine)
    try:
        _, _, wstats = check_render(tr, oracle, sc, g, pc, W, H, 0)
        assert q(tr.stats()) == q(wstats)
    finally:
        tr.set_classification(abi.TRT_CLASSIFY_AUTO)
        tr.enable_stats(False)
        tr.set_render_variant("listed")


@pytest.mark.parametrize("variant", ["static", "persistent", "listed"])
@pytest.mark.parametrize("size", [(1, 1), (7, 3), (8, 8), (33, 9), (5, 64), (129, 17)])
def test_render_ragged_sizes(tr, oracle, size, variant):
    """Frames smaller than a tile, than a macro tile, and not multiples of either, both cameras."""
    W, H = size
    tr.set_render_variant(variant)
    try:
        for name in ("mirror_d5", "toroidal_interior"):
            sc, g, pc, cam = RENDERS[name](W, H)
            check_render(tr, oracle, sc, g, pc, W, H, cam)
    finally:
        tr.set_render_variant("listed")


@pytest.mark.parametrize("fine", [0, 1])
@pytest.mark.parametrize("name", ["toroidal_interior", "toroidal_survey", "toroidal_tilted"])
def test_toroidal_classification_levels(tr, oracle, name, fine):
    """Both classification levels under the toroidal camera (varying ray origins)."""
    W, H = 200, 136
    sc, g, pc, cam = RENDERS[name](W, H)
    tr.set_classification(fine)
    tr.enable_stats(True)
    try:
        _, _, wstats = check_render(tr, oracle, sc, g, pc, W, H, cam)
        assert q(tr.stats()) == q(wstats)
    finally:
        tr.set_classification(abi.TRT_CLASSIFY_AUTO)
        tr.enable_stats(False)


def random_case(seed):
    """A seeded random scene + camera + push constants (both camera models, 1-8 tori that may
    intersect, every material kind, both light types)."""
    rng = np.random.default_rng(1000 + seed)
    mats = [camera.MIRROR, camera.PLASTIC, camera.MATTE, camera.FLAT]
    n = int(rng.integers(1, 9))
    tori = []
    for _ in range(n):
        R = float(rng.uniform(0.5, 3.0))
        tori.append((tuple(rng.uniform(-2.0, 2.0, 3)), R, float(rng.uniform(0.05, 0.9) * R), int(rng.integers(0, 4))))
    sc = abi.Scene(tori, mats)
    W, H = int(rng.integers(5, 30)) * 8 + int(rng.integers(0, 8)), int(rng.integers(5, 24)) * 8 + int(rng.integers(0, 8))
    toroidal = bool(rng.integers(0, 2))
    eye = rng.normal(size=3)
    eye = tuple(eye / np.linalg.norm(eye) * rng.uniform(0.0 if toroidal else 2.0, 8.0))
    center = tuple(rng.uniform(-1.0, 1.0, 3) + (np.array([6.0, 0.0, 0.0]) if toroidal else 0.0))
    g = camera.globals_for(eye, center, W, H, fov_deg=float(rng.uniform(20.0, 110.0)))
    pc = abi.make_push(clear=tuple(rng.uniform(0.0, 1.0, 3)) + (1.0,), light_pos=tuple(rng.uniform(-12.0, 16.0, 3)),
                       light_intensity=float(rng.uniform(10.0, 200.0)), light_type=int(rng.integers(0, 2)),
                       max_depth=int(rng.integers(1, 7)), rho=float(rng.uniform(0.5, 5.0)))
    if n > 1 and int(rng.integers(0, 3)) == 0:
        # every third scene: the tori share an axis line — shells inside, around and across one another, slightly
        # different R and heights (what the enclosure cull of T3 works on); sometimes the eye sits among the shells
        c0, R0 = tori[0][0], tori[0][1]
        tori = [((c0[0], c0[1] + (float(rng.uniform(-0.05, 0.05)) * R0 if i % 2 else 0.0), c0[2]), R0 * (1.0 + float(rng.uniform(-0.03, 0.03))),
                 float(rng.uniform(0.04, 0.9)) * R0 * 0.97, t_[3]) for i, t_ in enumerate(tori)]
        sc = abi.Scene(tori, mats)
        if int(rng.integers(0, 4)) == 0:
            a_ = float(rng.uniform(0.0, 6.28))
            eye = (c0[0] + R0 * np.cos(a_) * float(rng.uniform(0.7, 1.3)), c0[1] + float(rng.uniform(-0.3, 0.3)) * R0, c0[2] + R0 * np.sin(a_) * float(rng.uniform(0.7, 1.3)))
            g = camera.globals_for(eye, center, W, H, fov_deg=float(rng.uniform(20.0, 110.0)))
    return sc, g, pc, W, H, int(toroidal)


@pytest.mark.parametrize("seed", range(24))
def test_random_scenes(tr, oracle, seed):
    """Fuzz: 24 seeded random scenes/cameras through every render variant and both classification
    levels — first-hit records bit-exact, query counts identical, colours within tolerance."""
    sc, g, pc, W, H, cam = random_case(seed)
    for variant, fine in (("listed", 0), ("listed", 1), ("persistent", 1), ("static", 0)):
        tr.set_render_variant(variant)
        tr.enable_stats(True)
        tr.set_classification(fine)
        try:
            _, _, wstats = check_render(tr, oracle, sc, g, pc, W, H, cam)
            st = tr.stats()
            assert q(st) == q(wstats), (variant, fine)
            # work counters: the static variant traces every test; the macro-level classification only
            # removes tests that the bounding-volume culls of T1 reject too, so "solved" is unchanged
            if variant == "static":
                assert st["traced_tests"] == wstats["traced_tests"]
            if fine == 0:
                assert st["solved_tests"] == wstats["solved_tests"], (variant, fine)
            assert st["traced_tests"] >= st["solved_tests"] > 0 and st["evaluations"] >= st["solved_tests"]
        finally:
            tr.set_classification(abi.TRT_CLASSIFY_AUTO)
            tr.enable_stats(False)
            tr.set_render_variant("listed")


@pytest.mark.parametrize("variant", ["static", "persistent", "listed"])
def test_unaligned_hit_streams(tr, variant):
    """First-hit streams that are only 4-byte aligned (the clear path then may not use its
    dwordx4 stores) and a subset of the streams: same bits as the aligned render."""
    import torch
    dev = torch.device("cuda:0")
    W, H = 256, 96
    sc, g, pc = camera.single_torus_scene(), camera.baseline_camera(W, H), camera.baseline_push(3)
    s = torch.cuda.current_stream().cuda_stream
    names = ("t", "pz", "ny", "id")
    tr.set_render_variant(variant)
    try:
        ref = {k: torch.zeros(W * H, device=dev, dtype=torch.int32 if k == "id" else torch.float32) for k in names}
        img0 = torch.zeros(H, W, 4, device=dev)
        tr.render_dev(sc, g, pc, W, H, img0.data_ptr(), hit_ptrs={k: v.data_ptr() for k, v in ref.items()}, stream=s)
        big = {k: torch.zeros(W * H + 3, device=dev, dtype=ref[k].dtype) for k in names}
        off = {"t": 1, "pz": 2, "ny": 3, "id": 1}
        img1 = torch.zeros(H, W, 4, device=dev)
        tr.render_dev(sc, g, pc, W, H, img1.data_ptr(),
                      hit_ptrs={k: big[k][off[k]:].data_ptr() for k in names}, stream=s)
        torch.cuda.synchronize()
    finally:
        tr.set_render_variant("listed")
    assert torch.equal(img0.view(torch.int32), img1.view(torch.int32))
    for k in names:
        got = big[k][off[k]:off[k] + W * H]
        assert torch.equal(got.view(torch.int32), ref[k].view(torch.int32)), k
        assert big[k][:off[k]].abs().sum().item() == 0 and big[k][off[k] + W * H:].abs().sum().item() == 0   # nothing written outside


def test_cost_feedback_orders_tiles_and_changes_nothing():
    """Cost feedback: from the second frame on a multi-torus scene's heavy tiles are traced first — the
    order of the LIVE list is the ONLY thing that may change.  Four frames of the eight nested tori through one context:
    image, first-hit streams and every query / work counter bit-identical from frame to frame; then a different size and a
    single-torus frame in between (stale costs of another geometry must be harmless), and the first frame again."""
    import torch
    from toroidal_ray_tracing_amd.tracer import Tracer
    dev = torch.device("cuda:0")
    t = Tracer(0)
    s = torch.cuda.current_stream().cuda_stream
    try:
        sc8, sc1 = camera.nested_tori_scene(), camera.single_torus_scene()

        def frame(sc, W, H, depth=5, stats=False, rendered=False):
            g, pc = camera.baseline_camera(W, H), camera.baseline_push(depth)
            rgba = torch.full((H, W, 4), -3.0, device=dev)
            hits = {k: torch.full((H * W,), -3.0, device=dev) for k in ("t", "px", "py", "pz", "nx", "ny", "nz")}
            rend = torch.full((W * H, 16), -3.0, device=dev) if rendered else None
            t.enable_stats(stats)   # (the counted instantiations take no part in the feedback: they neither time nor reorder)
            t.render_dev(sc, g, pc, W, H, rgba.data_ptr(), hit_ptrs={k: v.data_ptr() for k, v in hits.items()},
                         rendered_ptr=rend.data_ptr() if rendered else 0, stream=s)
            torch.cuda.synchronize()
            return [rgba] + [hits[k] for k in sorted(hits)] + ([rend] if rendered else []), (t.stats() if stats else None)

        same = lambda a, b: all(torch.equal(x.view(torch.int32), y.view(torch.int32)) for x, y in zip(a, b))
        ref, st0 = frame(sc8, 1024, 768, stats=True)       # counted: no history is written
        for k in range(4):                                   # first of these without history, the others heavy-first
            got, _ = frame(sc8, 1024, 768)
            assert same(ref, got), f"frame {k + 2}"
        got, st = frame(sc8, 1024, 768, stats=True)        # a counted frame consumes nothing and counts the same
        assert same(ref, got) and st == st0
        refr, _ = frame(sc8, 1024, 768, rendered=True)     # the RenderedData instantiation takes part too
        for k in range(2):
            got, _ = frame(sc8, 1024, 768, rendered=True)
            assert same(refr, got)
        frame(sc8, 520, 264)       # another geometry: the cost words now belong to other tiles
        frame(sc1, 1024, 768)      # a single-torus frame does not use them at all
        got, _ = frame(sc8, 1024, 768)
        assert same(ref, got)
        t.set_solver(abi.TRT_SOLVE_F64)
        ref64, _ = frame(sc8, 1024, 768)
        for k in range(2):
            got, _ = frame(sc8, 1024, 768)
            assert same(ref64, got)
        # the feedback is device state like any other: three frames captured into ONE hipGraph (every frame orders its tiles
        # by the frame before, also across replays) reproduce the eager frame, replay after replay
        t.set_solver(abi.TRT_SOLVE_F32)
        t.enable_stats(False)
        W, H = 1024, 768
        g, pc = camera.baseline_camera(W, H), camera.baseline_push(5)
        outs = [torch.full((H, W, 4), -3.0, device=dev) for _ in range(3)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            t.render_dev(sc8, g, pc, W, H, outs[0].data_ptr(), stream=side.cuda_stream)   # scratch sized outside the capture
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for o in outs:
                    t.render_dev(sc8, g, pc, W, H, o.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
            for _ in range(3):
                for o in outs:
                    o.fill_(-3.0)
                graph.replay()
                side.synchronize()
                assert all(torch.equal(o.view(torch.int32), ref[0].view(torch.int32)) for o in outs)
        torch.cuda.current_stream().wait_stream(side)
    finally:
        t.close()


def test_frame_sequences_keep_no_state(oracle):
    """One ctx, 40 frames in a row that change size, camera model, scene, kernel variant, solver and
    depth at random: every frame must equal the oracle's — nothing (tile lists, their
    tile-list counters, the cached toroidal tables, grow-only scratch) may leak from one
    frame into the next."""
    from toroidal_ray_tracing_amd.tracer import Tracer
    rng = np.random.default_rng(4242)
    t = Tracer(0)
    try:
        for k in range(40):
            sc, g, pc, W, H, cam = random_case(int(rng.integers(0, 24)))
            W, H = int(rng.integers(1, 26)) * 8 + int(rng.integers(0, 8)), int(rng.integers(1, 20)) * 8 + int(rng.integers(0, 8))
            g = camera.globals_for(tuple(rng.normal(size=3) * 3.0), tuple(rng.uniform(-1, 1, 3) + (np.array([6.0, 0, 0]) if cam else 0)), W, H,
                                   fov_deg=float(rng.uniform(25, 100)))
            variant = ["listed", "persistent", "static"][int(rng.integers(0, 3))]
            solver = [abi.TRT_SOLVE_F32, abi.TRT_SOLVE_F64][int(rng.integers(0, 2))]
            t.set_render_variant(variant)
            t.set_solver(solver)
            t.enable_stats(bool(rng.integers(0, 2)))
            rgba, hits = t.render(sc, g, pc, W, H, cam)
            wr, wh, _, _ = oracle.render(sc, g, pc, W, H, cam, precision=solver, nthreads=8)
            assert_hits_equal(hits, wh, f"frame {k} ({variant}, {W}x{H}, cam {cam})")
            np.testing.assert_allclose(rgba, wr, rtol=COLOR_RTOL, atol=COLOR_ATOL)
    finally:
        t.close()


def test_render_is_graph_capturable(tr):
    """trt_render_dev makes no allocation and no synchronisation once its buffers exist, so a frame
    loop can be captured into a hipGraph (small frames are launch-bound: two launches per frame)
    and replayed: the replays reproduce the eager frames bit for bit, and the tile-list counters do
    not accumulate from replay to replay (the query counts stay those of ONE frame)."""
    import torch
    dev = torch.device("cuda:0")
    W = H = 256
    sc, g = camera.single_torus_scene(), camera.baseline_camera(W, H)
    pcs = [camera.baseline_push(d) for d in (1, 3, 5)]
    eager = [torch.zeros(H, W, 4, device=dev) for _ in pcs]
    cur = torch.cuda.current_stream()
    tr.enable_stats(True)
    try:
        for img, pc in zip(eager, pcs):           # also sizes the ctx's tile lists before the capture
            tr.render_dev(sc, g, pc, W, H, img.data_ptr(), stream=cur.cuda_stream)
        want_stats = tr.stats()                    # of the last eager frame
        replayed = [torch.zeros(H, W, 4, device=dev) for _ in pcs]
        graphs = []
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for img, pc in zip(replayed, pcs):     # ONE frame per graph: the hardest case for the counters
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=side):
                    tr.render_dev(sc, g, pc, W, H, img.data_ptr(), stream=side.cuda_stream)
                graphs.append(gr)
        cur.wait_stream(side)
        for _ in range(12):
            for img, gr in zip(replayed, graphs):
                img.zero_()
                gr.replay()
        torch.cuda.synchronize()
        got_stats = tr.stats()
    finally:
        tr.enable_stats(False)
    for a, b in zip(eager, replayed):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    assert got_stats == want_stats and got_stats["primary_tests"] == W * H
    # an eager frame after the replays sees clean counters too
    again = torch.zeros(H, W, 4, device=dev)
    tr.render_dev(sc, g, pcs[2], W, H, again.data_ptr(), stream=cur.cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(again.view(torch.int32), eager[2].view(torch.int32))


def test_graph_of_32_frames_and_mixed_eager_replay(tr):
    """The shape that faulted in round 1 (gpurun_out/bench_graph.log: 32 frames captured into ONE
    hipGraph at 256², stats off) — run once — and the sequences the advisor listed: capture, eager,
    replay, eager on one ctx.  Every frame publishes its own list lengths and leaves the accumulators
    zero (classify_publish), so eager frames and replays may be mixed in any order: images bit-identical,
    query counts those of ONE frame."""
    import torch
    dev = torch.device("cuda:0")
    W = H = 256
    sc, g = camera.single_torus_scene(), camera.baseline_camera(W, H)
    pc5, pc2 = camera.baseline_push(5), camera.baseline_push(2)
    cur = torch.cuda.current_stream()
    want5, want2 = torch.zeros(H, W, 4, device=dev), torch.zeros(H, W, 4, device=dev)
    tr.render_dev(sc, g, pc5, W, H, want5.data_ptr(), stream=cur.cuda_stream)   # also sizes the ctx
    tr.render_dev(sc, g, pc2, W, H, want2.data_ptr(), stream=cur.cuda_stream)
    torch.cuda.synchronize()
    # (1) 32 frames in one graph, stats off
    imgs = [torch.zeros(H, W, 4, device=dev) for _ in range(32)]
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    g32 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g32, stream=side):
            for k, img in enumerate(imgs):
                tr.render_dev(sc, g, pc5 if k % 2 == 0 else pc2, W, H, img.data_ptr(), stream=side.cuda_stream)
    cur.wait_stream(side)
    for _ in range(3):
        g32.replay()
    torch.cuda.synchronize()
    for k, img in enumerate(imgs):
        assert torch.equal(img.view(torch.int32), (want5 if k % 2 == 0 else want2).view(torch.int32)), k
    # (2) capture A (1 frame, counted) -> eager E1 -> replay A -> eager E2 -> replay A: stats and images
    tr.enable_stats(True)
    try:
        a_img, e_img = torch.zeros(H, W, 4, device=dev), torch.zeros(H, W, 4, device=dev)
        tr.render_dev(sc, g, pc5, W, H, e_img.data_ptr(), stream=cur.cuda_stream)
        one = tr.stats()
        ga = torch.cuda.CUDAGraph()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            with torch.cuda.graph(ga, stream=side):
                tr.render_dev(sc, g, pc5, W, H, a_img.data_ptr(), stream=side.cuda_stream)
        cur.wait_stream(side)
        for step in ("eager", "replay", "eager", "replay", "replay", "eager"):
            a_img.zero_(), e_img.zero_()
            if step == "eager":
                tr.render_dev(sc, g, pc5, W, H, e_img.data_ptr(), stream=cur.cuda_stream)
                got = e_img
            else:
                ga.replay()
                got = a_img
            torch.cuda.synchronize()
            assert tr.stats() == one, step
            assert torch.equal(got.view(torch.int32), want5.view(torch.int32)), step
    finally:
        tr.enable_stats(False)
    # (2b) the passes either side of the path are capturable too: render -> tonemap -> re-projection in one graph
    n = 20_000
    gen = torch.Generator(device=dev).manual_seed(3)
    cloud = torch.zeros(n, 8, device=dev)
    cloud[:, :3] = torch.rand(n, 3, device=dev, generator=gen) * 4 - 2
    cloud[:, 4:7] = torch.rand(n, 3, device=dev, generator=gen)
    vp = camera.perspective_vk(60, 1.0) @ camera.look_at((0.5, 1.0, 5.0), (0.0, 0.0, 0.0))
    img, img8, spl = (torch.zeros(H, W, 4, device=dev), torch.zeros(H, W, 4, dtype=torch.uint8, device=dev), torch.zeros(H, W, 4, device=dev))
    w8, wspl = torch.zeros_like(img8), torch.zeros_like(spl)
    tr.post_dev(want5.data_ptr(), W * H, 0, w8.data_ptr(), stream=cur.cuda_stream)        # eager references (size the scratch)
    tr.splat_dev(cloud.data_ptr(), n, vp, W, H, wspl.data_ptr(), stream=cur.cuda_stream)
    torch.cuda.synchronize()
    gp = torch.cuda.CUDAGraph()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        with torch.cuda.graph(gp, stream=side):
            tr.render_dev(sc, g, pc5, W, H, img.data_ptr(), stream=side.cuda_stream)
            tr.post_dev(img.data_ptr(), W * H, 0, img8.data_ptr(), stream=side.cuda_stream)
            tr.splat_dev(cloud.data_ptr(), n, vp, W, H, spl.data_ptr(), stream=side.cuda_stream)
    cur.wait_stream(side)
    for _ in range(3):
        img8.zero_(), spl.zero_()
        gp.replay()
    torch.cuda.synchronize()
    assert torch.equal(img8, w8) and torch.equal(spl.view(torch.int32), wspl.view(torch.int32))
    # (2c) a later, larger eager frame outgrows the ctx's tile lists: the graphs captured before keep replaying on the
    #      blocks their arguments name (outgrown scratch is retired, not freed, until trt_destroy)
    Wb = 1024
    bigger = torch.zeros(Wb, Wb, 4, device=dev)
    tr.render_dev(sc, camera.baseline_camera(Wb, Wb), pc5, Wb, Wb, bigger.data_ptr(), stream=cur.cuda_stream)
    torch.cuda.synchronize()
    for img_ in imgs:
        img_.zero_()
    g32.replay()
    torch.cuda.synchronize()
    for k, img_ in enumerate(imgs):
        assert torch.equal(img_.view(torch.int32), (want5 if k % 2 == 0 else want2).view(torch.int32)), k
    # (3) a capture that would need a larger scratch (or a toroidal table upload) is refused, not allocated
    from toroidal_ray_tracing_amd.tracer import Tracer, TrtError
    t2 = Tracer(0)
    try:
        big = torch.zeros(64, 64, 4, device=dev)
        gx = torch.cuda.CUDAGraph()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            with torch.cuda.graph(gx, stream=side):
                with pytest.raises(TrtError) as e:
                    t2.render_dev(sc, camera.baseline_camera(64, 64), pc2, 64, 64, big.data_ptr(), stream=side.cuda_stream)
                assert e.value.code == abi.TRT_E_INVALID and "captured" in str(e.value)
                torch.zeros(1, device=dev)   # keep the capture non-empty
        cur.wait_stream(side)
    finally:
        t2.close()


def test_post_pass_bit_exact(tr, oracle):
    """trt_post_dev (tonemap of post.frag) — float and UNORM8 outputs bit for bit equal to the
    oracle: the exp2/log2 polynomials use only correctly rounded operations."""
    import torch
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(0, 1.2, 1_000_003 * 4), np.logspace(-40, 38, 4001), [0, -2.5, np.nan, np.inf, 1, 1e-45, 0.8]])
    x = np.float32(x[: (len(x) // 4) * 4]).reshape(-1, 4)
    d_in = torch.from_numpy(x).to(dev)
    d_f = torch.empty_like(d_in)
    d_u = torch.empty(x.shape, dtype=torch.uint8, device=dev)
    tr.post_dev(d_in.data_ptr(), x.shape[0], d_f.data_ptr(), d_u.data_ptr(), stream=s)
    torch.cuda.synchronize()
    wf, wu = oracle.post(x)
    np.testing.assert_array_equal(d_f.cpu().numpy().view(np.uint32), wf.view(np.uint32))
    np.testing.assert_array_equal(d_u.cpu().numpy(), wu)
    # either output alone, empty input, and a rendered frame end to end
    d_u.zero_()
    tr.post_dev(d_in.data_ptr(), x.shape[0], 0, d_u.data_ptr(), stream=s)
    tr.post_dev(d_in.data_ptr(), 0, d_f.data_ptr(), 0, stream=s)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(d_u.cpu().numpy(), wu)
    W, H = 256, 128
    sc, g, pc = camera.single_torus_scene(material=camera.PLASTIC), camera.baseline_camera(W, H), camera.baseline_push(3)
    img = torch.empty(H, W, 4, device=dev)
    out8 = torch.empty(H, W, 4, dtype=torch.uint8, device=dev)
    tr.render_dev(sc, g, pc, W, H, img.data_ptr(), stream=s)
    tr.post_dev(img.data_ptr(), W * H, 0, out8.data_ptr(), stream=s)
    torch.cuda.synchronize()
    _, w8 = oracle.post(img.cpu().numpy())
    np.testing.assert_array_equal(out8.cpu().numpy(), w8)
    assert (out8[..., 3] == 255).all() and out8[..., :3].float().std() > 1


def test_splat_bit_exact(tr, oracle):
    """trt_splat_dev (point-cloud re-projection of ray_tracing__before_second) against the
    sequential rasteriser of the oracle: random clouds with duplicated points (depth ties),
    clipped and degenerate points, several image sizes."""
    import torch
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(21)
    for W, H, n, eye in [(200, 136, 50_000, (1.0, 2.0, 6.0)), (64, 64, 300_000, (0.0, 0.0, 4.0)), (333, 77, 1000, (5.0, 0.1, 0.2))]:
        pts = np.zeros((n, 8), np.float32)
        pts[:, :3] = rng.uniform(-3, 3, (n, 3))
        pts[:, 4:7] = rng.uniform(0, 1, (n, 3))
        pts[n // 2:n // 2 + n // 10] = pts[:n // 10]                  # duplicates -> equal depth, different colour
        pts[n // 2:n // 2 + n // 10, 4:7] = rng.uniform(0, 1, (n // 10, 3))
        pts[::97, :3] = np.finfo(np.float32).min                       # the "-nan" convention of loadPoints()
        pts[5::101, 0] = np.nan
        vp = camera.perspective_vk(65, W / H) @ camera.look_at(eye, (0, 0, 0))
        d_pts = torch.from_numpy(pts).to(dev)
        out = torch.empty(H, W, 4, device=dev)
        tr.splat_dev(d_pts.data_ptr(), n, vp, W, H, out.data_ptr(), stream=s)
        torch.cuda.synchronize()
        want = oracle.splat(pts, vp, W, H)
        np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
    # point sizes that straddle several 128x64 bins (binned form: <= 32 pixels) and one beyond it (one-pass form),
    # on an image of 3 x 4 bins with a ragged edge; repeated calls reuse the bin counters
    W, H, n = 300, 200, 40_000
    pts = np.zeros((n, 8), np.float32)
    pts[:, :3] = rng.uniform(-3, 3, (n, 3))
    pts[:, 4:7] = rng.uniform(0, 1, (n, 3))
    pts[n // 3:n // 3 + 500] = pts[:500]                               # depth ties across bins
    vp = camera.perspective_vk(70, W / H) @ camera.look_at((0.5, 1.0, 5.0), (0, 0, 0))
    d_pts = torch.from_numpy(pts).to(dev)
    out = torch.empty(H, W, 4, device=dev)
    for size in (1.0, 2.5, 9.0, 31.5, 40.0, 2.5):
        tr.splat_dev(d_pts.data_ptr(), n, vp, W, H, out.data_ptr(), point_size=size, stream=s)
        torch.cuda.synchronize()
        want = oracle.splat(pts, vp, W, H, point_size=size)
        np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32), err_msg=f"point_size {size}")
    # larger images take the other scatter kernels: up to 2,048 bins (LDS-sorted, the wider instantiation) and beyond (direct);
    # big points on the small image above make a chunk's records overflow the LDS staging area
    for W2, H2, n2 in [(4096, 2048, 120_000), (8192, 4100, 120_000)]:
        pts2 = np.zeros((n2, 8), np.float32)
        pts2[:, :3] = rng.uniform(-3, 3, (n2, 3))
        pts2[:, 4:7] = rng.uniform(0, 1, (n2, 3))
        pts2[n2 // 2:n2 // 2 + 2000] = pts2[:2000]
        vp2 = camera.perspective_vk(70, W2 / H2) @ camera.look_at((0.5, 1.0, 5.0), (0, 0, 0))
        d2 = torch.from_numpy(pts2).to(dev)
        out2 = torch.empty(H2, W2, 4, device=dev)
        for size in (2.5, 20.0):
            tr.splat_dev(d2.data_ptr(), n2, vp2, W2, H2, out2.data_ptr(), point_size=size, stream=s)
            torch.cuda.synchronize()
            want = oracle.splat(pts2, vp2, W2, H2, point_size=size)
            np.testing.assert_array_equal(out2.cpu().numpy().view(np.uint32), want.view(np.uint32), err_msg=f"{W2}x{H2} point_size {size}")
        del d2, out2
    # crowded bins: the paged scatter hands a bin page after page (a 2048² image has 512 bins, a page holds 4,096 records
    # for a cloud of this size): everything inside one bin, on a corner shared by four bins with big points (four records
    # per point), and a sheet in screen order (a block's whole run goes to one bin and spans pages) — a hundred window
    # rotations per bin, every one bit for bit
    W3 = H3 = 2048
    vp3 = camera.perspective_vk(60, 1.0) @ camera.look_at((0.0, 0.0, 5.0), (0, 0, 0))
    out3 = torch.empty(H3, W3, 4, device=dev)
    for kind, n3, size in (("cluster", 400_000, 2.5), ("corner", 150_000, 31.0), ("sheet", 400_000, 1.0)):
        pts3 = np.zeros((n3, 8), np.float32)
        u = rng.uniform(0, 1, (n3, 3))
        if kind == "cluster":
            pts3[:, :3] = (u - 0.5) * 0.05 + np.array([0.3, 0.2, 0.0])
        elif kind == "corner":
            pts3[:, :3] = (u - 0.5) * 0.02
        else:
            k3 = np.arange(n3); side = int(n3 ** 0.5) + 1
            pts3[:, 0] = ((k3 % side) / side - 0.5) * 5; pts3[:, 1] = ((k3 // side) / side - 0.5) * 5; pts3[:, 2] = u[:, 2] * 0.01
        pts3[:, 4:7] = rng.uniform(0, 1, (n3, 3))
        pts3[n3 // 2:n3 // 2 + 3000] = pts3[:3000]
        d3 = torch.from_numpy(pts3).to(dev)
        for rep in range(2):   # twice: the second call finds the bins' words and the pool as the first left them
            tr.splat_dev(d3.data_ptr(), n3, vp3, W3, H3, out3.data_ptr(), point_size=size, stream=s)
        torch.cuda.synchronize()
        want = oracle.splat(pts3, vp3, W3, H3, point_size=size)
        np.testing.assert_array_equal(out3.cpu().numpy().view(np.uint32), want.view(np.uint32), err_msg=f"crowded: {kind}")
        del d3
    del out3
    # no points at all: the clear colour everywhere
    tr.splat_dev(0, 0, vp, W, H, out.data_ptr(), clear=(0.1, 0.2, 0.3, 1.0), stream=s)
    torch.cuda.synchronize()
    assert torch.equal(out, torch.tensor([0.1, 0.2, 0.3, 1.0], device=dev).expand(H, W, 4))


def test_reprojection_and_post_in_a_graph(tr, oracle):
    """trt_splat_dev + trt_post_dev captured into ONE hipGraph once the ctx's scratch is sized and replayed:
    the bins' words, the page pool and the tickets are left zero by every call, so replays, eager calls with ANOTHER cloud in
    between and replays again all give the oracle's image — on a cloud crowded enough that bins rotate their page windows."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(9)
    W, H = 512, 256
    vp = camera.perspective_vk(60, W / H) @ camera.look_at((0.0, 0.5, 5.0), (0, 0, 0))

    def cloud(n, spread):
        pts = np.zeros((n, 8), np.float32)
        pts[:, :3] = rng.uniform(-1, 1, (n, 3)) * spread
        pts[:, 4:7] = rng.uniform(0, 1, (n, 3))
        pts[n // 2:n // 2 + 1000] = pts[:1000]
        return pts

    a_np, b_np = cloud(600_000, (0.6, 0.4, 0.3)), cloud(150_000, (2.5, 1.5, 1.0))   # a: 8 bins' worth of a 16-bin image, crowded
    d_a, d_b = torch.from_numpy(a_np).to(dev), torch.from_numpy(b_np).to(dev)
    img = torch.empty(H, W, 4, device=dev)
    out8 = torch.empty(H, W, 4, dtype=torch.uint8, device=dev)
    want_a = oracle.splat(a_np, vp, W, H)
    want_b = oracle.splat(b_np, vp, W, H)
    _, want_a8 = oracle.post(want_a)
    cur = torch.cuda.current_stream()
    tr.splat_dev(d_a.data_ptr(), len(a_np), vp, W, H, img.data_ptr(), stream=cur.cuda_stream)      # sizes the scratch
    torch.cuda.synchronize()
    np.testing.assert_array_equal(img.cpu().numpy().view(np.uint32), want_a.view(np.uint32))
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side), torch.cuda.graph(gr, stream=side):
        tr.splat_dev(d_a.data_ptr(), len(a_np), vp, W, H, img.data_ptr(), stream=side.cuda_stream)
        tr.post_dev(img.data_ptr(), W * H, 0, out8.data_ptr(), stream=side.cuda_stream)
    cur.wait_stream(side)
    for k in range(4):
        img.zero_(); out8.zero_()
        gr.replay()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(img.cpu().numpy().view(np.uint32), want_a.view(np.uint32), err_msg=f"replay {k}")
        np.testing.assert_array_equal(out8.cpu().numpy(), want_a8)
        if k == 1:   # an eager call with another (smaller: nothing has to grow) cloud between two replays
            tr.splat_dev(d_b.data_ptr(), len(b_np), vp, W, H, img.data_ptr(), stream=cur.cuda_stream)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(img.cpu().numpy().view(np.uint32), want_b.view(np.uint32))


def test_capture_then_reproject(tr, oracle):
    """The research pipeline end to end: toroidal capture (RenderedData pos + colour) ->
    Point cloud (createCloudDataBuffer: pos.w = color.w = 0) -> re-projection from a pinhole
    viewpoint, GPU == oracle bit for bit on the same captured data."""
    import torch
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    W = H = 256
    sc = camera.single_torus_scene(R=6.0, r=1.5, material=camera.PLASTIC)
    g, pc = camera.toroidal_camera(W, H), abi.make_push(max_depth=3, rho=4.0)
    rend = torch.empty(W * H, 16, device=dev)
    tr.render_dev(sc, g, pc, W, H, 0, camera=1, rendered_ptr=rend.data_ptr(), stream=s)
    cloud = torch.zeros(W * H, 8, device=dev)
    cloud[:, :3] = rend[:, 0:3]
    cloud[:, 4:7] = rend[:, 4:7]
    vp = camera.perspective_vk(60, 1.0) @ camera.look_at((0.5, 1.0, -1.0), (6.0, 0.0, 2.0))
    out = torch.empty(H, W, 4, device=dev)
    tr.splat_dev(cloud.data_ptr(), W * H, vp, W, H, out.data_ptr(), stream=s)
    torch.cuda.synchronize()
    want = oracle.splat(cloud.cpu().numpy(), vp, W, H)
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert 0.02 < (out[..., :3] != 0.8).any(dim=2).float().mean().item() < 0.98
