"""The C++ host mirror (HelloHip, toroidal_ray_tracing_amd/host/) and its text dumps: the
reference's on-disk formats (ray_tracing__before/hello_vulkan.cpp:1150-1259) carrying the same
numbers as the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from toroidal_ray_tracing_amd import abi, camera

pytestmark = pytest.mark.gpu


def _load(path):
    return np.loadtxt(path, dtype=np.float64, ndmin=2)


def test_reflections_example_matches_oracle(tmp_path, oracle):
    exe = os.path.join(ROOT, "examples", "reflections")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    (tmp_path / "data").mkdir()
    W, H = 96, 64
    out = subprocess.run([exe, str(W), str(H), "2", "5", str(tmp_path) + "/"], check=True, capture_output=True,
                         text=True).stdout
    assert "centre pixel" in out
    got = _load(tmp_path / "data" / "renderedColor0.000000.txt")      # std::to_string(rho = 0)
    assert got.shape == (W * H, 3)
    # same scene through the oracle; the C++ side builds its matrices in float (like nvmath), so
    # compare the frame the library renders for THOSE matrices: re-derive them the same way
    sc = camera.single_torus_scene()
    g = camera.baseline_camera(W, H)
    want, _, _, _ = oracle.render(sc, g, camera.baseline_push(5), W, H, want_hits=False, nthreads=4)
    want = want[..., :3].reshape(-1, 3)
    # float-built vs double-built camera matrices differ in the last ulp: silhouette pixels may
    # flip, everything else agrees to the 6 significant digits the text format keeps
    close = np.isclose(got, want, rtol=2e-4, atol=2e-5).all(axis=1)
    assert close.mean() > 0.995
    # the ground-truth dump (REFL/hello_vulkan.cpp:1065-1110 naming) is the same text, and the PPM holds the
    # tonemapped bytes of the post pass (post.frag): P6 header + W*H RGB triples == oracle.post of the frame
    assert (tmp_path / "data" / "torusgTruth.txt").read_text() == (tmp_path / "data" / "renderedColor0.000000.txt").read_text()
    raw = (tmp_path / "data" / "torusgTruth.ppm").read_bytes()
    head = f"P6\n{W} {H}\n255\n".encode()
    assert raw.startswith(head) and len(raw) == len(head) + W * H * 3
    ppm = np.frombuffer(raw[len(head):], np.uint8).reshape(H, W, 3)
    rgba = np.concatenate([got.reshape(H, W, 3), np.ones((H, W, 1))], axis=2).astype(np.float32)
    _, w8 = oracle.post(rgba)          # from the 6-digit text: bytes may differ by one count at rounding boundaries
    assert np.abs(ppm.astype(int) - w8[..., :3].astype(int)).max() <= 1 and (ppm == w8[..., :3]).mean() > 0.99


def test_toroidal_sweep_formats(tmp_path):
    exe = os.path.join(ROOT, "examples", "toroidal_sweep")
    (tmp_path / "data").mkdir()
    W, H = 64, 32
    subprocess.run([exe, str(tmp_path) + "/", str(W), str(H)], check=True, capture_output=True)
    names = sorted(os.listdir(tmp_path / "data"))
    rhos = [4.5 + 0.5 * k for k in range(12)]                          # 4.5 … 10.0 (BEF/main.cpp:236-258)
    for r in rhos:
        assert f"renderedPosition{r:.6f}.txt" in names and f"renderedColor{r:.6f}.txt" in names
    assert "origins.txt" in names and "directions.txt" in names
    org = _load(tmp_path / "data" / "origins.txt").reshape(W, H, 3)    # index x*H + y
    dirs = _load(tmp_path / "data" / "directions.txt").reshape(W, H, 3)
    # BEF rgen:56-57 with eye 0, omega = theta = 0, rho = 4.5: origin = rho (cos a, 0, sin a)
    a = np.radians(360.0 / W * np.arange(W))
    np.testing.assert_allclose(org[:, 0, 0], 4.5 * np.cos(a), atol=2e-5)
    np.testing.assert_allclose(org[:, 0, 2], 4.5 * np.sin(a), atol=2e-5)
    assert np.all(org[:, :, 1] == 0)
    b = np.radians(360.0 / H * np.arange(H))
    np.testing.assert_allclose(dirs[0, :, 1], np.sin(b), atol=2e-6)
    pos = _load(tmp_path / "data" / "renderedPosition4.500000.txt").reshape(W, H, 3)
    col = _load(tmp_path / "data" / "renderedColor4.500000.txt").reshape(H, W, 3)   # row-major
    hit = np.any(pos != 0, axis=2)
    assert 0.1 < hit.mean() < 0.9
    rho_hit = np.hypot(pos[..., 0], pos[..., 2])[hit]
    g = (rho_hit - 14.0) ** 2 + pos[..., 1][hit] ** 2 - 9.0            # on the R=14, r=3 torus
    assert np.abs(g).max() < 2e-3                                      # 6 significant digits in the file
    assert np.all(col[~hit.T] == 0.8)                                  # misses: clear colour * 0.8


def test_reproject_example_consumes_a_capture(tmp_path):
    """toroidal_sweep writes captures; reproject (the SEC consumer) loads one with the
    reference's text conventions ("x y z" lines, "-nan" -> lowest()) and rasterises it."""
    sweep = os.path.join(ROOT, "examples", "toroidal_sweep")
    reproj = os.path.join(ROOT, "examples", "reproject")
    (tmp_path / "data").mkdir()
    W = H = 64   # square: positions (x*H+y) and colours (row-major) are paired line by line
    subprocess.run([sweep, str(tmp_path) + "/", str(W), str(H)], check=True, capture_output=True)
    # inject the "-nan" convention of the reference's dumps into a few position lines
    pos = tmp_path / "data" / "renderedPosition4.500000.txt"
    lines = pos.read_text().splitlines()
    lines[5] = "-nan(ind) -nan(ind) -nan(ind)"
    lines[7] = "garbage"
    pos.write_text("\n".join(lines) + "\n")
    out = subprocess.run([reproj, str(tmp_path) + "/", "4.500000", "128", "96"], check=True, capture_output=True,
                         text=True).stdout
    n_points, covered = int(out.split()[0]), int(out.split("pixels covered")[0].split(",")[-1])
    assert n_points == W * H and 0 < covered < 128 * 96
    # the re-projected image as the reference dumps it (SEC/hello_vulkan.cpp:781-825): one "r g b" line per pixel
    img = _load(tmp_path / "data" / "torusptCloudImage_10.txt")
    assert img.shape == (128 * 96, 3) and abs(int((img[:, :2] != 0.8).any(axis=1).sum()) - covered) <= 2   # 6-digit text
    raw = (tmp_path / "data" / "torusptCloudImage_10.ppm").read_bytes()
    assert raw.startswith(b"P6\n128 96\n255\n") and len(raw) == len(b"P6\n128 96\n255\n") + 128 * 96 * 3


def test_frames_in_flight_example():
    """examples/frames_in_flight (plain C ABI): three contexts on three streams render the same frame of the eight nested tori
    (FP64 solve) alternately; the program compares their images bit for bit and exits non-zero if they differ."""
    exe = os.path.join(ROOT, "examples", "frames_in_flight")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    p = subprocess.run([exe, "3", "9", "320", "1", "1"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "images of the 3 contexts identical" in p.stdout and p.stdout.count("in flight:") == 3
    # … and trt_render_batch_dev: the 1/8 (above) and the 1/4 part of consecutive frames (maxDepth 1, 2, …) in one pair of
    # launches each, compared by the program with the same frames rendered one by one
    assert "8 per launch; batch identical to the frames rendered alone" in p.stdout
    p = subprocess.run([exe, "1", "8", "320", "1", "1", "4"], capture_output=True, text=True)
    assert p.returncode == 0 and "4 per launch; batch identical to the frames rendered alone" in p.stdout, p.stdout + p.stderr
    p = subprocess.run([exe, "2", "4", "200", "0", "0"], capture_output=True, text=True)    # one torus, FP32
    assert p.returncode == 0 and "identical" in p.stdout
    p = subprocess.run([exe, "1", "16", "512", "0", "0", "8"], capture_output=True, text=True)   # one torus, FP32, 8 parts
    assert p.returncode == 0 and "8 per launch; batch identical to the frames rendered alone" in p.stdout, p.stdout + p.stderr
