"""Multi-rank path on CPU (gloo, world_size 2): the row-group tiling and the in-place all-gathers
reproduce the single-rank frame; `python bench.py --gpus 2` starts two ranks by itself.  The per-rank renders come from the CPU
oracle here (there is no GPU in this container) through a stand-in tracer, so the REAL
``TiledFrame`` (pipelined all-gather, retire, finish) is what runs; on the GPU box the same
class runs over RCCL (tests/test_gpu_parity.py::test_tiled_render_matches_full checks the
kernels' tiling against full-frame renders on one GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from toroidal_ray_tracing_amd import abi, camera
from toroidal_ray_tracing_amd import distributed as trtd


def test_owned_rows_partition():
    for H, G, N in [(64, 8, 2), (96, 4, 3), (70, 8, 4), (16, 8, 8)]:
        rows = [trtd.owned_rows(H, G, N, p) for p in range(N)]
        assert sorted(sum(rows, [])) == list(range(H))
        # matches the C ABI's count
        from toroidal_ray_tracing_amd import lib
        L = lib.load()
        import ctypes as C
        for p in range(N):
            t = abi.trt_tiling(G, N, p, 1)
            assert L.trt_tiling_rows(C.byref(t), H) == len(rows[p])
    t = abi.trt_tiling(0, 2, 0, 1)
    assert lib.load().trt_tiling_rows(C.byref(t), 64) == 0  # invalid tiling


def test_deinterleave_inverts_tiling():
    H, W, G, N = 48, 5, 4, 3
    img = torch.arange(H * W * 4, dtype=torch.float32).view(H, W, 4)
    parts = [img[trtd.owned_rows(H, G, N, p)] for p in range(N)]
    gathered = torch.stack(parts)
    assert torch.equal(trtd.deinterleave(gathered, H, W, G, N), img)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _OracleTracer:
    """Stand-in for tracer.Tracer on CPU: the same render_tiled_dev signature, pixels from the
    CPU oracle written straight into the (CPU) buffer whose address TiledFrame passes."""

    def __init__(self):
        from oracle import oracle
        self.oracle = oracle

    def tiling_rows(self, tiling, H):
        return len(trtd.owned_rows(H, tiling.group_rows, tiling.n_parts, tiling.part))

    def render_tiled_dev(self, scene, g, pc, W, H, tiling, rgba_ptr, camera=0, hit_ptrs=None, stream=0):
        import ctypes
        rows = trtd.owned_rows(H, tiling.group_rows, tiling.n_parts, tiling.part)
        local = np.zeros((len(rows), W, 4), np.float32)
        G = tiling.group_rows
        for k in range(0, len(rows), G):
            y0 = rows[k]
            band, _, _, _ = self.oracle.render(scene, g, pc, W, H, camera, rows=(y0, y0 + G), want_hits=False)
            local[k:k + G] = band[y0:y0 + G]
        ctypes.memmove(rgba_ptr, local.ctypes.data, local.nbytes)

    def render_batch_dev(self, scene, frames, W, H, tiling=None, camera=0, stream=0):
        for g, pc, rgba_ptr, hit_ptrs in frames:
            self.render_tiled_dev(scene, g, pc, W, H, tiling, rgba_ptr, camera=camera, hit_ptrs=hit_ptrs, stream=stream)

    def post_dev(self, rgba_ptr, n_pixels, f32_out_ptr=0, unorm8_out_ptr=0, stream=0):
        import ctypes
        src = np.ctypeslib.as_array(ctypes.cast(rgba_ptr, ctypes.POINTER(ctypes.c_float)), shape=(n_pixels, 4))
        _, u8 = self.oracle.post(src)
        ctypes.memmove(unorm8_out_ptr, u8.ctypes.data, u8.nbytes)


class _Stream:
    cuda_stream = 0


def _worker(rank, world, port, W, H, G, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tr = _OracleTracer()
        sc, g = camera.single_torus_scene(), camera.baseline_camera(W, H)
        frame = trtd.TiledFrame(tr, W, H, world, rank, torch.device("cpu"), group_rows=G)
        assert frame.local_rows == H // world and "pipelined" in frame.describe() and frame.cycles == H // (G * world)
        ok = True
        # three frames with different maxDepth through the two-deep pipeline; after finish() the
        # assembled frame is the LAST one, and after each step the frame from two steps ago is retired
        for depth in (1, 2, 3):
            frame.render(sc, g, camera.baseline_push(depth), abi.TRT_CAMERA_PINHOLE, _Stream())
        full = frame.finish().clone()
        # the 8-bit gather mode: tonemapped rows, same pipeline
        frame8 = trtd.TiledFrame(tr, W, H, world, rank, torch.device("cpu"), group_rows=G, gather="rgba8")
        for depth in (2, 3):
            frame8.render(sc, g, camera.baseline_push(depth), abi.TRT_CAMERA_PINHOLE, _Stream())
        full8 = frame8.finish().clone()
        # gathers once per batch of two frames: frames 1 and 3 are gathered (maxDepth 2, then 3), frames 0, 2, 4 are not;
        # the batch's tiling is the balanced one (interleaved groups), and the frame handed out is the last GATHERED one
        frameb = trtd.TiledFrame(tr, W, H, world, rank, torch.device("cpu"), gather_every=2)
        assert frameb.group_rows == trtd.default_group_rows(H, world, trtd.DEFAULT_CYCLES) and "every 2 frames" in frameb.describe()
        for depth in (1, 2, 1, 3, 1):
            frameb.render(sc, g, camera.baseline_push(depth), abi.TRT_CAMERA_PINHOLE, _Stream())
        fullb = frameb.finish().clone()
        # three frames in flight (three contexts, three output sets; on the CPU the "streams" are synchronous), a gather
        # every 4th frame: frames 3 and 7 are gathered — from output sets 0 and 1 — and frame 7 (maxDepth 3) is handed out
        framek = trtd.TiledFrame([tr, _OracleTracer(), _OracleTracer()], W, H, world, rank, torch.device("cpu"), gather_every=4)
        assert "3 frames in flight" in framek.describe() and len(framek.locals) == 3
        for depth in (1, 2, 1, 2, 1, 2, 1, 3, 1):
            framek.render(sc, g, camera.baseline_push(depth), abi.TRT_CAMERA_PINHOLE, _Stream())
        fullk = framek.finish().clone()
        # what bench.py --gpus N runs: two frames per launch, the launches alternating over two contexts (four output sets),
        # a gather every 6th frame: frame 5 is gathered out of set 1 of the THIRD launch (context 0 again), frame 11
        # (maxDepth 3, set 3, context 1) is handed out; a thirteenth frame stays queued until finish() flushes it
        frameq = trtd.TiledFrame([tr, _OracleTracer()], W, H, world, rank, torch.device("cpu"), gather_every=6, batch=2)
        assert "2 frames per launch" in frameq.describe() and "2 streams" in frameq.describe() and len(frameq.locals) == 4
        for depth in (1, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 3, 1):
            frameq.render(sc, g, camera.baseline_push(depth), abi.TRT_CAMERA_PINHOLE, _Stream())
        fullq = frameq.finish().clone()
        if rank == 0:
            want, _, _, _ = tr.oracle.render(sc, g, camera.baseline_push(3), W, H, want_hits=False)
            ok = bool(np.array_equal(full.numpy(), want))
            ok = ok and full8.dtype == torch.uint8 and bool(np.array_equal(full8.numpy(), tr.oracle.post(want)[1]))
            ok = ok and bool(np.array_equal(fullb.numpy(), want)) and bool(np.array_equal(fullk.numpy(), want))
            ok = ok and bool(np.array_equal(fullq.numpy(), want))
            out.put(ok)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gather_reproduces_frame():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    W, H, G, world = 64, 64, 8, 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, H, G, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    assert all(p.exitcode == 0 for p in procs)
    assert out.get(timeout=5) is True


def _bench(args, env=None):
    import subprocess
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, cwd=ROOT,
                          timeout=280, env=e)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n", [2, 4, 8])
def test_bench_launcher_starts_the_ranks_itself(n):
    """`python bench.py --gpus 2` (no torchrun around it): the parent spawns 2 ranks, relays rank 0's line, and
    that line says n_gpus 2.  --backend gloo --dry-run is the CPU rehearsal: rank-coded rows instead of rendered
    ones, the REAL TiledFrame gathers, every rank checks the assembled frame."""
    import json
    p = _bench(["--gpus", str(n), "--backend", "gloo", "--dry-run", "--size", "128" if n == 8 else "64"])   # 8 ranks: the job size of config 5
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["dry_run"] is True and out["gathered_frame_ok"] is True and out["value"] is None


def test_bench_never_reports_the_wrong_job_size():
    """--gpus 8 inside a 1-rank environment (or with fewer ranks than asked) exits non-zero without a result line."""
    p = _bench(["--gpus", "8", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "{" not in p.stdout and "WORLD_SIZE=1" in p.stderr
    # a measurement over gloo is refused as well (gloo is the rehearsal backend)
    p = _bench(["--gpus", "1", "--backend", "gloo", "--no-cpu-baseline"])
    assert p.returncode != 0 and "{" not in p.stdout


def test_default_group_rows():
    assert trtd.default_group_rows(4096, 8, 1) == 512 and trtd.default_group_rows(4096, 2, 1) == 2048
    assert trtd.default_group_rows(4096, 8) == 32 and trtd.default_group_rows(4096, 2) == 128
    assert trtd.default_group_rows(8192, 8) == 64 and trtd.default_group_rows(4096, 1) == 4096
    assert trtd.default_group_rows(64, 2) == 8 and trtd.default_group_rows(48, 3) == 8
    for H, N in [(4096, 8), (4096, 4), (4096, 2), (8192, 8), (64, 2), (48, 3)]:
        G = trtd.default_group_rows(H, N)
        assert H % (G * N) == 0
