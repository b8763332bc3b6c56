"""Multi-GPU image tiling: one process per GPU, rows dealt round-robin in groups, RCCL
all-gather of the rgba32f framebuffer (BASELINE.json north_star).

The reference is single-GPU (``compatibleDevices[0]``, ray_tracing_reflections/main.cpp:143);
pixels are independent (one raygen invocation per pixel, no inter-pixel communication in
raytrace.rgen), so the frame shards by rows with no exchange during the render.  The only
collective is the all-gather of the finished framebuffer, and only when world_size > 1.

Layout.  With N ranks and groups of G rows, rank p owns rows y with (y // G) % N == p and
renders them into a COMPACT local buffer [H/N, W, 4] (``trt_tiling.compact = 1``): exactly
the send buffer ``all_gather_into_tensor`` wants.  The gathered buffer [N, H/N, W, 4] is a
row permutation of the image; ``deinterleave`` restores row-major order with one strided
copy (view [N, H/(G·N), G, W, 4] → permute(1,0,2,3,4)).

xGMI is point-to-point (7 links per GPU): a ring all-gather is bound by ONE link per step,
so the gather is issued as ONE large collective per frame (RCCL then spreads channels over
all links) instead of many small per-band ones.
"""
import torch
import torch.distributed as dist

from . import abi

DEFAULT_GROUP_ROWS = 8   # one 8-row tile band: matches the kernels' 8×8 wave tiles


def owned_rows(H, group_rows, n_parts, part):
    """Image rows owned by `part`, in local-buffer order (pure-Python mirror of trt_tiling)."""
    return [y for y in range(H) if (y // group_rows) % n_parts == part]


def deinterleave(gathered, H, W, group_rows, n_parts, channels=4):
    """[N, H/N, W, C] gathered compact buffers → row-major [H, W, C] image (a strided view;
    call .contiguous() / copy_ to materialise).  Requires H % (group_rows*n_parts) == 0."""
    cycles = H // (group_rows * n_parts)
    v = gathered.view(n_parts, cycles, group_rows, W, channels)
    return v.permute(1, 0, 2, 3, 4).reshape(H, W, channels)


class TiledFrame:
    """Framebuffer + first-hit streams of one rank, and the per-frame render/gather step.

    With world > 1 the step is software-pipelined over two buffer sets: the all-gather of
    frame k (RCCL, asynchronous on the process group's stream) runs while frame k+1 renders, so
    the frame rate is max(render, gather) instead of their sum; ``finish()`` drains the pipe.
    """

    def __init__(self, tracer, W, H, world, rank, device, want_hits=(), group_rows=DEFAULT_GROUP_ROWS,
                 gather=True):
        """gather: True / "fp32" — all-gather the rgba32f framebuffer (default, what north_star
        prescribes); "rgba8" — tonemap each rank's rows (trt_post_dev, post.frag) and all-gather
        the 8-bit image a swapchain would present: 4x fewer bytes over xGMI, the rgba32f image
        stays sharded; False / "none" — no collective."""
        self.tr, self.W, self.H, self.world, self.rank = tracer, W, H, world, rank
        mode = {True: "fp32", False: "none"}.get(gather, gather)
        if mode not in ("fp32", "rgba8", "none"):
            raise ValueError(f"gather={gather!r}")
        self.mode = mode if world > 1 else "none"
        self.group_rows, self.gather = group_rows, self.mode != "none"
        if world > 1 and H % (group_rows * world) != 0:
            raise ValueError(f"H={H} must be a multiple of group_rows*world={group_rows * world}")
        self.tiling = abi.trt_tiling(group_rows, world, rank, 1 if world > 1 else 0)
        self.local_rows = tracer.tiling_rows(self.tiling, H) if world > 1 else H
        self.local_pixels = self.local_rows * W
        f32 = dict(dtype=torch.float32, device=device)
        nbuf = 2 if self.gather else 1
        self.locals = [torch.empty(self.local_rows, W, 4, **f32) for _ in range(nbuf)]
        self.hits = {k: torch.empty(self.local_pixels, **f32) for k in want_hits if k != "id"}
        if "id" in want_hits:
            self.hits["id"] = torch.empty(self.local_pixels, dtype=torch.int32, device=device)
        # concatenated along dim 0 (the form both RCCL and gloo accept); viewed as [N, H/N, W, 4]
        gdt = dict(dtype=torch.uint8 if self.mode == "rgba8" else torch.float32, device=device)
        self.sends = [torch.empty(self.local_rows, W, 4, **gdt) for _ in range(nbuf)] if self.mode == "rgba8" else self.locals
        self.gathered = [torch.empty(world * self.local_rows, W, 4, **gdt) for _ in range(nbuf)] if self.gather else None
        self.full = torch.empty(H, W, 4, **gdt) if self.gather else self.locals[0]
        self._pending = [None] * nbuf   # in-flight all-gather of each buffer set
        self._k = 0

    @property
    def local(self):
        return self.locals[0]

    def describe(self):
        if self.world == 1:
            return "single GPU, full frame"
        what = {"fp32": "all_gather_into_tensor(rgba32f)", "rgba8": "post pass + all_gather_into_tensor(rgba8)"}.get(self.mode)
        return (f"{self.world} ranks x {self.local_rows} rows in interleaved groups of {self.group_rows}; "
                f"{what + ' pipelined behind the next frame + de-interleave' if self.gather else 'no gather'}")

    def _retire(self, b):
        """Wait for the gather of buffer set b and assemble its frame."""
        if self._pending[b] is not None:
            self._pending[b].wait()   # orders the current stream behind the collective
            self._pending[b] = None
            self.full.copy_(deinterleave(self.gathered[b], self.H, self.W, self.group_rows, self.world))

    def render(self, scene, g, pc, camera, stream, events=None):
        """One frame: render this rank's rows; when world > 1 start the all-gather of this frame
        and finish (wait + de-interleave) the frame that used this buffer set two steps ago.
        `events` = (start, end) torch.cuda.Events recorded around the render launches only."""
        hp = {k: v.data_ptr() for k, v in self.hits.items()}
        b = self._k % len(self.locals)
        self._k += 1
        if self.gather:
            self._retire(b)
        if events:
            events[0].record(stream)
        if self.world == 1:
            self.tr.render_dev(scene, g, pc, self.W, self.H, self.locals[b].data_ptr(), camera=camera,
                               hit_ptrs=hp, stream=stream.cuda_stream)
        else:
            self.tr.render_tiled_dev(scene, g, pc, self.W, self.H, self.tiling, self.locals[b].data_ptr(),
                                     camera=camera, hit_ptrs=hp, stream=stream.cuda_stream)
        if events:
            events[1].record(stream)
        if self.mode == "rgba8":
            self.tr.post_dev(self.locals[b].data_ptr(), self.local_pixels, 0, self.sends[b].data_ptr(),
                             stream=stream.cuda_stream)
        if self.gather:
            self._pending[b] = dist.all_gather_into_tensor(self.gathered[b], self.sends[b], async_op=True)
        return self.full

    def finish(self):
        """Drain the pipeline: every frame rendered so far is gathered and assembled."""
        if self.gather:
            for b in range(len(self.locals)):
                self._retire((self._k + b) % len(self.locals))
        return self.full
