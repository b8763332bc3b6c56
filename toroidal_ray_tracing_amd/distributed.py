"""Multi-GPU image tiling: one process per GPU, rows dealt round-robin in groups, RCCL
all-gather of the rgba32f framebuffer (BASELINE.json north_star).

The reference is single-GPU (``compatibleDevices[0]``, ray_tracing_reflections/main.cpp:143);
pixels are independent (one raygen invocation per pixel, no inter-pixel communication in
raytrace.rgen), so the frame shards by rows with no exchange during the render.  The only
collective is the all-gather of the finished framebuffer, and only when world_size > 1.

Layout.  With N ranks and groups of G rows, rank p owns the rows y with (y // G) % N == p and
renders them into a COMPACT local buffer [H/N, W, 4] (``trt_tiling.compact = 1``).  One *cycle*
of the tiling is N consecutive groups = G·N consecutive image rows, group p of it coming from
rank p — so the all-gather of the ranks' c-th groups, ``all_gather_into_tensor(frame[c·G·N :
(c+1)·G·N], local[c·G : (c+1)·G])``, lands every row AT ITS PLACE in the row-major frame: the
gathered frame needs no de-interleaving copy (round 1 paid 2 × 268 MB of HBM traffic per frame
for one).  How many cycles: when the frame is gathered the step is bound by the collective, not by
the render, so ONE cycle — plain row bands, one large collective per frame, no launch
overhead of eight small ones (xGMI is point-to-point, 7 links per GPU; RCCL spreads a large
collective over all of them); without a gather (`gather="none"`) eight cycles, an interleaving
fine enough to balance the load (the torus sits in the middle rows).

How often: `gather_every` = F gathers only every F-th frame — the frame that leaves the render loop.  The reference's
loop renders 60 frames per camera radius and reads the image back once, after the 60th (BEF/main.cpp:339-343 `counter ==
60 → saveRender`, :384-399 copy + write); replicating every one of them on every GPU would make the step link-bound
by two orders of magnitude for frames nobody takes off the GPU.  `bench.py --gpus N` gathers once per step
(= one batch of 64 frames); `gather_every=1` is the per-frame replication.
"""
import torch
import torch.distributed as dist

from . import abi

DEFAULT_CYCLES = 8          # groups per rank without a gather (load balance)
DEFAULT_CYCLES_GATHER = 1   # … and with one: a single collective per frame


def default_group_rows(H, world, cycles=DEFAULT_CYCLES):
    """Largest number of cycles <= `cycles` for which the groups are whole 8-row tile bands
    (tile culling needs groups of 8k rows); falls back to one band per rank."""
    if world <= 1:
        return H
    for c in range(cycles, 0, -1):
        if H % (world * c) == 0 and (H // (world * c)) % 8 == 0:
            return H // (world * c)
    for c in range(cycles, 0, -1):
        if H % (world * c) == 0:
            return H // (world * c)
    raise ValueError(f"H={H} is not a multiple of the number of ranks {world}")


def owned_rows(H, group_rows, n_parts, part):
    """Image rows owned by `part`, in local-buffer order (pure-Python mirror of trt_tiling)."""
    return [y for y in range(H) if (y // group_rows) % n_parts == part]


def deinterleave(gathered, H, W, group_rows, n_parts, channels=4):
    """[N, H/N, W, C] stacked compact buffers → row-major [H, W, C] image (a strided view).
    Not used by TiledFrame any more (its gathers land in place); kept for callers that stack
    compact parts themselves (tests).  Requires H % (group_rows*n_parts) == 0."""
    cycles = H // (group_rows * n_parts)
    v = gathered.view(n_parts, cycles, group_rows, W, channels)
    return v.permute(1, 0, 2, 3, 4).reshape(H, W, channels)


class TiledFrame:
    """Framebuffer + first-hit streams of one rank, and the per-frame render/gather step.

    With world > 1 the step is software-pipelined over two buffer sets: the all-gathers of
    frame k (RCCL, asynchronous on the process group's stream) run while frame k+1 renders, so
    the frame rate is max(render, gather) instead of their sum; ``finish()`` drains the pipe and
    returns the last complete row-major frame.
    """

    def __init__(self, tracer, W, H, world, rank, device, want_hits=(), group_rows=None,
                 gather=True, force_collective=False, gather_every=1):
        """gather: True / "fp32" — all-gather the rgba32f framebuffer (default, what north_star
        prescribes); "rgba8" — tonemap each rank's rows (trt_post_dev, post.frag) and all-gather
        the 8-bit image a swapchain would present: 4x fewer bytes over xGMI, the rgba32f image
        stays sharded; False / "none" — no collective.  force_collective: issue the collective
        even when world == 1 (exercises the RCCL path on a one-GPU box).  gather_every: F > 1 gathers only every F-th
        frame (frames F-1, 2F-1, …); the frames in between render into the same buffer set, and the sets swap at each
        gather, so the collective of a batch runs behind the whole next batch."""
        self.tr, self.W, self.H, self.world, self.rank = tracer, W, H, world, rank
        mode = {True: "fp32", False: "none"}.get(gather, gather)
        if mode not in ("fp32", "rgba8", "none"):
            raise ValueError(f"gather={gather!r}")
        self.mode = mode if (world > 1 or force_collective) else "none"
        self.gather = self.mode != "none"
        # per-frame gathers: the step is bound by the collective — one large one per frame; gathers once per batch (or none):
        # the step is bound by the render — interleaved groups balance it, and the batch's frame is gathered group by group
        per_frame = self.gather and max(1, int(gather_every)) == 1
        self.group_rows = group_rows or default_group_rows(H, world, DEFAULT_CYCLES_GATHER if per_frame else DEFAULT_CYCLES)
        if world > 1 and H % (self.group_rows * world) != 0:
            raise ValueError(f"H={H} must be a multiple of group_rows*world={self.group_rows * world}")
        self.cycles = H // (self.group_rows * world) if world > 1 else 1
        self.tiling = abi.trt_tiling(self.group_rows, world, rank, 1 if world > 1 else 0)
        self.local_rows = tracer.tiling_rows(self.tiling, H) if world > 1 else H
        self.local_pixels = self.local_rows * W
        f32 = dict(dtype=torch.float32, device=device)
        nbuf = 2 if self.gather else 1
        self.locals = [torch.empty(self.local_rows, W, 4, **f32) for _ in range(nbuf)]
        self.hits = {k: torch.empty(self.local_pixels, **f32) for k in want_hits if k != "id"}
        if "id" in want_hits:
            self.hits["id"] = torch.empty(self.local_pixels, dtype=torch.int32, device=device)
        gdt = dict(dtype=torch.uint8 if self.mode == "rgba8" else torch.float32, device=device)
        self.sends = [torch.empty(self.local_rows, W, 4, **gdt) for _ in range(nbuf)] if self.mode == "rgba8" else self.locals
        # the gathered frame, row-major [H, W, 4] — one per buffer set (frame k+1 is gathered while
        # a consumer may still read frame k)
        self.fulls = [torch.empty(H, W, 4, **gdt) for _ in range(nbuf)] if self.gather else self.locals
        self._pending = [[] for _ in range(nbuf)]   # in-flight all-gathers of each buffer set
        self.gather_every = max(1, int(gather_every))
        self._k = 0      # frames rendered
        self._cur = 0    # buffer set the next frame renders into
        self._last = 0   # buffer set of the most recent gathered (or, without a gather, rendered) frame

    @property
    def local(self):
        return self.locals[0]

    @property
    def full(self):
        """The most recent complete frame (after ``finish()``: the last one rendered)."""
        return self.fulls[self._last]

    def describe(self):
        if self.world == 1 and not self.gather:
            return "single GPU, full frame"
        what = {"fp32": "all_gather_into_tensor(rgba32f)", "rgba8": "post pass + all_gather_into_tensor(rgba8)"}.get(self.mode)
        when = "per frame" if self.gather_every == 1 else f"every {self.gather_every} frames (the frame that leaves the loop)"
        behind = "the next frame" if self.gather_every == 1 else "the next batch"
        tail = (f"{self.cycles} x {what} {when}, each landing in place in the row-major frame, pipelined behind {behind}"
                if self.gather else "no gather")
        return f"{self.world} ranks x {self.local_rows} rows in interleaved groups of {self.group_rows}; {tail}"

    def _retire(self, b):
        """Wait for the gathers of buffer set b: its row-major frame is then complete."""
        if self._pending[b]:
            for w in self._pending[b]:
                w.wait()   # orders the current stream behind the collective
            self._pending[b] = []
            self._last = b

    def render(self, scene, g, pc, camera, stream, events=None):
        """One frame: render this rank's rows; when gathering, start the all-gathers of this frame
        and retire the frame that used this buffer set two steps ago.
        `events` = (start, end) torch.cuda.Events recorded around the render launches only."""
        hp = {k: v.data_ptr() for k, v in self.hits.items()}
        b = self._cur
        self._k += 1
        do_gather = self.gather and self._k % self.gather_every == 0
        if self.gather:
            self._retire(b)   # the gather that last read this set (two gathers ago) is complete before its rows are overwritten
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
        if self.mode == "rgba8" and do_gather:
            self.tr.post_dev(self.locals[b].data_ptr(), self.local_pixels, 0, self.sends[b].data_ptr(),
                             stream=stream.cuda_stream)
        if do_gather:
            G, span = self.group_rows if self.world > 1 else self.H, (self.group_rows * self.world if self.world > 1 else self.H)
            for c in range(self.cycles):
                self._pending[b].append(dist.all_gather_into_tensor(
                    self.fulls[b][c * span:(c + 1) * span], self.sends[b][c * G:(c + 1) * G], async_op=True))
            self._cur = (b + 1) % len(self.locals)
        elif not self.gather:
            self._last = b
        return self.fulls[b]

    def restart(self):
        """Drain the pipeline and start counting frames anew: the next gather is that of frame `gather_every` from here."""
        self.finish()
        self._k = 0

    def finish(self):
        """Drain the pipeline: every frame rendered so far is gathered; returns the last one."""
        if self.gather:
            for i in range(len(self.locals)):
                self._retire((self._cur + i) % len(self.locals))   # oldest gather first: the newest one sets `full`
        return self.full
