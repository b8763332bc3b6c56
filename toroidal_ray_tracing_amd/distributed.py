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
for one).  How many cycles: when EVERY frame is gathered the step is bound by the collective, not by
the render, so ONE cycle — plain row bands, one large collective per frame, no launch
overhead of eight small ones (xGMI is point-to-point, 7 links per GPU; RCCL spreads a large
collective over all of them); with a gather per batch, or none, DEFAULT_CYCLES (16) cycles: an interleaving
fine enough to balance the load (the torus sits in the middle rows), and the batch's frame is
gathered group by group (DEFAULT_CYCLES groups per rank).

How often: `gather_every` = F gathers only every F-th frame — the frame that leaves the render loop.  The reference's
loop renders 60 frames per camera radius and reads the image back once, after the 60th (BEF/main.cpp:339-343 `counter ==
60 → saveRender`, :384-399 copy + write); replicating every one of them on every GPU would make the step link-bound
by two orders of magnitude for frames nobody takes off the GPU.  `bench.py --gpus N` gathers once per step
(= one batch of 64 frames); `gather_every=1` is the per-frame replication.
"""
import torch
import torch.distributed as dist

from . import abi

DEFAULT_CYCLES = 16         # groups per rank when the render binds the step: a gather per batch, or none (load balance:
                            # the slowest of 8 parts is 91 / 93 / 94 % of linear with 8 / 16 / 32 groups per rank, tools/bench_tiled.py --batch)
DEFAULT_CYCLES_GATHER = 1   # … and with a gather after EVERY frame: a single collective per frame


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


class _NullStream:
    """Stands in for a HIP stream on the CPU (gloo rehearsals and tests): everything is synchronous."""
    cuda_stream = 0

    def wait_event(self, ev):
        pass

    def wait_stream(self, s):
        pass


class TiledFrame:
    """Framebuffer + first-hit streams of one rank, and the per-frame render / per-batch gather step.

    Frames are independent, so K of them can be in flight at once: `streams=K` gives every frame one of K HIP streams
    in turn, each with its own context (tile lists, counters) and its own output set.  A 1/8 part of the baseline frame
    is latency-bound on one stream — one frame's traced tiles do not fill the chip, and the next frame's kernels wait for
    the slowest of them: 41 µs per frame; on 2 / 3 / 4 streams 21.5 / 17.7 / 16.4 µs (tools/bench_tiled_streams.py).

    A gathered frame is first copied (fp32) or tonemapped (rgba8) into one of two staging buffers — 1/N of the frame,
    once per gather — and the all-gathers read the staging copy on the process group's stream: no render stream ever
    waits for a link-bound collective, and the collective of a batch runs behind the whole next batch.  ``finish()``
    drains everything and returns the last complete row-major frame.
    """

    def __init__(self, tracer, W, H, world, rank, device, want_hits=(), group_rows=None,
                 gather=True, force_collective=False, gather_every=1, output_sets=1, batch=1):
        """tracer: one Tracer, or a list of K of them (K render streams, one context each).
        gather: True / "fp32" — all-gather the rgba32f framebuffer (default, what north_star
        prescribes); "rgba8" — tonemap each rank's rows (trt_post_dev, post.frag) and all-gather
        the 8-bit image a swapchain would present: 4x fewer bytes over xGMI, the rgba32f image
        stays sharded; False / "none" — no collective.  force_collective: issue the collective
        even when world == 1 (exercises the RCCL path on a one-GPU box).  gather_every: F > 1 gathers only every F-th
        frame (frames F-1, 2F-1, …).  output_sets: M > K rotates the frames over M output sets (rounded up to a multiple
        of the K streams, so that a set always belongs to one stream) — consecutive frames then never write the same
        buffers and nothing of a frame is still in the 256-MB Infinity Cache when it is written again (bench.py's
        roofline pass).  batch: B > 1 collects B consecutive frames and renders them with ONE pair of launches
        (trt_render_batch_dev): B parts of a frame are the work of B/N full frames — a launch that fills the chip where a
        single 1/N part does not.  With K tracers the launches take the K streams in turn, so that the classification of
        one batch runs beside the render kernel of the batch before.  The output sets are then at least K·B."""
        self.trs = list(tracer) if isinstance(tracer, (list, tuple)) else [tracer]
        self.tr = self.trs[0]
        self.W, self.H, self.world, self.rank = W, H, world, rank
        mode = {True: "fp32", False: "none"}.get(gather, gather)
        if mode not in ("fp32", "rgba8", "none"):
            raise ValueError(f"gather={gather!r}")
        self.mode = mode if (world > 1 or force_collective) else "none"
        self.gather = self.mode != "none"
        self.gather_every = max(1, int(gather_every))
        # per-frame gathers: the step is bound by the collective — one large one per frame; gathers once per batch (or none):
        # the step is bound by the render — interleaved groups balance it, and the batch's frame is gathered group by group
        per_frame = self.gather and self.gather_every == 1
        self.group_rows = group_rows or default_group_rows(H, world, DEFAULT_CYCLES_GATHER if per_frame else DEFAULT_CYCLES)
        if world > 1 and H % (self.group_rows * world) != 0:
            raise ValueError(f"H={H} must be a multiple of group_rows*world={self.group_rows * world}")
        self.cycles = H // (self.group_rows * world) if world > 1 else 1
        self.tiling = abi.trt_tiling(self.group_rows, world, rank, 1 if world > 1 else 0)
        self.local_rows = self.tr.tiling_rows(self.tiling, H) if world > 1 else H
        self.local_pixels = self.local_rows * W
        self.device = torch.device(device)
        K = len(self.trs)
        self.batch = max(1, int(batch))
        if self.batch > abi.TRT_MAX_BATCH:
            raise ValueError(f"batch={self.batch} > TRT_MAX_BATCH")
        self._queued = []          # batch mode: (g, pc, output set, gather?) of the frames not yet launched
        self._flushes = 0          # batch mode: launches since the last join() — launch j goes to stream j % K
        f32 = dict(dtype=torch.float32, device=device)
        # one output set per frame in flight (K streams x B frames per launch), or more (output_sets): frame i renders into
        # set i % n_sets on stream (i // B) % K; n_sets is a multiple of K·B, so a set is only ever written on one stream
        unit = K * self.batch
        self.n_sets = ((max(unit, int(output_sets)) + unit - 1) // unit) * unit
        self.locals = [torch.empty(self.local_rows, W, 4, **f32) for _ in range(self.n_sets)]
        self.hit_sets = []
        for _ in range(self.n_sets):
            h = {k: torch.empty(self.local_pixels, **f32) for k in want_hits if k != "id"}
            if "id" in want_hits:
                h["id"] = torch.empty(self.local_pixels, dtype=torch.int32, device=device)
            self.hit_sets.append(h)
        self.hits = self.hit_sets[0]
        # K > 1: the render streams are this object's own; K == 1: the caller's stream is the render stream
        cuda = self.device.type == "cuda"
        self._own = [torch.cuda.Stream(device=self.device) if cuda else _NullStream() for _ in range(K)] if K > 1 else None
        self._forked = False
        self._caller = None
        # two staging buffers and two gathered frames: gather j+1 is issued while a consumer may still read frame j
        gdt = dict(dtype=torch.uint8 if self.mode == "rgba8" else torch.float32, device=device)
        self.sends = [torch.empty(self.local_rows, W, 4, **gdt) for _ in range(2)] if self.gather else []
        self.fulls = [torch.empty(H, W, 4, **gdt) for _ in range(2)] if self.gather else self.locals
        self._pending = [[], []]   # in-flight all-gathers reading staging buffer j
        self._stage = 0            # staging buffer of the next gather
        self._k = 0                # frames rendered
        self._last = 0             # gathered frame (or, without a gather, output set) that finish() hands out
        self._newest = None        # gathered frame of the most recent gather (in flight until finish())

    @property
    def local(self):
        return self.locals[0]

    @property
    def full(self):
        """The frame finish() returned last: the last GATHERED one (without a gather: the last one rendered).  Valid only
        between finish() and the next render(): while frames are in flight both gathered frames may be the target of
        a collective that has not been waited for."""
        return self.fulls[self._last]

    def describe(self):
        if self.world == 1 and not self.gather:
            return "single GPU, full frame"
        what = {"fp32": "all_gather_into_tensor(rgba32f)", "rgba8": "post pass + all_gather_into_tensor(rgba8)"}.get(self.mode)
        when = "per frame" if self.gather_every == 1 else f"every {self.gather_every} frames (the frame that leaves the loop)"
        behind = "the next frame" if self.gather_every == 1 else "the next batch"
        tail = (f"{self.cycles} x {what} {when}, each landing in place in the row-major frame, pipelined behind {behind}"
                if self.gather else "no gather")
        k = f"; {len(self.trs)} frames in flight on {len(self.trs)} streams" if len(self.trs) > 1 else ""
        if self.batch > 1:
            k = f"; {self.batch} frames per launch (trt_render_batch_dev)" + (f", launches alternating over {len(self.trs)} streams" if len(self.trs) > 1 else "")
        return f"{self.world} ranks x {self.local_rows} rows in interleaved groups of {self.group_rows}; {tail}{k}"

    def _on(self, stream):
        import contextlib
        return torch.cuda.stream(stream) if self.device.type == "cuda" else contextlib.nullcontext()

    def _retire(self, j):
        """Order the current stream behind the gathers that read staging buffer j (and wrote gathered frame j)."""
        if self._pending[j]:
            for w in self._pending[j]:
                w.wait()
            self._pending[j] = []

    def render(self, scene, g, pc, camera, stream, events=None):
        """One frame: render this rank's rows on the frame's stream (the caller's `stream`, or with K tracers the next of
        the K own streams, which branch off `stream` at the first frame after construction / join()); on a gather frame
        copy or tonemap the rows into a staging buffer and start the all-gathers behind that.
        `events` = (start, end) torch.cuda.Events recorded around the render launches only.
        Returns nothing: a frame is handed out by finish() (with frames and collectives in flight neither gathered
        frame is stable)."""
        K = len(self.trs)
        k = self._k % K
        o = self._k % self.n_sets   # output set; n_sets is a multiple of K (K·B in batch mode), so set o is only ever written on one stream
        self._k += 1
        self._caller = stream
        if self.batch > 1:
            self._camera, self._scene = camera, scene
            self._queued.append((g, pc, o, self.gather and self._k % self.gather_every == 0))
            if len(self._queued) == self.batch:
                self.flush(stream)
            if not self.gather:
                self._last = o
            return
        if self._own is not None:
            if not self._forked:
                if self.device.type == "cuda":
                    for s in self._own:
                        s.wait_stream(stream)
                self._forked = True
            s = self._own[k]
        else:
            s = stream
        tr, hp = self.trs[k], {n: v.data_ptr() for n, v in self.hit_sets[o].items()}
        do_gather = self.gather and self._k % self.gather_every == 0
        if events:
            events[0].record(s)
        if self.world == 1:
            tr.render_dev(scene, g, pc, self.W, self.H, self.locals[o].data_ptr(), camera=camera, hit_ptrs=hp, stream=s.cuda_stream)
        else:
            tr.render_tiled_dev(scene, g, pc, self.W, self.H, self.tiling, self.locals[o].data_ptr(), camera=camera,
                                hit_ptrs=hp, stream=s.cuda_stream)
        if events:
            events[1].record(s)
        if do_gather:
            self._gather(o, s, tr)
        if not self.gather:
            self._last = o

    def flush(self, stream=None):
        """Batch mode: launch the frames collected so far (a full batch launches by itself)."""
        stream = stream or self._caller
        if not self._queued:
            return
        q, self._queued = self._queued, []
        kb = self._flushes % len(self.trs)   # (full batches between two join()s: n_sets / B is a multiple of K, a set keeps its stream)
        self._flushes += 1
        tr = self.trs[kb]
        if self._own is not None:
            if not self._forked:
                if self.device.type == "cuda":
                    for s in self._own:
                        s.wait_stream(stream)
                self._forked = True
            stream = self._own[kb]
        frames = [(g, pc, self.locals[o].data_ptr(), {n: v.data_ptr() for n, v in self.hit_sets[o].items()}) for g, pc, o, _ in q]
        tr.render_batch_dev(self._scene, frames, self.W, self.H, self.tiling if self.world > 1 else None, camera=self._camera,
                            stream=stream.cuda_stream)
        for _, _, o, do_gather in q:
            if do_gather:
                self._gather(o, stream, tr)

    def _gather(self, o, s, tr):
        """Copy or tonemap output set o into a staging buffer on stream s and start the all-gathers behind that."""
        j = self._stage
        self._stage ^= 1
        with self._on(s):
            self._retire(j)   # the gather that read this staging buffer two gathers ago is complete before it is overwritten
            if self.mode == "rgba8":
                tr.post_dev(self.locals[o].data_ptr(), self.local_pixels, 0, self.sends[j].data_ptr(), stream=s.cuda_stream)
            else:
                self.sends[j].copy_(self.locals[o], non_blocking=True)
            G, span = self.group_rows if self.world > 1 else self.H, (self.group_rows * self.world if self.world > 1 else self.H)
            for c in range(self.cycles):
                self._pending[j].append(dist.all_gather_into_tensor(
                    self.fulls[j][c * span:(c + 1) * span], self.sends[j][c * G:(c + 1) * G], async_op=True))
        self._newest = j

    # -- a whole step as ONE hipGraph ------------------------------------------------------------------------------
    def capture_step(self, scene, g, pc, camera, stream, n_frames):
        """Capture `n_frames` consecutive frames — K streams, their fork and join, two kernel nodes per frame — into one
        hipGraph; step() then replays it with ONE host call instead of 2·n_frames launches through Python and the C ABI
        (a 1/8 part of the baseline frame takes the GPU ≈15 µs; enqueueing it eagerly takes the host as long).  The
        *_dev entry points are capturable once the contexts' scratch is sized (include/trt.h): render one eager frame
        per stream first.  The frames of a step must map to streams and output sets the same way in every step:
        n_frames must be a multiple of the number of output sets and the frame counter must stand at a multiple of it
        (restart()).  A gather (every gather_every-th frame) stays outside the graph: step() issues it eagerly behind
        the replay, and it may only fall on a step's last frame."""
        if self.device.type != "cuda":
            raise RuntimeError("capture_step needs a GPU")
        if n_frames % self.n_sets or self._k % self.n_sets:
            raise ValueError(f"a captured step must cover whole rounds of the {self.n_sets} output sets (n_frames={n_frames}, frame counter {self._k})")
        if self.gather and self.gather_every % n_frames:
            raise ValueError("a gather may only fall on the last frame of a captured step (gather_every must be a multiple of n_frames)")
        self.join(stream)
        K = len(self.trs)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(stream)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side), torch.cuda.graph(graph, stream=side):
            if self._own is not None:
                for s in self._own:
                    s.wait_stream(side)
            if self.batch > 1:
                if n_frames % self.batch:
                    raise ValueError("a captured step must consist of whole batches")
                self._caller = side
                for f in range(n_frames):
                    self._queued.append((g, pc, f % self.n_sets, False))
                    self._camera, self._scene = camera, scene
                    if len(self._queued) == self.batch:
                        self.flush(side)
            for f in range(n_frames if self.batch == 1 else 0):
                k, o = f % K, f % self.n_sets
                s = self._own[k] if self._own is not None else side
                hp = {n: v.data_ptr() for n, v in self.hit_sets[o].items()}
                if self.world == 1:
                    self.trs[k].render_dev(scene, g, pc, self.W, self.H, self.locals[o].data_ptr(), camera=camera, hit_ptrs=hp, stream=s.cuda_stream)
                else:
                    self.trs[k].render_tiled_dev(scene, g, pc, self.W, self.H, self.tiling, self.locals[o].data_ptr(), camera=camera,
                                                 hit_ptrs=hp, stream=s.cuda_stream)
            if self._own is not None:
                for s in self._own:
                    side.wait_stream(s)
        stream.wait_stream(side)
        self._forked, self._flushes = False, 0   # (the capture forked and joined the own streams itself)
        self._graph, self._graph_frames = graph, n_frames
        return graph

    def step(self, stream):
        """Replay the captured step on `stream` (which must be torch's current stream); on a gather step, start the
        gather of the step's last frame behind it."""
        n = self._graph_frames
        self._caller = stream
        self._graph.replay()
        self._k += n
        o = (self._k - 1) % self.n_sets
        if self.gather and self._k % self.gather_every == 0:
            self._gather(o, stream, self.trs[((n - 1) // self.batch) % len(self.trs)])
        if not self.gather:
            self._last = o

    def join(self, stream=None):
        """K own streams: order `stream` (default: the stream of the last render call) behind every frame issued so far;
        the next render() branches off again."""
        stream = stream or self._caller
        if self.batch > 1 and stream is not None:
            self.flush(stream)
        self._flushes = 0
        if self._own is not None and self._forked and stream is not None:
            if self.device.type == "cuda":
                for s in self._own:
                    stream.wait_stream(s)
            self._forked = False

    def restart(self):
        """Drain the pipeline and start counting frames anew: the next gather is that of frame `gather_every` from here."""
        self.finish()
        self._k = 0

    def finish(self):
        """Drain the pipeline: every frame issued so far is rendered, every gather started is complete on the caller's
        stream; returns the last gathered frame."""
        self.join()
        if self.gather:
            with self._on(self._caller) if self._caller is not None else self._on(None):
                for i in range(2):
                    self._retire(self._stage ^ i)   # oldest gather first
            if self._newest is not None:
                self._last = self._newest           # every gather has been waited for: the newest frame is complete
        return self.full
