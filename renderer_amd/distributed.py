"""Sort-last rendering across ranks (one process per GPU, torch.distributed; the "nccl" backend
is RCCL on ROCm).

The reference is single-GPU; this is the one step the path adds.  Every rank holds the whole
scene, rasterises only its contiguous range of meshlet instances into a full-frame 64-bit
visibility buffer, and the buffers are composited with one 64-bit MIN reduction: min is
associative and commutative and the clear value is its identity, so the result is bit-identical
to the single-GPU buffer.  Keys never set bit 63, hence MIN over int64 (supported by every
backend, gloo included) equals the unsigned 64-bit min.  Each rank then shades H/N rows and the
RGBA8 rows are all-gathered (half the bytes of gathering keys).

A rank only ever reads the composited keys of the rows it shades, so on RCCL the reduction runs
as a reduce-scatter (the first half of a ring all-reduce: (N-1)/N of the buffer per rank instead
of twice that); the full all-reduce stays available (``composite="all_reduce"``) and is what
backends without reduce-scatter (gloo) use.
"""
import torch
import torch.distributed as dist


def instance_range(n_instances, rank, world):
    """Contiguous share of the scene's meshlet instances for ``rank``."""
    return (n_instances * rank) // world, (n_instances * (rank + 1)) // world


def row_range(height, rank, world):
    return (height * rank) // world, (height * (rank + 1)) // world


def _host_staged(t):
    """gloo moves device tensors through the host anyway (and stalls on large ones): for a device
    tensor on that backend the collectives below run on an explicit host copy."""
    return t.is_cuda and dist.get_backend() == "gloo"


def composite_min(vis_i64):
    """In-place MIN all-reduce of a visibility buffer viewed as int64."""
    assert vis_i64.dtype == torch.int64
    if _host_staged(vis_i64):
        host = vis_i64.cpu()  # synchronises with the current stream
        dist.all_reduce(host, op=dist.ReduceOp.MIN)
        vis_i64.copy_(host)
        return vis_i64
    dist.all_reduce(vis_i64, op=dist.ReduceOp.MIN)
    return vis_i64


def composite_min_rows(vis_i64, height, rank, world):
    """MIN reduce-scatter by row strips: afterwards ``vis_i64`` holds the composited keys in this
    rank's rows (the rest of the buffer is not meaningful).  Needs equal strips and a backend
    with reduce-scatter; callers fall back to :func:`composite_min` otherwise."""
    assert vis_i64.dtype == torch.int64 and height % world == 0
    flat = vis_i64.view(-1)
    strip = flat.numel() // world
    # in place: the output is this rank's slice of the input (RCCL's documented in-place form)
    dist.reduce_scatter_tensor(flat[rank * strip:(rank + 1) * strip], flat, op=dist.ReduceOp.MIN)
    return vis_i64


def gather_rows(color, rank, world):
    """All-gather the row strips each rank shaded into the full (H, W, 4) uint8 image, in place."""
    height = color.shape[0]
    bounds = [row_range(height, k, world) for k in range(world)]
    sizes = {hi - lo for lo, hi in bounds}
    lo, hi = bounds[rank]
    if len(sizes) == 1 and dist.get_backend() == "nccl":
        # equal strips: this rank's rows already sit at their offset of the output buffer (RCCL's in-place form)
        try:
            dist.all_gather_into_tensor(color.view(-1), color[lo:hi].reshape(-1))
        except (RuntimeError, ValueError):  # argument check refused the aliasing: same on every rank
            dist.all_gather_into_tensor(color.view(-1), color[lo:hi].reshape(-1).clone())
    elif _host_staged(color):
        parts = [torch.empty((b - a,) + tuple(color.shape[1:]), dtype=color.dtype) for a, b in bounds]
        dist.all_gather(parts, color[lo:hi].cpu())
        for (a, b), part in zip(bounds, parts):
            if a != lo:
                color[a:b].copy_(part)
    else:
        dist.all_gather([color[a:b] for a, b in bounds], color[lo:hi].clone())
    return color


class SortLastFrame:
    """One frame on ``world`` ranks around any renderer exposing set_partition /
    set_shade_rows / render_visbuffer / shade (the HIP ``Renderer`` with device tensors, or the
    CPU oracle with host tensors in the tests)."""

    def __init__(self, renderer, n_instances, height, rank, world, vis_tensor, color_tensor, composite="auto"):
        self.r, self.rank, self.world, self.height = renderer, rank, world, height
        self.vis, self.color = vis_tensor, color_tensor
        if composite == "auto":
            scatter_ok = dist.get_backend() == "nccl" and height % world == 0
            composite = "reduce_scatter" if scatter_ok else "all_reduce"
        assert composite in ("reduce_scatter", "all_reduce")
        self.composite = composite
        renderer.set_partition(*instance_range(n_instances, rank, world))
        renderer.set_shade_rows(*row_range(height, rank, world))
        self._stream_handle = None

    def _follow_current_stream(self):
        """The collectives run on torch's current stream: the renderer's kernels must be ordered on
        the same one (its own private stream would race with them).  Handle 0 is HIP's null stream,
        which the library is told by its explicit handle (hipStreamLegacy = 1)."""
        if not getattr(self.vis, "is_cuda", False) or not hasattr(self.r, "set_stream"):
            return
        handle = torch.cuda.current_stream().cuda_stream or 1
        if handle != self._stream_handle:
            self.r.set_stream(handle)
            self._stream_handle = handle

    def step(self, camera=None, sun=None):
        self._follow_current_stream()
        self.r.render_visbuffer(camera, sun)
        if self.composite == "reduce_scatter":
            try:
                composite_min_rows(self.vis, self.height, self.rank, self.world)
            except (RuntimeError, ValueError):
                # a backend build that refuses the in-place form raises while checking its arguments, on
                # every rank alike and before anything is enqueued: all ranks switch to the all-reduce
                self.composite = "all_reduce"
                composite_min(self.vis)
        else:
            composite_min(self.vis)
        self.r.shade()
        gather_rows(self.color, self.rank, self.world)


class SortLastPipeline:
    """Several frames in flight on one rank: ``depth`` renderers (each with its own buffers and
    its own HIP stream) take turns, so that frame i+1 is rasterised while frame i's keys and
    colour rows are on the links.  The host never waits inside :meth:`step`; every rank issues
    the same collectives in the same order, and a slot's stream orders its reuse.

    ``make_renderer`` builds a HIP ``Renderer`` with the scene already set."""

    def __init__(self, make_renderer, n_instances, height, rank, world, depth=2, composite="auto"):
        self.slots = []
        for _ in range(max(1, depth)):
            r = make_renderer()
            stream = torch.cuda.Stream()
            vis = torch.as_tensor(r.visbuffer_device_array(), device="cuda")
            color = torch.as_tensor(r.color_device_array(), device="cuda")
            self.slots.append((SortLastFrame(r, n_instances, height, rank, world, vis, color, composite), stream))
        self.issued = 0

    @property
    def composite(self):
        """The exchange step in use (a refused reduce-scatter falls back to the all-reduce)."""
        return self.slots[0][0].composite

    def step(self, camera=None, sun=None):
        """Issue one frame; returns its slot's ``SortLastFrame`` (``.color`` is valid once
        ``.stream`` / the device has been synchronised)."""
        frame, stream = self.slots[self.issued % len(self.slots)]
        self.issued += 1
        with torch.cuda.stream(stream):
            frame.step(camera, sun)
        frame.stream = stream
        return frame

    def renderers(self):
        return [f.r for f, _ in self.slots]

    def close(self):
        torch.cuda.synchronize()
        for f, _ in self.slots:
            f.r.close()
