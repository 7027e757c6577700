"""Sort-last rendering across ranks (one process per GPU, torch.distributed; the "nccl" backend
is RCCL on ROCm).

The reference is single-GPU; this is the one step the path adds.  Every rank holds the whole
scene, rasterises only its share of the meshlet instances into a full-frame 64-bit
visibility buffer, and the buffers are composited with one 64-bit MIN reduction: min is
associative and commutative and the clear value is its identity, so the result is bit-identical
to the single-GPU buffer.  Keys never set bit 63, hence MIN over int64 (supported by every
backend, gloo included) equals the unsigned 64-bit min.  Each rank then shades H/N rows and the
RGBA8 rows are gathered (half the bytes of gathering keys).

Two ways to split the work (``split``):

``interleaved``  (default) every draw's meshlets in chunks of ``chunk`` dealt round-robin over the ranks,
                 rotated by the draw index, and the frame's rows in bands of ``band`` rows dealt round-robin.
                 Meshlet-instance ids follow the draw order, which for an instanced scene is a spatial order:
                 dealing small pieces out evens both the raster and the shading work between the ranks.
                 Draws made of a few screen-filling triangles (``replicate_pixels``: mean projected triangle
                 area) are drawn by every rank, each on the rows it shades: their pixel work cannot be dealt
                 out by meshlets.
``contiguous``   rank k takes the k-th N-th of the meshlet-instance ids and the k-th N-th of the rows
                 (on the bench field at two ranks: 72 % of the fragments on rank 0, 88 % of the covered
                 pixels on rank 1).

Three forms of the exchange step, chosen once when a frame object is built:

``p2p``             the library's own peer-to-peer exchange (``rdr_composite_*``): each rank reads the keys of
                    its rows straight out of every peer's buffer over xGMI and takes the minimum, then writes
                    its shaded rows straight into every peer's image.  xGMI is point to point (seven links per
                    GPU), so this uses all links at once where a ring moves every byte over one link per hop.
``reduce_scatter``  RCCL MIN reduce-scatter by row strips + all-gather of the rows.
``all_reduce``      RCCL / gloo MIN all-reduce of the whole buffer + all-gather (the only form gloo offers).

``auto`` takes ``p2p`` when every rank could open every peer's buffers and a probe frame composited that way
equals the collective's result on every rank, else ``reduce_scatter`` where the backend has it, else ``all_reduce``.
"""
import torch
import torch.distributed as dist


def instance_range(n_instances, rank, world):
    """Contiguous share of the scene's meshlet instances for ``rank``."""
    return (n_instances * rank) // world, (n_instances * (rank + 1)) // world


def row_range(height, rank, world):
    return (height * rank) // world, (height * (rank + 1)) // world


DEFAULT_CHUNK = 64  # meshlets per dealt chunk: one cull command
DEFAULT_BAND = 32   # rows per dealt band: one tile row of the peer-to-peer exchange
# rdr_set_partition_replication's threshold (mean projected triangle area above which a draw is drawn by every rank,
# on its own rows).  Off by default: at 3840x2160 it evens the ranks' geometry time out (interior scene, 8 ranks:
# max / mean 1.45 -> 1.07) but does not shorten the slowest rank, because the tile path a rank then runs is a chain
# of dependent steps of about 0.1 ms whatever its share of the pixels (profiles/r3/rank_alone_*.txt).
DEFAULT_REPLICATE_PIXELS = 0.0


def band_spans(height, band, rank, world):
    """Row spans [(lo, hi), ...] of ``rank`` when bands of ``band`` rows are dealt round-robin."""
    return [(lo, min(lo + band, height)) for lo in range(rank * band, height, band * world)]


def band_row_index(height, band, rank, world):
    """The same rows as one int64 index tensor."""
    spans = band_spans(height, band, rank, world)
    if not spans:
        return torch.empty(0, dtype=torch.int64)
    return torch.cat([torch.arange(lo, hi, dtype=torch.int64) for lo, hi in spans])


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
    with reduce-scatter."""
    assert vis_i64.dtype == torch.int64 and height % world == 0
    flat = vis_i64.view(-1)
    strip = flat.numel() // world
    # in place: the output is this rank's slice of the input (RCCL's documented in-place form)
    dist.reduce_scatter_tensor(flat[rank * strip:(rank + 1) * strip], flat, op=dist.ReduceOp.MIN)
    return vis_i64


def gather_rows(color, rank, world, in_place=True):
    """All-gather the row strips each rank shaded into the full (H, W, 4) uint8 image."""
    height = color.shape[0]
    bounds = [row_range(height, k, world) for k in range(world)]
    sizes = {hi - lo for lo, hi in bounds}
    lo, hi = bounds[rank]
    if len(sizes) == 1 and dist.get_backend() == "nccl":
        # equal strips: this rank's rows already sit at their offset of the output buffer (RCCL's in-place form)
        mine = color[lo:hi].reshape(-1)
        dist.all_gather_into_tensor(color.view(-1), mine if in_place else mine.clone())
    elif _host_staged(color):
        parts = [torch.empty((b - a,) + tuple(color.shape[1:]), dtype=color.dtype) for a, b in bounds]
        dist.all_gather(parts, color[lo:hi].cpu())
        for (a, b), part in zip(bounds, parts):
            if a != lo:
                color[a:b].copy_(part)
    else:
        dist.all_gather([color[a:b] for a, b in bounds], color[lo:hi].clone())
    return color


def composite_min_bands(vis_i64, width, row_index, rank, world):
    """MIN reduce-scatter when the ranks' rows are interleaved bands: the rows are packed rank-major into a
    scratch tensor (ranks with fewer rows padded with the identity), reduce-scattered, and this rank's rows
    written back into ``vis_i64``.  ``row_index``: every rank's rows (band_row_index), on the tensor's device."""
    rows = vis_i64.view(-1, width)
    n_max = max(int(i.numel()) for i in row_index)
    packed = torch.full((world, n_max, width), torch.iinfo(torch.int64).max, dtype=torch.int64, device=vis_i64.device)
    for k, idx in enumerate(row_index):
        packed[k, :idx.numel()] = rows.index_select(0, idx)
    out = torch.empty((n_max, width), dtype=torch.int64, device=vis_i64.device)
    dist.reduce_scatter_tensor(out, packed, op=dist.ReduceOp.MIN)
    mine = row_index[rank]
    rows.index_copy_(0, mine, out[:mine.numel()])
    return vis_i64


def gather_bands(color, row_index, rank, world):
    """All-gather of interleaved row bands into the full (H, W, 4) uint8 image: packed, padded to the largest
    share, gathered, and the other ranks' rows copied into place."""
    n_max = max(int(i.numel()) for i in row_index)
    staged = _host_staged(color)
    src = color.cpu() if staged else color
    index = [i.cpu() for i in row_index] if staged else row_index
    mine = torch.zeros((n_max,) + tuple(color.shape[1:]), dtype=color.dtype, device=src.device)
    mine[:index[rank].numel()] = src.index_select(0, index[rank])
    if dist.get_backend() == "nccl":
        parts = torch.empty((world, n_max) + tuple(color.shape[1:]), dtype=color.dtype, device=src.device)
        dist.all_gather_into_tensor(parts, mine)
    else:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
    for k in range(world):
        if k != rank and index[k].numel():
            src.index_copy_(0, index[k], parts[k][:index[k].numel()])
    if staged:
        color.copy_(src)
    return color


def _all_agree(ok):
    """True when ``ok`` holds on every rank (a MIN all-reduce of one flag)."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def probe_collectives(device_tensors, height, world):
    """Which in-place collective forms this backend build accepts, found once on scratch tensors (the frame loop
    must not discover a refused form while other ranks have already enqueued theirs).  Returns
    (reduce_scatter_ok, in_place_gather_ok); every rank gets the same answer."""
    if dist.get_backend() != "nccl" or not device_tensors or height % world:
        return False, False
    rank = dist.get_rank()
    scratch = torch.zeros(world * 16, dtype=torch.int64, device="cuda")
    try:
        dist.reduce_scatter_tensor(scratch[rank * 16:(rank + 1) * 16], scratch, op=dist.ReduceOp.MIN)
        torch.cuda.synchronize()
        scatter = True
    except (RuntimeError, ValueError):
        scatter = False
    scatter = _all_agree(scatter)
    img = torch.zeros(world * 16, dtype=torch.uint8, device="cuda")
    try:
        dist.all_gather_into_tensor(img, img[rank * 16:(rank + 1) * 16])
        torch.cuda.synchronize()
        gather = True
    except (RuntimeError, ValueError):
        gather = False
    return scatter, _all_agree(gather)


class SortLastFrame:
    """One frame on ``world`` ranks around any renderer exposing set_partition /
    set_shade_rows / render_visbuffer / shade (the HIP ``Renderer`` with device tensors, or the
    CPU oracle with host tensors in the tests)."""

    def __init__(self, renderer, n_instances, height, rank, world, vis_tensor, color_tensor, composite="auto",
                 split="interleaved", chunk=DEFAULT_CHUNK, band=DEFAULT_BAND, replicate_pixels=DEFAULT_REPLICATE_PIXELS):
        self.r, self.rank, self.world, self.height = renderer, rank, world, height
        self.vis, self.color = vis_tensor, color_tensor
        assert composite in ("auto", "p2p", "reduce_scatter", "all_reduce")
        assert split in ("interleaved", "contiguous")
        self.split, self.band = split, band
        device = bool(getattr(vis_tensor, "is_cuda", False))
        if split == "interleaved":
            renderer.set_partition(0, n_instances)
            renderer.set_partition_interleaved(chunk, rank, world)
            renderer.set_shade_bands(band, rank, world)
            if hasattr(renderer, "set_partition_replication"):
                renderer.set_partition_replication(replicate_pixels)
            self.rows = band_row_index(height, band, rank, world)  # this rank's rows (host tensor)
            self._row_index = [band_row_index(height, band, k, world).to(vis_tensor.device) for k in range(world)]
        else:
            renderer.set_partition_interleaved(DEFAULT_CHUNK, 0, 1)
            if hasattr(renderer, "set_partition_replication"):
                renderer.set_partition_replication(0.0)
            renderer.set_partition(*instance_range(n_instances, rank, world))
            renderer.set_shade_rows(*row_range(height, rank, world))
            self.rows = torch.arange(*row_range(height, rank, world), dtype=torch.int64)
            self._row_index = None
        self._stream_handle = None
        self.timing = None  # list of four events per frame once time_composite() was called
        # the in-place collective forms need equal contiguous strips; banded rows go through packed scratch tensors
        scatter_ok, self._gather_in_place = probe_collectives(device, height if split == "contiguous" else world, world)
        self._p2p = False
        if composite in ("auto", "p2p") and device and hasattr(renderer, "composite_export"):
            self._p2p = self._connect_p2p()
            if composite == "p2p" and not self._p2p:
                raise RuntimeError("composite='p2p': the ranks could not open each other's buffers")
        if self._p2p:
            self.composite = "p2p"
        elif composite in ("auto", "p2p", "reduce_scatter") and scatter_ok:
            self.composite = "reduce_scatter"
        else:
            self.composite = "all_reduce"

    def _connect_p2p(self):
        """Exchange IPC handles and open every peer's buffers; all ranks take the same decision."""
        handles, ok = [None] * self.world, True
        try:
            mine = self.r.composite_export()
        except Exception:
            mine, ok = b"", False
        dist.all_gather_object(handles, mine)
        ok = ok and all(len(h) == len(mine) and len(h) > 0 for h in handles)
        if _all_agree(ok):
            try:
                self.r.composite_connect(self.rank, self.world, handles)
            except Exception:
                ok = False
            if _all_agree(ok):
                return True
            try:
                self.r.composite_disconnect()
            except Exception:
                pass
        return False

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

    def time_composite(self):
        """Start recording device-side time stamps around the exchange and the gather of every frame."""
        self.timing = []

    def composite_ms(self):
        """Mean device time of (key exchange, row gather) per recorded frame; synchronises."""
        if not self.timing:
            return None
        torch.cuda.synchronize()
        ex = sum(a.elapsed_time(b) for a, b, _, _ in self.timing) / len(self.timing)
        ga = sum(c.elapsed_time(d) for _, _, c, d in self.timing) / len(self.timing)
        self.timing = []
        return ex, ga

    def step(self, camera=None, sun=None):
        self._follow_current_stream()
        ev = None
        if self.timing is not None and getattr(self.vis, "is_cuda", False):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        self.r.render_visbuffer(camera, sun)
        if ev:
            ev[0].record()
        if self.composite == "p2p":
            self.r.composite_exchange()
        elif self.composite == "reduce_scatter" and self._row_index is not None:
            composite_min_bands(self.vis, self.vis.numel() // self.height, self._row_index, self.rank, self.world)
        elif self.composite == "reduce_scatter":
            composite_min_rows(self.vis, self.height, self.rank, self.world)
        else:
            composite_min(self.vis)
        if ev:
            ev[1].record()
        self.r.shade()
        if ev:
            ev[2].record()
        if self.composite == "p2p":
            self.r.composite_gather()
        elif self._row_index is not None:
            gather_bands(self.color, self._row_index, self.rank, self.world)
        else:
            gather_rows(self.color, self.rank, self.world, self._gather_in_place)
        if ev:
            ev[3].record()
            self.timing.append(tuple(ev))

    def close(self):
        if self._p2p:
            if getattr(self.vis, "is_cuda", False):
                torch.cuda.synchronize()
            dist.barrier()  # no rank unmaps buffers a peer may still be reading
            self.r.composite_disconnect()
            self._p2p = False


class SortLastPipeline:
    """Several frames in flight on one rank: ``depth`` renderers (each with its own buffers and
    its own HIP stream) take turns, so that frame i+1 is rasterised while frame i's keys and
    colour rows are on the links.  The host never waits inside :meth:`step`; every rank issues
    the same exchanges in the same order, and a slot's stream orders its reuse.

    ``make_renderer`` builds a HIP ``Renderer`` with the scene already set.  With ``composite="auto"`` a probe
    frame decides between the peer-to-peer exchange and the collectives (see the module docstring)."""

    def __init__(self, make_renderer, n_instances, height, rank, world, depth=2, composite="auto", verify=True,
                 split="interleaved", chunk=DEFAULT_CHUNK, band=DEFAULT_BAND):
        self.slots = []
        for _ in range(max(1, depth)):
            r = make_renderer()
            stream = torch.cuda.Stream()
            vis = torch.as_tensor(r.visbuffer_device_array(), device="cuda")
            color = torch.as_tensor(r.color_device_array(), device="cuda")
            self.slots.append((SortLastFrame(r, n_instances, height, rank, world, vis, color, composite, split, chunk, band), stream))
        self.issued = 0
        if verify and composite == "auto" and self.composite == "p2p" and not self._p2p_matches_collective(height, rank, world):
            # the direct exchange is not trusted on this machine: every slot goes back to the collectives
            scatter_ok, _ = probe_collectives(True, height if split == "contiguous" else world, world)
            for f, _ in self.slots:
                f.close()
                f.composite = "reduce_scatter" if scatter_ok else "all_reduce"

    def _p2p_matches_collective(self, height, rank, world):
        """Probe frames composited both ways on slot 0: the keys of this rank's rows and the gathered image must be
        identical on every rank.  Two peer-to-peer frames with different suns and cameras run first, and the second
        is the one compared: a machine on which a rank's caches kept lines of a peer's buffers from the frame before
        would pass a single frame and fail this.  The peer-to-peer frames contain no collective, so a rank on which
        they fail (an exception, a peer that never arrives) still meets the others in the agreement that follows."""
        f, stream = self.slots[0]
        mine = f.rows.to("cuda")  # the rows whose composited keys this rank holds
        keys = image = None
        with torch.cuda.stream(stream):
            try:
                f.step(sun=(0.3, 0.8, 0.52))  # another sun: other shadows, the same keys
                f.step(camera=self._probe_camera(f.r))  # another view: other keys everywhere
                f.step()
                torch.cuda.synchronize()
                ok = f.r.frame_status() == 0
                keys = f.vis.view(height, -1).index_select(0, mine)
                image = f.color.clone()
            except Exception:
                ok = False
            if not _all_agree(ok):
                return False
            f.composite = "all_reduce"
            f.step()
            torch.cuda.synchronize()
            same = bool(torch.equal(keys, f.vis.view(height, -1).index_select(0, mine))) and bool(torch.equal(image, f.color))
            f.composite = "p2p"
        return _all_agree(same)

    @staticmethod
    def _probe_camera(renderer):
        """The scene's camera moved sideways and turned a little (None when the renderer does not expose its scene:
        the probe then repeats the default view)."""
        cam = getattr(getattr(renderer, "scene", None), "camera", None)
        if cam is None:
            return None
        from .scene import Camera
        px, py, pz = cam.position
        dx, dy, dz = cam.direction
        return Camera((px + 2.5, py + 1.0, pz - 1.5), (dx + 0.15, dy + 0.05, dz))

    @property
    def composite(self):
        """The exchange step in use."""
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

    def time_composite(self):
        for f, _ in self.slots:
            f.time_composite()

    def composite_ms(self):
        parts = [f.composite_ms() for f, _ in self.slots]
        parts = [p for p in parts if p]
        if not parts:
            return None
        return sum(p[0] for p in parts) / len(parts), sum(p[1] for p in parts) / len(parts)

    def close(self):
        torch.cuda.synchronize()
        for f, _ in self.slots:
            f.close()
        for f, _ in self.slots:
            f.r.close()
