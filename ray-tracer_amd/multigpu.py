"""Multi-GPU frame: the image is cut into row bands, one process and one GPU per band, the scene replicated.

The reference is single-GPU; this is the sharding the path admits.  Every kernel of the frame reads only its own
pixel, except SpatialReusePass, which gathers reservoirs and depths from up to MaxOffset rows above and below
(plus one more row above, because a negative x offset wraps into the previous row:
BVH_ReSTIR_DL/.../ReSTIR_BVH_RayTracer.compute:526-535).  So each process

  * keeps FULL-FRAME per-pixel buffers but renders only rows [b[r], b[r+1]) (rt_set_row_band): pixel coordinates,
    RNG seeds and buffer indices stay global, which makes the banded result bit-identical to the single-GPU one;
  * after TemporalReuse exchanges, ONCE per frame, its border rows of DepthBuffer and of the first pass's source
    reservoir buffer with its two neighbours (nearest-neighbour send/recv: over RCCL each transfer rides one direct
    xGMI link), deep enough that it can run the earlier spatial passes on a slightly widened band instead of
    exchanging again before every pass (halo_plan);
  * sends its band of Result (the displayed image) to rank 0 at the end of the frame.

The schedule (which rows, which buffer, which peer) is in plain functions over a `RowStore`, so the same code runs
over GPU memory with RCCL and, in the CPU tests, over numpy arrays with gloo.
"""
import torch
import torch.distributed as dist


def band_bounds(height, world, weights=None, min_rows=8):
    """Row boundaries b[0..world] of the bands, each a multiple of 8 (the tile height) except the last.

    weights: optional per-row cost estimate; bands then get roughly equal cost instead of equal height, but never
    fewer than min_rows rows (a band must be at least as tall as the halo its neighbours take from it)."""
    if weights is None:
        cuts = [round(height * r / world / 8) * 8 for r in range(world)] + [height]
    else:
        import numpy as np
        min_rows = -(-max(min_rows, 8) // 8) * 8
        c = np.concatenate([[0.0], np.cumsum(np.asarray(weights, dtype=np.float64))])
        cuts = [0]
        for r in range(1, world):
            row = int(np.searchsorted(c, c[-1] * r / world))
            cut = max(cuts[-1] + min_rows, round(row / 8) * 8)
            cut = min(cut, (height - (world - r) * min_rows) // 8 * 8)      # leave room for the bands below
            cuts.append(cut)
        cuts.append(height)
    for a, b in zip(cuts, cuts[1:]):
        if b <= a:
            raise ValueError(f"image of {height} rows cannot be cut into {world} non-empty bands of whole tiles")
    return cuts


def row_cost(depth, hit_cost=24.0):
    """Per-row cost estimate from a rendered depth buffer: every pixel costs one unit (the streaming work), a pixel
    whose primary ray hit geometry `hit_cost` more (its shadow and light rays and the reuse passes).  The ratio is
    the one measured on the 80kScene stand-in at 1080p (0.09 ns per pixel, 2.2 ns per hit pixel)."""
    import numpy as np
    hits = np.isfinite(np.asarray(depth)).sum(axis=1).astype(np.float64)
    return depth.shape[1] + hit_cost * hits


def spatial_halo(max_offset):
    """Rows a spatial pass reads beyond a band edge: offsets reach MaxOffset rows down and MaxOffset rows up, and a
    negative x offset borrows from one row further up."""
    return max_offset + 1


def max_offset_of_pass(i, blur):
    return int(3 ** i) * blur          # Main.cs:394


class RowStore:
    """Access to image rows of named per-pixel buffers as lists of 1-D uint8 torch tensors (memory views).

    segments(..., recv=False) returns tensors to send from, recv=True tensors to receive into; commit() is called
    once the transfers have completed (a store that stages through host memory copies the received rows back)."""

    def segments(self, name, row0, row1, recv=False):
        raise NotImplementedError

    def commit(self):
        pass


def _exchange(store, names, rank, world, bounds, halo, height):
    """Sends this rank's border rows of `names` to its neighbours and receives theirs, in place."""
    ops, keep = [], []
    for peer, mine, theirs in ((rank - 1, (bounds[rank], min(bounds[rank] + halo, bounds[rank + 1])),
                                (max(bounds[rank] - halo, 0), bounds[rank])),
                               (rank + 1, (max(bounds[rank + 1] - halo, bounds[rank]), bounds[rank + 1]),
                                (bounds[rank + 1], min(bounds[rank + 1] + halo, height)))):
        if peer < 0 or peer >= world:
            continue
        # a halo taller than the neighbour's band would need rows of a rank further away
        if theirs[1] - theirs[0] > bounds[peer + 1] - bounds[peer]:
            raise ValueError(f"halo of {halo} rows exceeds the band of rank {peer}; use fewer GPUs or a larger image")
        for name in names:
            for t in store.segments(name, *mine):
                ops.append(dist.P2POp(dist.isend, t, peer))
            for t in store.segments(name, *theirs, recv=True):
                ops.append(dist.P2POp(dist.irecv, t, peer))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        store.commit()
    return len(ops)


def _gather(store, names, rank, world, bounds, dst=0):
    """Every rank sends its band of `names` to rank `dst`, which receives them into its own full-frame buffers."""
    ops = []
    if rank == dst:
        for r in range(world):
            if r != dst:
                for name in names:
                    for t in store.segments(name, bounds[r], bounds[r + 1], recv=True):
                        ops.append(dist.P2POp(dist.irecv, t, r))
    else:
        for name in names:
            for t in store.segments(name, bounds[rank], bounds[rank + 1]):
                ops.append(dist.P2POp(dist.isend, t, dst))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        store.commit()


def halo_plan(iterations, blur):
    """Per spatial pass i: (rows it reads beyond the rows it writes, rows it must write beyond the band).

    All halos are exchanged ONCE, before the first pass: a rank receives enough rows of the pass-0 source that it can
    run the earlier passes on a band widened by what the later passes will read, recomputing a few of its
    neighbours' rows instead of waiting for them (one exchange per frame instead of one per pass).  Pass i is run on
    band +- extend[i], with extend[i] = sum of the halos of the passes after it."""
    halos = [spatial_halo(max_offset_of_pass(i, blur)) for i in range(iterations)]
    extend = [sum(halos[i + 1:]) for i in range(iterations)]
    return halos, extend, sum(halos)


def interior_rows(rank, world, bounds, iterations, blur):
    """Rows of the first spatial pass that read nothing a neighbour owns: the band shrunk by that pass's halo on every
    side that has a neighbour (the twin of mgInterior in host/frame_driver.cpp)."""
    b0, b1 = bounds[rank], bounds[rank + 1]
    halo0 = spatial_halo(max_offset_of_pass(0, blur)) if iterations > 0 else 0
    lo = b0 + halo0 if rank > 0 else b0
    hi = b1 - halo0 if rank < world - 1 else b1
    return lo, max(hi, lo)


def render_frame_banded(stages, store, rank, world, bounds, height, iterations, blur):
    """One frame of the ReSTIR sequence on this rank's band, one halo exchange and one gather.

    `stages` provides the compute: update_per_frame(), set_rows(r0, r1), initial_trace(), temporal_reuse(),
    spatial_pass(i), transfer(), trace_rays(), accumulate(); they act on the rows last given to set_rows.
    The order is the native loop's (host/frame_driver.cpp): the first spatial pass runs on the interior rows BEFORE the
    halos are exchanged (natively: while they travel), on the border rows after."""
    b0, b1 = bounds[rank], bounds[rank + 1]
    stages.set_rows(b0, b1)
    stages.update_per_frame()
    stages.initial_trace()
    stages.temporal_reuse()
    if iterations > 0:
        _, extend, total = halo_plan(iterations, blur)
        lo, hi = interior_rows(rank, world, bounds, iterations, blur)
        if hi > lo:
            stages.set_rows(lo, hi)
            stages.spatial_pass(0)
        _exchange(store, ["DepthBuffer", "Candidates"], rank, world, bounds, total, height)
        for i in range(iterations):
            # never widen past the image, nor past the rows whose inputs were received
            wlo = max(b0 - extend[i] if rank > 0 else b0, 0)
            whi = min(b1 + extend[i] if rank < world - 1 else b1, height)
            if i == 0:
                if hi <= lo:
                    strips = [(wlo, whi)]
                else:
                    strips = [(wlo, min(lo, whi)), (max(hi, wlo), whi)]
                for s0, s1 in strips:
                    if s1 > s0:
                        stages.set_rows(s0, s1)
                        stages.spatial_pass(0)
            else:
                stages.set_rows(wlo, whi)
                stages.spatial_pass(i)
        stages.set_rows(b0, b1)
    stages.transfer()
    stages.trace_rays()
    stages.accumulate()
    _gather(store, ["Result"], rank, world, bounds)   # the displayed image; the running sum stays per band


def render_ris_frame_banded(stages, store, rank, world, bounds):
    """One frame of the RIS project (BVH_RIS_DL/Assets/Scripts/Main.cs:132-144, 319-376) on this rank's band.  Its
    kernels read only their own pixel, so there is nothing to exchange; rank 0 collects the displayed image.

    `stages` provides set_rows(r0, r1), update_per_frame(), generate_candidates(), trace_rays(), accumulate()."""
    stages.set_rows(bounds[rank], bounds[rank + 1])
    stages.update_per_frame()
    stages.generate_candidates()
    stages.trace_rays()
    stages.accumulate()
    _gather(store, ["Result"], rank, world, bounds)


# ------------------------------------------------------------------------------------------------- GPU side
class _DevicePtr:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}


class GpuRowStore(RowStore):
    """Rows of the driver's buffers and textures as torch tensors aliasing the library's device memory."""

    BUFFERS = ("Candidates", "CandidatesB", "TemporalFrameBuffer", "HitInfos")

    def __init__(self, ctx, drv, device, stage_through_host=False):
        # stage_through_host: for backends that cannot move device memory (gloo, used to rehearse the multi-process
        # path on a single GPU); rows then travel device -> host -> peer -> host -> device.
        self.ctx, self.drv, self.device, self.stage = ctx, drv, device, stage_through_host
        self._pending = []
        self._views = {}

    def segments(self, name, row0, row1, recv=False):
        if row1 <= row0:
            return []
        if name in self.BUFFERS:
            segs = self.ctx.buffer_rows(self.drv.buffer_handle(name), self.drv.W, row0, row1)
        else:
            segs = self.ctx.texture_rows(self.drv.texture_handle(name), row0, row1)
        dev = []
        for p, n in segs:      # the tensors only alias memory: build each once (pointers move only on re-upload)
            t = self._views.get((p, n))
            if t is None:
                t = self._views[(p, n)] = torch.as_tensor(_DevicePtr(p, n), device=self.device)
            dev.append(t)
        if not self.stage:
            return dev
        if recv:
            host = [torch.empty(t.shape, dtype=torch.uint8) for t in dev]
            self._pending += list(zip(host, dev))
            return host
        return [t.cpu() for t in dev]      # synchronising copy on the current (= the library's) stream

    def commit(self):
        for host, dev in self._pending:
            dev.copy_(host)
        self._pending = []


class _DriverStages:
    def __init__(self, ctx, drv):
        self.drv = drv
        self.set_rows = ctx.set_row_band
        self.update_per_frame = drv.update_per_frame
        self.initial_trace = lambda: drv.run_stage(0)
        self.temporal_reuse = lambda: drv.run_stage(1)
        self.spatial_pass = drv.run_spatial_pass
        self.transfer = drv.run_transfer
        self.trace_rays = lambda: drv.run_stage(3)
        self.accumulate = drv.run_post_processing


class DistributedRunner:
    """bench.py's runner for N > 1: one process per GPU, RCCL ("nccl" backend) between them."""

    def __init__(self, rt, scene, cfg, rank, world, local_rank, backend="nccl", weights=None):
        self.rank, self.world, self.cfg = rank, world, cfg
        torch.cuda.set_device(local_rank)
        self.device = torch.device("cuda", local_rank)
        if not dist.is_initialized():
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=self.device if backend == "nccl" else None)
        self.H = cfg.Resolution[1]
        self.bounds = band_bounds(self.H, world, weights)
        self.ctx = rt.RtContext(local_rank)
        self.drv = rt.Driver(self.ctx, cfg, scene, self.bounds[rank], self.bounds[rank + 1])
        self.store = GpuRowStore(self.ctx, self.drv, self.device, stage_through_host=(backend != "nccl"))
        self.stages = _DriverStages(self.ctx, self.drv)
        # torch orders its collectives against the CURRENT torch stream: make that the library's stream, so sends
        # wait for the kernels that produced the rows and later kernels wait for the receives, all on the GPU.
        self.stream = torch.cuda.ExternalStream(self.ctx.stream(), device=self.device)

    def render_frame(self):
        with torch.cuda.stream(self.stream):
            render_frame_banded(self.stages, self.store, self.rank, self.world, self.bounds, self.H,
                                self.cfg.SpatialReuseIterations, self.cfg.SpatialReuseBlur)

    def sync(self):
        self.ctx.sync()
        torch.cuda.synchronize(self.device)

    def barrier(self):
        dist.barrier()

    def reduce(self, dt, rays):
        t = torch.tensor([dt], dtype=torch.float64, device=self.device)
        r = torch.tensor([rays], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        return float(t.item()), float(r.item())

    def band_pixels(self):
        return (self.bounds[self.rank + 1] - self.bounds[self.rank]) * self.cfg.Resolution[0]

    def close(self):
        self.drv.close()
        self.ctx.close()
        if dist.is_initialized():
            dist.destroy_process_group()


# ---------------------------------------------------------------------------------- native (C++ + RCCL) frame loop
def native_band_bounds(height, world):
    """rth_mg_band_bounds: the C++ twin of band_bounds (equal-height bands)."""
    import ctypes as C
    import importlib
    host = importlib.import_module("ray-tracer_amd").host_lib()
    out = (C.c_int * (world + 1))()
    if host.rth_mg_band_bounds(height, world, out) != 0:
        raise ValueError(f"image of {height} rows cannot be cut into {world} non-empty bands of whole tiles")
    return list(out)


def native_halo_plan(iterations, blur):
    """rth_mg_halo_plan: the C++ twin of halo_plan."""
    import ctypes as C
    import importlib
    host = importlib.import_module("ray-tracer_amd").host_lib()
    n = max(iterations, 1)
    halo, extend = (C.c_int * n)(), (C.c_int * n)()
    total = host.rth_mg_halo_plan(iterations, blur, halo, extend)
    return list(halo)[:iterations], list(extend)[:iterations], total


def native_balanced_bounds(height, world, weights, min_rows):
    """rth_mg_balanced_bounds: the C++ twin of band_bounds(height, world, weights, min_rows)."""
    import ctypes as C
    import importlib
    host = importlib.import_module("ray-tracer_amd").host_lib()
    w = (C.c_double * height)(*[float(x) for x in weights])
    out = (C.c_int * (world + 1))()
    if host.rth_mg_balanced_bounds(height, world, w, min_rows, out) != 0:
        raise ValueError(f"image of {height} rows cannot be cut into {world} non-empty bands of whole tiles")
    return list(out)


class RcclRunner:
    """One rank of the native multi-GPU frame loop (host/frame_driver.cpp: rth_mg_*): the banded schedule above, run
    from C++ with RCCL send/recv on the library's stream, so no Python runs between the dispatches of a frame.
    torch.distributed (gloo) is only the side channel that carries the RCCL unique id and the timing reductions."""

    def __init__(self, rt, scene, cfg, rank, world, local_rank, force_comm=False, balance=True):
        import ctypes as C
        self.rt, self.rank, self.world = rt, rank, world
        self.host = rt.host_lib()
        self.ctx = rt.RtContext(local_rank)
        # an RisConfig selects the RIS project's frame (no halo: its kernels read only their own pixel)
        self.ris = isinstance(cfg, rt.RisConfig)
        self.drv = rt.RisDriver(self.ctx, cfg, scene) if self.ris else rt.Driver(self.ctx, cfg, scene)
        self.W, self.H = cfg.Resolution[0], cfg.Resolution[1]
        # Every rank takes part in the id broadcast whatever happened on rank 0 (a rank that raised before it would
        # leave the others waiting in it): rank 0 sends the id, or the reason it has none, and all ranks raise together.
        uid, why = None, ""
        if world > 1 or force_comm:
            buf = C.create_string_buffer(128)
            if rank == 0:
                if self.host.rth_mg_unique_id(buf) == 0:
                    uid = buf.raw
                else:
                    why = "ncclGetUniqueId failed (is librccl.so loadable?)"
            if world > 1:
                if not dist.is_initialized():
                    dist.init_process_group("gloo", rank=rank, world_size=world)
                box = [(uid, why)]
                dist.broadcast_object_list(box, src=0)
                uid, why = box[0]
            if uid is None:
                self.drv.close()
                self.ctx.close()
                raise RuntimeError(why or "no RCCL unique id")
        err = C.create_string_buffer(512)
        create = self.host.rth_mg_create_ris if self.ris else self.host.rth_mg_create
        self.h = create(self.drv.h, rank, world, None, uid, err, 512)
        if not self.h:
            raise RuntimeError("rth_mg_create: " + err.value.decode())
        if balance and world > 1 and not self.ris:
            # cost-balanced bands: each rank traces the primary rays of its equal-height band once, the per-row costs are
            # summed over the ranks with one small all-reduce, and all ranks derive the same cuts (rth_mg_balance)
            if self.host.rth_mg_balance(self.h) != 0:
                raise RuntimeError("rth_mg_balance: " + self.host.rth_mg_last_error(self.h).decode())
        b = (C.c_int * (world + 1))()
        self.host.rth_mg_bounds(self.h, b)
        self.bounds = list(b)

    def render_frames(self, n):
        if self.host.rth_mg_render_frames(self.h, n) != 0:
            raise RuntimeError("rth_mg_render_frames: " + self.host.rth_mg_last_error(self.h).decode())

    def render_frame(self):
        self.render_frames(1)

    def sync(self):
        self.ctx.sync()

    def barrier(self):
        if self.world > 1:
            dist.barrier()

    def reduce(self, seconds, rays):
        """(max over ranks of seconds, sum over ranks of rays)"""
        if self.world == 1:
            return seconds, rays
        t = torch.tensor([seconds], dtype=torch.float64)
        r = torch.tensor([rays], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        return float(t[0]), int(r[0])

    def band_pixels(self):
        return (self.bounds[self.rank + 1] - self.bounds[self.rank]) * self.W

    TIME_NAMES = ("frame", "before_halo", "interior_rows", "after_halo", "halo_exchange", "halo_exposed", "gather")

    def timed_frames(self, n):
        """Renders n frames with the per-frame stamps on (each frame is waited for) and returns the average milliseconds
        of rth_mg_times: dispatch-stream time of the frame and of its parts, transfer-stream time of the halo exchange
        and of the gather, and the part of the halo exchange the dispatch stream waited for."""
        import ctypes as C
        self.host.rth_mg_enable_timing(self.h, 1)
        try:
            self.render_frames(n)
            self.sync()
            out = (C.c_double * len(self.TIME_NAMES))()
            self.host.rth_mg_times(self.h, out)
        finally:
            self.host.rth_mg_enable_timing(self.h, 0)
        return {k: float(v) for k, v in zip(self.TIME_NAMES, out)}

    def close(self, keep_group=False):
        """keep_group: the torch.distributed side channel stays up (another runner of this process still uses it)."""
        if self.h:
            self.host.rth_mg_destroy(self.h)
            self.h = None
        self.drv.close()
        self.ctx.close()
        if self.world > 1 and dist.is_initialized() and not keep_group:
            dist.destroy_process_group()
