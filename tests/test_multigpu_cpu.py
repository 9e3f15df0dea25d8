"""The multi-GPU schedule (row bands, one halo exchange per frame, widened early spatial passes, frame-buffer
gather) on the CPU:
world_size 2 and 3 over gloo, with the CPU oracle doing each rank's band.  The banded result must equal the
single-process full-frame result bit for bit, which is the property the GPU path relies on."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NumpyRowStore:
    def __init__(self, orc):
        self.orc = orc

    def segments(self, name, row0, row1, recv=False):
        if row1 <= row0:
            return []
        W = self.orc.W
        if name in self.orc.buf:
            flat = torch.from_numpy(self.orc.buf[name].view(np.uint8))
            return [flat[row0 * W * 52: row1 * W * 52]]
        a = self.orc.tex[name]
        flat = torch.from_numpy(a.reshape(-1).view(np.uint8))
        texel = a.itemsize * (a.shape[2] if a.ndim == 3 else 1)
        return [flat[row0 * W * texel: row1 * W * texel]]


    def commit(self):
        pass


class OracleStages:
    def __init__(self, orc, iterations):
        self.orc, self.iterations = orc, iterations
        self.update_per_frame = orc.update_per_frame
        self.set_rows = orc.set_band
        self.initial_trace = orc.initial_trace
        self.temporal_reuse = orc.temporal_reuse
        self.trace_rays = orc.trace_rays
        self.accumulate = orc.accumulate

    def spatial_pass(self, i):
        self.orc.spatial_pass(i, i % 2 == 0)

    def transfer(self):
        if self.iterations % 2 == 1:
            self.orc.transfer_to_original()


def _worker(rank, world, port, res, frames, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import harness as h
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        scene = h.rt.build_scene_s0(subdivide=False, threads=1)
        cfg = h.small_config(*res)
        bounds = mg.band_bounds(res[1], world)
        banded = h.OracleRenderer(scene.arrays(), cfg, threads=1, records=False, row0=bounds[rank], row1=bounds[rank + 1])
        full = h.OracleRenderer(scene.arrays(), cfg, threads=1, records=False)
        store, stages = NumpyRowStore(banded), OracleStages(banded, cfg.SpatialReuseIterations)
        W = res[0]
        for frame in range(frames):
            mg.render_frame_banded(stages, store, rank, world, bounds, res[1], cfg.SpatialReuseIterations, cfg.SpatialReuseBlur)
            full.render_frame()
            lo, hi = bounds[rank] * W, bounds[rank + 1] * W
            for name in ("HitInfos", "Candidates", "CandidatesB", "TemporalFrameBuffer"):
                assert np.array_equal(h.bits(banded.buf[name][lo:hi]), h.bits(full.buf[name][lo:hi])), (rank, frame, name)
            r0, r1 = bounds[rank], bounds[rank + 1]
            assert np.array_equal(h.bits(banded.tex["AccumulatedResult"][r0:r1]), h.bits(full.tex["AccumulatedResult"][r0:r1]))
            if rank == 0:   # the gathered image is complete on rank 0
                assert np.array_equal(h.bits(banded.tex["Result"]), h.bits(full.tex["Result"])), frame
        hit = np.isfinite(full.tex["DepthBuffer"])
        assert hit.any() and not hit.all()      # the image has both geometry and sky
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def _ris_worker(rank, world, port, res, frames, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import harness as h
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        scene = h.rt.build_scene_s0(subdivide=False, threads=1)
        cfg = h.small_ris_config(*res)
        bounds = mg.band_bounds(res[1], world)
        banded = h.RisOracleRenderer(scene.arrays(), cfg, threads=1, records=False)
        full = h.RisOracleRenderer(scene.arrays(), cfg, threads=1, records=False)
        banded.set_rows = banded.set_band
        store = NumpyRowStore(banded)
        W = res[0]
        tot = cfg.SceneObjectCandidatesNum * cfg.TriCandidatesNum
        for frame in range(frames):
            mg.render_ris_frame_banded(banded, store, rank, world, bounds)
            full.render_frame()
            r0, r1 = bounds[rank], bounds[rank + 1]
            assert np.array_equal(h.bits(banded.buf["HitInfos"][r0 * W:r1 * W]), h.bits(full.buf["HitInfos"][r0 * W:r1 * W])), (rank, frame)
            assert np.array_equal(h.bits(banded.buf["Candidates"][r0 * W * tot:r1 * W * tot]),
                                  h.bits(full.buf["Candidates"][r0 * W * tot:r1 * W * tot])), (rank, frame)
            assert np.array_equal(h.bits(banded.tex["AccumulatedResult"][r0:r1]), h.bits(full.tex["AccumulatedResult"][r0:r1]))
            if rank == 0:
                assert np.array_equal(h.bits(banded.tex["Result"]), h.bits(full.tex["Result"])), frame
        assert full.tex["Result"][..., :3].any()
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,res", [(2, (64, 96)), (3, (48, 144))])
def test_banded_equals_full_frame(tmp_path, world, res):
    mp.spawn(_worker, args=(world, _free_port(), res, 3, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_ris_banded_equals_full_frame(tmp_path):
    """The RIS project's frame on two ranks: bands and the gather only (its kernels read no other pixel)."""
    mp.spawn(_ris_worker, args=(2, _free_port(), (64, 48), 2, str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(2))


def test_band_bounds_and_halo():
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    b = mg.band_bounds(1080, 8)
    assert b[0] == 0 and b[-1] == 1080 and all(x % 8 == 0 for x in b[:-1]) and all(y > x for x, y in zip(b, b[1:]))
    assert mg.band_bounds(2160, 8)[1] == 272
    with pytest.raises(ValueError):
        mg.band_bounds(16, 4)
    assert [mg.max_offset_of_pass(i, 3) for i in range(3)] == [3, 9, 27]      # Main.cs:394
    assert mg.spatial_halo(27) == 28
    assert mg.halo_plan(3, 3) == ([4, 10, 28], [38, 28, 0], 42)
    w = np.zeros(1080)
    w[400:800] = 1.0                      # all the cost in the middle rows
    bw = mg.band_bounds(1080, 4, weights=w)
    assert bw[1] >= 400 and bw[3] <= 800  # cost-balanced cuts fall inside the expensive region
    # a minimum height (the halo) is respected at both ends even when all the cost sits in a few rows
    w = np.zeros(1080)
    w[1000:1010] = 1.0
    bw = mg.band_bounds(1080, 8, weights=w, min_rows=42)
    assert all(b - a >= 48 for a, b in zip(bw, bw[1:-1])) and bw[-1] - bw[-2] >= 42 and all(x % 8 == 0 for x in bw[:-1])
    depth = np.full((16, 10), np.inf, np.float32)
    depth[4:8, :5] = 1.0
    assert list(mg.row_cost(depth, 2.0)) == [10.0] * 4 + [20.0] * 4 + [10.0] * 8


def test_native_plan_matches_python():
    """The C++ frame loop (rth_mg_*) computes its bands and halos with its own code: it must agree with the Python
    schedule that the gloo tests above validate."""
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    for height in (64, 96, 270, 1080, 1088, 2160, 4320):
        for world in (1, 2, 3, 4, 5, 7, 8):
            try:
                want = mg.band_bounds(height, world)
            except ValueError:
                with pytest.raises(ValueError):
                    mg.native_band_bounds(height, world)
                continue
            assert mg.native_band_bounds(height, world) == want, (height, world)
    for iterations in range(0, 5):
        for blur in (1, 2, 3, 5):
            halos, extend, total = mg.halo_plan(iterations, blur)
            assert mg.native_halo_plan(iterations, blur) == (halos, extend, total), (iterations, blur)
    # cost-balanced cuts: the C++ routine every rank runs on the summed per-row costs against the Python one
    rng = np.random.default_rng(5)
    for height, world, min_rows in ((1080, 8, 42), (2160, 8, 42), (1080, 3, 8), (400, 4, 14), (2160, 5, 30)):
        for trial in range(6):
            w = rng.random(height) ** 4 * (1 + 24 * 1920)
            w[: rng.integers(0, height // 2)] = 1920.0                   # a run of sky rows
            assert mg.native_balanced_bounds(height, world, w, min_rows) == mg.band_bounds(height, world, w, min_rows), (height, world, trial)
    with pytest.raises(ValueError):
        mg.native_balanced_bounds(64, 4, np.ones(64), 42)


def test_interior_rows():
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    b = [0, 136, 272, 408]
    assert mg.interior_rows(0, 3, b, 3, 3) == (0, 132)      # no neighbour above: the band's first rows need no halo
    assert mg.interior_rows(1, 3, b, 3, 3) == (140, 268)
    assert mg.interior_rows(2, 3, b, 3, 3) == (276, 408)
    assert mg.interior_rows(1, 3, [0, 8, 12, 400], 3, 3) == (12, 12)   # a band shorter than two halos has no interior
    assert mg.interior_rows(0, 1, [0, 100], 3, 3) == (0, 100)


class _FakeHost:
    """Stands in for librthost.so in the test below: no RCCL on rank 0."""
    def rth_mg_unique_id(self, buf):
        return -1


def _uid_failure_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    mg = importlib.import_module("ray-tracer_amd.multigpu")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        class Rt:   # just enough of the package for RcclRunner.__init__ to reach the id exchange
            RisConfig = type("RisConfig", (), {})

            @staticmethod
            def host_lib():
                return _FakeHost()

            class RtContext:
                def __init__(self, dev):
                    pass

                def close(self):
                    pass

            class Driver:
                def __init__(self, ctx, cfg, scene):
                    self.h = None

                def close(self):
                    pass

        cfg = type("Cfg", (), {"Resolution": [64, 64]})()
        try:
            mg.RcclRunner(Rt, None, cfg, rank, world, 0)
            raised = ""
        except RuntimeError as e:
            raised = str(e)
        assert "ncclGetUniqueId failed" in raised, raised   # the SAME error on every rank, and nobody is left waiting
        ok = torch.tensor([1])
        dist.all_reduce(ok)                                 # the collective sequence is still in step
        assert int(ok[0]) == world
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_unique_id_failure_reaches_every_rank(tmp_path):
    """If rank 0 cannot make the RCCL id (librccl missing), every rank must learn it from the broadcast and raise; a
    rank 0 that raised before the broadcast would leave the others blocked in it."""
    mp.spawn(_uid_failure_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(2))
