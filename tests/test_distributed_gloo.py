"""world_size-2 (and 3) `gloo` tests of the row-strip driver on the CPU.

forge3d_amd.distributed.StripRenderer is exercised exactly as bench.py / a multi-GPU job
uses it -- per-frame 4-row reservoir halo exchange (batch_isend_irecv), per-window
all-reduce(MAX) of the statistics record, final gather of the strips -- with the kernel
emulator standing in for the HIP session.  The stitched image must equal the single-strip
image bit for bit, because RNG and state are keyed by full-image coordinates.
"""
from __future__ import annotations

import os
import pickle
import socket
import sys
import tempfile
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_path, mode):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import scenes
    from emul import emul
    from forge3d_amd.distributed import StripRenderer, init_process_group

    init_process_group(world, rank, backend="gloo")
    dem = scenes.golden_dem(4)
    kw = scenes.scene_kwargs(dem)
    if mode == "fixed":
        kw = scenes.fixed_frames(kw, 6, spp=2)
    else:  # converge through two Welford windows
        kw = {**kw, "variance_threshold": 5e-3, "max_frames": 256, "spp": 1}
    backend, extra = emul.EmulBackend(), {}
    if mode in ("balanced", "refined", "probefail", "wholemap"):
        # synthetic costs: rows of the top half ("sky") cost 1, the others 5 -> unequal strips.  "balanced": the cut comes
        # from the row-cost map the ranks measure together, each its equal share of the rows (round 6, one all-gather);
        # "refined": three measured rounds on top of a FLAT map; "probefail": the last rank's probe raises -- its rows count as
        # the mean of the others', nobody aborts (round-5 advice); "wholemap": a backend that can only probe the whole frame
        class CostBackend(emul.EmulBackend):
            def row_costs(self, dem, width, height, cam, kw, frames=3, row_begin=0, row_end=None):
                import numpy as np

                row_end = height if row_end is None else row_end
                assert (row_begin, row_end) == (height * rank // world, height * (rank + 1) // world), "every rank probes its equal share"
                if mode == "probefail" and rank == world - 1:
                    raise RuntimeError("[Memory] injected: the probe session exceeds the memory budget")
                if mode == "refined":
                    return np.ones(row_end - row_begin)
                return np.asarray([1.0 if y < height // 2 else 5.0 for y in range(row_begin, row_end)])

            def probe(self, dem, width, height, cam, row_begin, row_end, kw, frames=2, whole_loop=False):
                return float(sum(1.0 if y < height // 2 else 5.0 for y in range(row_begin, row_end)))

        if mode == "wholemap":
            class CostBackend(emul.EmulBackend):  # noqa: F811
                def row_costs(self, dem, width, height, cam, kw, frames=3):
                    import numpy as np

                    return np.asarray([1.0 if y < height // 2 else 5.0 for y in range(height)])

        backend = CostBackend()
        if mode == "refined":
            extra["balance_iters"] = 4
        kw = scenes.fixed_frames(kw, 6, spp=2)
    if mode == "bounds":
        extra["row_bounds"] = [0, 7, 50] if world == 2 else [0, 4, 41, 50]
        kw = scenes.fixed_frames(kw, 6, spp=2)
    if mode == "noconv":  # the variance gate cannot be met: every rank must raise the reference's message
        kw = {**scenes.scene_kwargs(dem), "variance_threshold": 1e-12, "max_frames": 8, "min_frames": 2, "spp": 1}
    if mode in ("rankfail", "halotimeout"):  # one rank's session breaks in the middle of the render: nobody may be left in a collective
        kw = scenes.fixed_frames(kw, 40, spp=1)
    r = StripRenderer(dem, 72, 50, scenes.CAM, rank=rank, world=world, backend=backend, **extra, **kw)
    if mode in ("noconv", "rankfail", "halotimeout"):
        if mode == "halotimeout":
            # ONE rank's device-side halo wait timed out.  The count must ride in the record every rank
            # all-reduces -- a rank that raised before that all-reduce would pair its next, 1-word agreement all-reduce with
            # the others' statistics all-reduce -- and every rank must raise the same error after the same collective.
            r.peer_halos = True  # (the emulator exchanges halos over gloo; only the time-out bookkeeping of the peer path is played)
            r.session.halo_timeouts = (lambda: 1) if rank == world - 1 else (lambda: 0)
            r.session.enqueue_batch_strip = lambda first, count, collect=False: [
                (r.session.enqueue_frame_part(f, 1, collect and f + 1 == first + count), r.exchange_halos(f & 1)) for f in range(first, first + count)]
        if mode == "rankfail" and rank == world - 1:
            real = r.session.enqueue_frame_part

            def broken(frame, part, collect=False):
                if frame >= 33:
                    raise RuntimeError("[Device] Device error: injected failure on the last rank")
                return real(frame, part, collect)

            r.session.enqueue_frame_part = broken
        message = ""
        try:
            r.render()
        except RuntimeError as exc:
            message = str(exc)
        r.close()
        with open(f"{out_path}.{rank}", "w") as f:
            f.write(message)
        if rank == 0:
            with open(out_path, "wb") as f:
                pickle.dump({"messages": world}, f)
        dist.barrier()
        dist.destroy_process_group()
        return
    if mode == "probefail":
        assert r.cost_probe_failed_ranks == [world - 1] and (r.cost_probe_failure is not None) == (rank == world - 1)
        assert r.balance_log[0]["failed_ranks"] == [world - 1]
        sizes = [b1 - b0 for b0, b1 in zip(r.bounds, r.bounds[1:])]
        assert sum(sizes) == 50 and min(sizes) >= 4 and sizes[0] > sizes[-1], r.bounds  # the known rows still shape the cut
    if mode in ("balanced", "refined", "wholemap"):
        sizes = [b1 - b0 for b0, b1 in zip(r.bounds, r.bounds[1:])]
        costs = [sum(1.0 if y < 25 else 5.0 for y in range(b0, b1)) for b0, b1 in zip(r.bounds, r.bounds[1:])]
        assert sizes[0] > sizes[-1] and max(costs) / (sum(costs) / world) < 1.15, (r.bounds, costs)  # (ROW_COST_FLOOR weighs every row a little: 50 rows cut three ways cannot do better)
        assert len(r.balance_log) == 1 if mode in ("balanced", "wholemap") else len(r.balance_log) >= 3, r.balance_log
    image = r.render() if mode == "converge" else None
    if mode in ("fixed", "balanced", "refined", "bounds", "probefail", "wholemap"):
        r.run_frames(0, 6, collect_last=True)
        var = r.window_variance(6)
        image = r.gather_image(6)
        if image is not None:
            image["variance"] = var
    t = r.max_over_ranks(float(rank))
    assert t == float(world - 1)
    r.close()
    if rank == 0:
        with open(out_path, "wb") as f:
            pickle.dump(image, f)
    else:
        assert image is None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run(world, mode):
    import torch.multiprocessing as mp

    out = tempfile.mktemp(suffix=".pkl")
    port = _free_port()
    if world == 1:
        _worker(0, 1, port, out, mode)
    else:
        mp.spawn(_worker, args=(world, port, out, mode), nprocs=world, join=True)
    with open(out, "rb") as f:
        image = pickle.load(f)
    os.unlink(out)
    return image


@pytest.mark.parametrize("world", [2, 3, 4])  # 4: interior ranks with a neighbour on both sides, the driver's many-rank defaults
def test_strips_over_gloo_reproduce_the_single_strip_image(world):
    single = _run(1, "fixed")
    multi = _run(world, "fixed")
    for key in ("rgba", "albedo", "normal", "depth"):
        assert np.array_equal(single[key], multi[key], equal_nan=True), key
    assert single["variance"] == multi["variance"]


@pytest.mark.parametrize("world,mode", [(2, "balanced"), (3, "balanced"), (3, "refined"), (2, "bounds"), (3, "bounds"), (3, "probefail"), (2, "wholemap")])
def test_unequal_strips_reproduce_the_single_strip_image(world, mode):
    """Load-balanced (measured, here with a synthetic cost probe) and hand-picked boundaries:
    every rank derives the same partition and the stitched image is still bit-identical."""
    single = _run(1, "fixed")
    multi = _run(world, mode)
    for key in ("rgba", "albedo", "normal", "depth"):
        assert np.array_equal(single[key], multi[key], equal_nan=True), key
    assert single["variance"] == multi["variance"]


def test_partition_rows_balances_density_and_respects_the_halo_minimum():
    from forge3d_amd.distributed import partition_rows, rebalance

    h = 1080
    density = np.where(np.arange(h) < 400, 0.2, 1.0)
    for world in (2, 4, 8):
        b = partition_rows(density, world)
        assert b[0] == 0 and b[-1] == h and all(b1 - b0 >= 4 for b0, b1 in zip(b, b[1:]))
        cost = [density[b0:b1].sum() for b0, b1 in zip(b, b[1:])]
        assert max(cost) / (sum(cost) / world) < 1.02
    # degenerate densities fall back to equal strips; tiny images keep >= HALO_ROWS rows per strip
    assert partition_rows(np.zeros(16), 4) == [0, 4, 8, 12, 16]
    assert partition_rows([1e9] + [0] * 11 + [1e9], 3) == [0, 4, 9, 13]
    with pytest.raises(ValueError):
        partition_rows(np.ones(11), 3)
    # the multiplicative update converges on a step-shaped cost within a few rounds
    true = np.where(np.arange(h) < 333, 0.15, 1.0)
    est, b = np.ones(h), partition_rows(np.ones(h), 8)
    for _ in range(4):
        est = rebalance(est, b, [true[b0:b1].sum() for b0, b1 in zip(b, b[1:])])
        b = partition_rows(est, 8)
    cost = [true[b0:b1].sum() for b0, b1 in zip(b, b[1:])]
    assert max(cost) / (sum(cost) / 8) < 1.05


def test_distributed_convergence_gate_matches_single_process():
    import scenes
    from oracle import oracle

    multi = _run(2, "converge")
    dem = scenes.golden_dem(4)
    want = oracle.render(dem, 72, 50, scenes.CAM, **{**scenes.scene_kwargs(dem), "variance_threshold": 5e-3,
                                                     "max_frames": 256, "spp": 1})
    assert multi["frames"] == want["frames"] and multi["converged"]
    assert np.float32(multi["variance"]) == np.float32(want["variance"])
    for key in ("rgba", "albedo", "normal", "depth"):
        assert np.array_equal(multi[key], want[key], equal_nan=True), key


def _messages(world, mode):
    import glob

    _run(world, mode)
    out = []
    for path in sorted(glob.glob(os.path.join(tempfile.gettempdir(), "*.pkl.[0-9]"))):
        out.append(open(path).read())
        os.unlink(path)
    return out


def test_every_rank_raises_the_reference_non_convergence_message():
    msgs = _messages(2, "noconv")
    assert len(msgs) == 2
    for m in msgs:
        assert "did not converge: per-pixel luminance variance" in m and "over the last 32-frame window after 8 frames" in m
        assert "(threshold 1.0e-12); raise max_frames or simplify the scene" in m and "refusing to return a fake reference" in m


def test_a_failing_rank_stops_every_rank_instead_of_hanging_them():
    msgs = _messages(2, "rankfail")
    assert len(msgs) == 2
    assert any("injected failure on the last rank" in m for m in msgs)
    assert any("another rank of the strip job failed" in m for m in msgs)


@pytest.mark.parametrize("world", [2, 3])
def test_a_halo_time_out_on_one_rank_is_raised_by_every_rank_after_the_same_collective(world):
    msgs = _messages(world, "halotimeout")
    assert len(msgs) == world
    for m in msgs:
        assert "halo wait timed out" in m, msgs


def test_strip_rows_partition_the_image():
    from forge3d_amd.distributed import strip_rows

    for h, n in ((1080, 8), (50, 3), (7, 2), (4096, 8)):
        bounds = [strip_rows(h, n, r) for r in range(n)]
        assert bounds[0][0] == 0 and bounds[-1][1] == h
        assert all(bounds[i][1] == bounds[i + 1][0] for i in range(n - 1))
        sizes = [e - b for b, e in bounds]
        assert max(sizes) - min(sizes) <= 1
