"""Two OS processes sharing the one GPU of the test box, each driving its own strip through the product
backend (HipBackend: libf3dhip sessions on torch-owned device buffers), joined by a `gloo` process
group.  RCCL refuses two ranks on one device, so the bytes are staged through the host here
(StripRenderer._comm_device) -- everything else is the multi-GPU path of bench.py: measured load
balancing with real probe frames, per-frame 4-row reservoir halo exchange, all-reduced variance gate,
gathered strips.  The stitched image must equal the single-process image bit for bit."""
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


def _worker(rank, world, port, out_path, in_flight=0, peer_halos=True, height=200):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import scenes
    from forge3d_amd.distributed import StripRenderer, init_process_group

    torch.cuda.set_device(0)
    init_process_group(world, rank, backend="gloo")
    dem = scenes.golden_dem()
    kw = scenes.fixed_frames(scenes.scene_kwargs(dem), 34, spp=4)  # crosses a Welford window
    r = StripRenderer(dem, 256, height, scenes.CAM, rank=rank, world=world, device=0, frames_in_flight=in_flight, peer_halos=peer_halos, **kw)
    r.run_frames(0, 32, collect_last=True)   # two calls: the second starts with the time-out count cleared and rising frame numbers
    r.window_variance(32)
    r.run_frames(32, 2, collect_last=True)
    var = r.window_variance(34)
    halo = r.session.halo_stats() if r.peer_halos else None
    image = r.gather_image(34)
    info = {"bounds": r.bounds, "balance_rounds": len(r.balance_log), "lanes": r.session.sample_lanes(),
            "in_flight": r.session.frames_in_flight(), "peer_halos": r.peer_halos,
            "halo_timeouts": r.session.halo_timeouts() if r.peer_halos else 0, "halo": halo,
            "peer_halo_failure": getattr(r, "peer_halo_failure", None)}
    if r.peer_halos:  # frame numbers of a connected session only rise (f3d_terrain_pt.h)
        try:
            r.session.enqueue_batch_strip(0, 1)
            info["replay_refused"] = False
        except ValueError as exc:
            info["replay_refused"] = "only rise" in str(exc)
    r.close()
    if rank == 0:
        image["variance"] = var
        image["info"] = info
        with open(out_path, "wb") as f:
            pickle.dump(image, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("in_flight,peer_halos", [(0, False), (6, False), (0, True), (6, True)])
def test_two_processes_on_one_gpu_reproduce_the_single_process_image(in_flight, peer_halos):
    """peer_halos: the strips pull each other's edge rows on the device (reservoirs mapped with hipIpcOpenMemHandle, frame
    counters polled by k_halo_pull) and a window of frames is one call into the library; else the classic exchange
    (point-to-point after every frame, staged through the host here)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    multi = _run_strips(2, port, in_flight, peer_halos, 200)
    _check_against_one_strip(multi, 2, in_flight, peer_halos, 200)


def _run_strips(world, port, in_flight, peer_halos, height, _retried=False):
    import torch.multiprocessing as mp

    out = tempfile.mktemp(suffix=".pkl")
    try:
        mp.spawn(_worker, args=(world, port, out, in_flight, peer_halos, height), nprocs=world, join=True)
    except Exception:  # noqa: BLE001 -- a rendezvous port taken between probing and use (seen once in ~20 runs): once more, on a new port
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_worker, args=(world, port, out, in_flight, peer_halos, height), nprocs=world, join=True)
    with open(out, "rb") as f:
        multi = pickle.load(f)
    os.unlink(out)
    if peer_halos and not multi["info"]["peer_halos"] and not _retried:
        # every rank fell back to the classic exchange: on this box all ranks share ONE GPU, and a link probe that waits on
        # the device can time out when the processes' queues are starved (seen twice in ~30 runs).  Say why, try once more.
        print("peer halos fell back:", multi["info"].get("peer_halo_failure"), file=sys.stderr)
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        return _run_strips(world, port, in_flight, peer_halos, height, _retried=True)
    return multi


def _check_against_one_strip(multi, world, in_flight, peer_halos, height):
    sys.path.insert(0, str(ROOT / "tests"))
    import scenes
    from forge3d_amd.session import TerrainSession

    dem = scenes.golden_dem()
    kw = scenes.fixed_frames(scenes.scene_kwargs(dem), 34, spp=4)
    with TerrainSession(dem, 256, height, scenes.CAM, **kw) as sess:
        sess.enqueue_frames(0, 34, True)
        m2, bad = sess.window_stats()
        single = sess.resolve(34)
    assert not bad
    info = multi["info"]
    assert info["peer_halos"] == peer_halos and info["halo_timeouts"] == 0, info.get("peer_halo_failure")
    if in_flight is not None:
        assert info["in_flight"] == in_flight  # 6: batches traced in one launch, halos exchanged between the merges
    # (>= 1: the measured loop stops as soon as a re-partition from the gathered times repeats the boundaries it timed --
    # with noisy one-GPU timings that is sometimes the equal split itself, seen 2 in ~40 runs)
    assert info["balance_rounds"] >= 1 and info["bounds"][0] == 0 and info["bounds"][-1] == height and len(info["bounds"]) == world + 1
    if peer_halos:  # rank 0 pulls from the strip below it only: one block per frame, and the waits were timed
        assert info["halo"]["pulls"] == 34 and info["halo"]["timeouts"] == 0 and info["halo"]["frames_published"] == 34
        assert info["halo"]["wait_ms"][0] == 0.0 and info["halo"]["longest_wait_ms"] < info["halo"]["timeout_ms"]
        assert info["replay_refused"] is True
    assert np.float32(multi["variance"]) == np.float32(max(0.0, m2) / np.float32(1.0))  # frame 34: window of 2
    for key in ("rgba", "albedo", "normal", "depth"):
        assert np.array_equal(multi[key], single[key], equal_nan=True), key


@pytest.mark.gpu
@pytest.mark.parametrize("world,in_flight", [(4, 0), (4, None), (8, None)])
def test_four_and_eight_processes_on_one_gpu_with_peer_halos(world, in_flight):
    """The node-sized job on the one GPU of the test box: `world` OS processes, real IPC handles, INTERIOR ranks with a
    neighbour on both sides, `world`-way measured balancing.  in_flight None = the driver's default for that many ranks
    (16 frames in flight from 4 ranks on): the path bench.py --gpus 8 takes.  Collectives over gloo (RCCL refuses several
    ranks per device; its own smoke test is test_gpu_nccl_smoke.py).  The stitched image must equal the one-strip image."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    multi = _run_strips(world, port, in_flight, True, 240)
    if in_flight is None:
        assert multi["info"]["in_flight"] == 16
    _check_against_one_strip(multi, world, in_flight, True, 240)
