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


def _scene_kwargs(mesh):
    """The golden scene, optionally with buildings on it (the mesh-capable strip kernels: BASELINE.json configs[3])."""
    sys.path.insert(0, str(ROOT / "tests"))
    import scenes

    dem = scenes.golden_dem()
    kw = scenes.fixed_frames(scenes.scene_kwargs(dem), 34, spp=4)  # crosses a Welford window
    if mesh:
        from forge3d_amd import datasets

        v, i = datasets.proxy_buildings(dem * np.float32(kw["exaggeration"]), kw["spacing"][0], n_boxes=400, seed=11)
        v = v.astype(np.float32)
        # (proxy_buildings sizes its boxes for a 10 m DEM: shrink them to this scene's 0.8-unit cells)
        centre = v.reshape(-1, 8, 3).mean(axis=1, keepdims=True)
        v = (centre + (v.reshape(-1, 8, 3) - centre) * np.float32(0.08)).reshape(-1, 3).astype(np.float32)
        kw = dict(kw, mesh_vertices=v, mesh_indices=i)
    return dem, kw


def _worker(rank, world, port, out_path, in_flight=0, peer_halos=True, height=200, mesh=False):
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
    dem, kw = _scene_kwargs(mesh)
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


def _run_strips(world, port, in_flight, peer_halos, height, _retried=False, mesh=False):
    import torch.multiprocessing as mp

    out = tempfile.mktemp(suffix=".pkl")
    try:
        mp.spawn(_worker, args=(world, port, out, in_flight, peer_halos, height, mesh), nprocs=world, join=True)
    except Exception:  # noqa: BLE001 -- a rendezvous port taken between probing and use (seen once in ~20 runs): once more, on a new port
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_worker, args=(world, port, out, in_flight, peer_halos, height, mesh), nprocs=world, join=True)
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
        return _run_strips(world, port, in_flight, peer_halos, height, _retried=True, mesh=mesh)
    return multi


def _check_against_one_strip(multi, world, in_flight, peer_halos, height, mesh=False):
    sys.path.insert(0, str(ROOT / "tests"))
    import scenes
    from forge3d_amd.session import TerrainSession

    dem, kw = _scene_kwargs(mesh)
    with TerrainSession(dem, 256, height, scenes.CAM, **kw) as sess:
        sess.enqueue_frames(0, 34, True)
        m2, bad = sess.window_stats()
        single = sess.resolve(34)
    assert not bad
    info = multi["info"]
    assert info["peer_halos"] == peer_halos and info["halo_timeouts"] == 0, info.get("peer_halo_failure")
    if in_flight is not None:
        assert info["in_flight"] == in_flight  # 6: batches traced in one launch, halos exchanged between the merges
    # (1: the cut from rank 0's row-cost map, round 5; the measured refinement rounds are opt-in)
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


@pytest.mark.gpu
def test_eight_processes_with_a_mesh_in_the_scene():
    """BASELINE.json configs[3] names 8 GPUs: the mesh-capable strip kernels (k_trace / k_merge / the fused form with the mesh
    walk) as eight processes over real IPC handles on the one GPU, 16 frames in flight; the stitched image must equal the
    one-strip image of the same scene, buildings included."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    multi = _run_strips(8, port, None, True, 240, mesh=True)
    assert multi["info"]["in_flight"] == 16
    _check_against_one_strip(multi, 8, None, True, 240, mesh=True)
    sys.path.insert(0, str(ROOT / "tests"))
    dem, kw = _scene_kwargs(True)
    from forge3d_amd.session import TerrainSession

    plain = dict(kw)
    plain.pop("mesh_vertices"), plain.pop("mesh_indices")
    with TerrainSession(dem, 256, 240, __import__("scenes").CAM, **plain) as sess:
        sess.enqueue_frames(0, 34)
        bare = sess.resolve(34)
    assert not np.array_equal(bare["rgba"], multi["rgba"])  # the buildings are in the picture
