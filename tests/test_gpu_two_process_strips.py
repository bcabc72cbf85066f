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


def _worker(rank, world, port, out_path, in_flight=0, peer_halos=True):
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
    r = StripRenderer(dem, 256, 200, scenes.CAM, rank=rank, world=world, device=0, frames_in_flight=in_flight, peer_halos=peer_halos, **kw)
    r.run_frames(0, 34, collect_last=True)
    var = r.window_variance(34)
    image = r.gather_image(34)
    info = {"bounds": r.bounds, "balance_rounds": len(r.balance_log), "lanes": r.session.sample_lanes(),
            "in_flight": r.session.frames_in_flight(), "peer_halos": r.peer_halos,
            "halo_timeouts": r.session.halo_timeouts() if r.peer_halos else 0}
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
    import torch.multiprocessing as mp

    sys.path.insert(0, str(ROOT / "tests"))
    import scenes
    from forge3d_amd.session import TerrainSession

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tempfile.mktemp(suffix=".pkl")
    mp.spawn(_worker, args=(2, port, out, in_flight, peer_halos), nprocs=2, join=True)
    with open(out, "rb") as f:
        multi = pickle.load(f)
    os.unlink(out)
    dem = scenes.golden_dem()
    kw = scenes.fixed_frames(scenes.scene_kwargs(dem), 34, spp=4)
    with TerrainSession(dem, 256, 200, scenes.CAM, **kw) as sess:
        sess.enqueue_frames(0, 34, True)
        m2, bad = sess.window_stats()
        single = sess.resolve(34)
    assert not bad
    assert multi["info"]["peer_halos"] == peer_halos and multi["info"]["halo_timeouts"] == 0
    assert multi["info"]["in_flight"] == in_flight  # 6: batches traced in one launch, halos exchanged between the merges
    assert multi["info"]["balance_rounds"] >= 2 and multi["info"]["bounds"][0] == 0 and multi["info"]["bounds"][-1] == 200
    assert np.float32(multi["variance"]) == np.float32(max(0.0, m2) / np.float32(1.0))  # frame 34: window of 2
    for key in ("rgba", "albedo", "normal", "depth"):
        assert np.array_equal(multi[key], single[key], equal_nan=True), key
