"""RCCL loaded once: the product's collectives on the `nccl` backend with the one rank a one-GPU box can give it.

`torch.distributed`'s nccl backend IS RCCL on ROCm.  The multi-GPU driver (forge3d_amd/distributed.py) is covered by gloo
tests on the CPU emulator (world 2 / 3) and by several OS processes sharing the test box's GPU over real IPC handles
(tests/test_gpu_two_process_strips.py) -- but RCCL refuses two ranks per device, so none of those ever initialised it.
Here a world of ONE rank goes through every collective the driver issues on device tensors (all_gather of the halo
exports, the agreement all-reduces, the barriers, the per-window all-reduce of the statistics record, the device gather of
the resolved strips through StripRenderer.gather_image's on_device branch) with `force_collectives`, and the image must
be the plain single-GPU image."""
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


def _worker(rank, port, out_path):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import scenes
    from forge3d_amd.distributed import StripRenderer, init_process_group

    torch.cuda.set_device(0)
    init_process_group(1, 0, backend="nccl", force=True)
    assert dist.get_backend() == "nccl"
    probe = torch.arange(8, dtype=torch.float32, device="cuda")  # RCCL itself, before the driver uses it
    dist.all_reduce(probe, op=dist.ReduceOp.MAX)
    assert probe.tolist() == list(range(8))
    dem = scenes.golden_dem()
    kw = scenes.fixed_frames(scenes.scene_kwargs(dem), 34, spp=4)
    r = StripRenderer(dem, 256, 200, scenes.CAM, rank=0, world=1, device=0, force_collectives=True, **kw)
    r.run_frames(0, 34, collect_last=True)
    var = r.window_variance(34)
    comm_device = str(r._comm_device())
    image = r.gather_image(34)
    image["info"] = {"peer_halos": r.peer_halos, "comm_device": comm_device, "variance": var, "balance_rounds": len(r.balance_log),
                     "halo": r.session.halo_stats() if r.peer_halos else None}
    r.close()
    with open(out_path, "wb") as f:
        pickle.dump(image, f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_world_of_one_on_the_nccl_backend_renders_the_single_gpu_image():
    import torch.multiprocessing as mp

    sys.path.insert(0, str(ROOT / "tests"))
    import scenes
    from forge3d_amd.session import TerrainSession

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tempfile.mktemp(suffix=".pkl")
    mp.spawn(_worker, args=(port, out), nprocs=1, join=True)  # its own process: the process group must not outlive the test
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
    info = multi["info"]
    assert info["comm_device"].startswith("cuda")  # the collectives moved DEVICE tensors (gather_image's on_device branch)
    assert info["peer_halos"] is True and info["halo"]["pulls"] == 0 and info["halo"]["frames_published"] == 34
    assert info["balance_rounds"] >= 1
    assert np.float32(info["variance"]) == np.float32(max(0.0, m2) / np.float32(1.0))
    for key in ("rgba", "albedo", "normal", "depth"):
        assert np.array_equal(multi[key], single[key], equal_nan=True), key
