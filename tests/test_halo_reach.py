"""The spatial pass reaches FOUR rows down, not three: its neighbour offset is floor(u * 7) - 3 with u = f32(x) / 2^32
(reference pt_restir_spatial.wgsl:171-176), and f32(x) rounds to 2^32 for the top 128 values of x, so u is exactly 1.0
for 2^-25 of the draws.  A strip therefore needs 4 halo rows below it (f3d_scene.h kHaloRows).  Found by
tools/gpu_fuzz_fd.py (scene 4237: pixel (62, 10) of frame 1 looks at (61, 14)); with a 3-row halo that read left the
strip's reservoir buffer."""
from __future__ import annotations

import numpy as np
import pytest

import scenes
from emul import emul


def _strips(dem, size, cam, kw, bounds, frames, rows_exchanged, bogus_beyond=False):
    import torch

    from forge3d_amd.session import HALO_ROWS as R, reservoir_buffer_bytes

    W = size[0]
    backend = emul.EmulBackend()
    bufs, sessions = [], []
    for b, e in bounds:
        res = [torch.zeros(reservoir_buffer_bytes(e - b, W), dtype=torch.uint8) for _ in range(2)]
        bufs.append(res)
        sessions.append(backend.make_session(dem, W, size[1], cam, b, e, res, backend.empty_i32(4), kw))
    row = W * 16
    n = rows_exchanged
    for f in range(frames):
        for s in sessions:
            s.enqueue_frames(f, 1, f + 1 == frames)
        for i in range(len(bounds) - 1):
            up, dn = bufs[i][f & 1], bufs[i + 1][f & 1]
            rows_up = bounds[i][1] - bounds[i][0]
            dn[(R - n) * row:R * row] = up[(rows_up + R - n) * row:(rows_up + R) * row]  # my bottom rows -> their top halo
            up[(rows_up + R) * row:(rows_up + R + n) * row] = dn[R * row:(R + n) * row]   # their top rows -> my bottom halo
            if bogus_beyond:  # rows the exchange left out hold a valid sun reservoir nobody computed
                rec = np.array([5.0, 0.0, 5.0, 1.0], np.float32)  # w_sum, m | sun bit, weight, target_pdf
                rec[1:2].view(np.uint32)[0] = 0x80000001
                fake = torch.from_numpy(np.tile(rec.view(np.uint8), W * (R - n)))
                up[(rows_up + R + n) * row:(rows_up + 2 * R) * row] = fake
    out = np.concatenate([s.resolve(frames)["rgba"] for s in sessions], axis=0)
    for s in sessions:
        s.close()
    return out


def test_a_unit_random_number_reaches_the_fourth_row_below():
    """Golden scene, seed 17256: neighbour 1 of pixel (18, 27) in the spatial pass of frame 0 is (x, 31)."""
    dem, size = scenes.golden_dem(4), (72, 50)
    kw = scenes.fixed_frames(dict(scenes.scene_kwargs(dem), seed=17256), 3, spp=2)
    full = emul.render(dem, size[0], size[1], scenes.CAM, **kw)["rgba"].reshape(size[1], size[0], 4)
    from oracle import oracle

    # the reference's arithmetic, restated: the oracle walks the full image and makes the same +4 read
    assert np.array_equal(oracle.render(dem, size[0], size[1], scenes.CAM, **kw)["rgba"].reshape(size[1], size[0], 4), full)
    bounds = [(0, 28), (28, size[1])]  # pixel row 27 is the upper strip's last, row 31 the lower strip's fourth
    assert np.array_equal(_strips(dem, size, scenes.CAM, kw, bounds, 3, 4), full)
    # the nominal radius 3 is not enough: without the fourth row the chain that starts at (18, 27) goes wrong
    three = np.argwhere((_strips(dem, size, scenes.CAM, kw, bounds, 3, 3) != full).any(-1)).tolist()
    assert [27, 18] in three and len(three) <= 4, three
    kw2 = dict(kw, max_frames=2, min_frames=2)
    full2 = emul.render(dem, size[0], size[1], scenes.CAM, **kw2)["rgba"].reshape(size[1], size[0], 4)
    bogus = _strips(dem, size, scenes.CAM, kw2, bounds, 2, 3, bogus_beyond=True)
    assert np.argwhere((bogus != full2).any(-1)).tolist() == [[27, 18]]


def test_a_lone_strip_reads_only_its_own_halo_rows():
    """tools/gpu_fuzz_fd.py scene 4237, rows 4..11 alone: pixel (62, 10) of frame 1 looks at (61, 14).  With a 3-row
    halo that read left the reservoir buffer, and the image depended on what the allocator had put behind it."""
    dem, size, cam, kw = scenes.random_scene(4237)
    kw = dict(kw, max_frames=2, min_frames=2, variance_threshold=1e30)
    lone = emul.render(dem, size[0], size[1], cam, rows=(4, 11), **kw)["rgba"].reshape(7, size[0], 4)
    assert np.array_equal(_strips(dem, size, cam, kw, [(4, 11)], 2, 0), lone)


@pytest.mark.gpu
def test_gpu_lone_strip_is_the_same_image_in_every_session():
    """The HIP sessions of that strip -- fused frames and frames in flight, created one after the other so that the
    allocator hands them different neighbours -- all give the emulator's image."""
    from forge3d_amd.session import TerrainSession

    dem, size, cam, kw = scenes.random_scene(4237)
    for frames in (2, 22):
        kw = dict(kw, max_frames=frames, min_frames=frames, variance_threshold=1e30)
        want = emul.render(dem, size[0], size[1], cam, rows=(4, 11), **kw)["rgba"].reshape(7, size[0], 4)
        for in_flight in (0, 3, 0, 16, 0):
            with TerrainSession(dem, size[0], size[1], cam, kernel_variant=4000000, frames_in_flight=in_flight,
                                memory_budget_bytes=8 << 30, row_begin=4, row_end=11, **kw) as s:
                s.enqueue_frames(0, frames, True)
                s.window_stats()
                assert np.array_equal(s.resolve(frames)["rgba"], want), (frames, in_flight)


def test_the_draw_that_equals_one():
    """f32(x) / 2^32 == 1.0 for x >= 2^32 - 128, and floor(7 * 1.0) - 3 == 4."""
    x = np.array([2**32 - 129, 2**32 - 128, 2**32 - 1], np.uint32)
    u = x.astype(np.float32) / np.float32(4294967296.0)
    assert u.tolist() == [np.float32(1.0) - np.float32(2.0**-24), 1.0, 1.0]
    assert (np.floor(u * np.float32(7.0)).astype(int) - 3).tolist() == [3, 4, 4]
