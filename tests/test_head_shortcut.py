"""The invariant behind k_head's shortcut (csrc/f3d_frame.h head_neighbourhood_empty, round 4; window corrected in round 5):
a pixel whose [-3, +4] x [-3, +4] neighbourhood -- the spatial pass's real reach: floor(u * 7) - 3 with u == 1.0 for 2^-25
of the draws (pt_restir_spatial.wgsl:199-204) -- holds no reservoir sample (m == 0 everywhere) gets, from the real frame head (csrc/f3d_shade.h frame_head /
spatial_reuse, here compiled for the host), exactly what the shortcut writes without running it -- the pixel's own
light-type bit and target pdf around zeros, and the "no usable history" head record -- whatever the other words of the
empty records hold.  The kernel itself is covered by the device suite (every sample-lane render with sky in it) and the
fuzzers; this pins the claim on the CPU."""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest

from emul import emul

HALO = 4


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_frame_head_of_an_empty_neighbourhood_is_what_the_shortcut_writes(seed):
    rng = np.random.default_rng(seed)
    w, h = 96, 64
    res = np.zeros((h + 2 * HALO, w, 4), np.uint32)
    f = res.view(np.float32)
    # every word random first: w_sum, weight and target pdf of an EMPTY record are not zero in general
    f[..., 0] = rng.uniform(0.0, 50.0, res.shape[:2])
    f[..., 2] = rng.uniform(0.0, 4.0, res.shape[:2])
    f[..., 3] = rng.uniform(0.0, 2.0, res.shape[:2])
    m = rng.integers(1, 500, res.shape[:2]).astype(np.uint32)
    # blobs of pixels that hold samples; everywhere else m = 0
    yy, xx = np.mgrid[0:h + 2 * HALO, 0:w]
    holds = np.zeros(res.shape[:2], bool)
    for _ in range(6):
        cx, cy, r = rng.integers(0, w), rng.integers(0, h + 2 * HALO), rng.integers(3, 14)
        holds |= (xx - cx) ** 2 + (yy - cy) ** 2 < r * r
    m[~holds] = 0
    res[..., 1] = m | (rng.integers(0, 2, res.shape[:2]).astype(np.uint32) << 31)  # light-type bit random, also on empty records
    g = np.zeros((h, w, 4), np.float32)
    g[..., :3] = rng.normal(size=(h, w, 3))
    g[..., 3] = rng.integers(0, 2, (h, w))
    g[g[..., 3] == 0, :3] = 0.0  # the G-buffer of a miss
    wi = np.asarray([0.3, 0.8, 0.52], np.float32)
    applies = C.c_uint32(0)
    fn = emul.lib().emul_head_shortcut_mismatches
    fn.restype = C.c_uint32
    for frame in (1, 7, 200):
        bad = fn(C.c_uint32(w), C.c_uint32(h), C.c_uint32(frame), res.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.POINTER(C.c_float)),
                 wi.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(0x9E3779B9 ^ seed), C.c_uint32(0x85EBCA6B), C.byref(applies),
                 C.c_uint32(4))
        assert applies.value > 500  # a good part of the image, incl. pixels beside the blobs' reach
        assert bad == 0


# ---- the constructed case: a neighbour draw that IS 1.0 --------------------------------------------------------------
M32 = 0xFFFFFFFF


def xorshift(x: int) -> int:
    x ^= (x << 13) & M32
    x ^= x >> 17
    x ^= (x << 5) & M32
    return x


def _undo_left(y: int, k: int) -> int:  # x from y = x ^ (x << k)
    x = y
    for _ in range(32 // k + 1):
        x = y ^ ((x << k) & M32)
    return x


def _undo_right(y: int, k: int) -> int:  # x from y = x ^ (x >> k)
    x = y
    for _ in range(32 // k + 1):
        x = y ^ (x >> k)
    return x


def xorshift_inverse(y: int) -> int:
    return _undo_left(_undo_right(_undo_left(y, 5), 17), 13)


def offset_of(x: int) -> int:
    """floor(f32(x) / 2^32 * 7) - 3 as the pass computes it (f3d_shade.h spatial_reuse, neighbour)."""
    u = np.float32(x) / np.float32(4294967296.0)
    return int(np.floor(np.float32(u) * np.float32(7.0))) - 3


def unit_draw_case(width: int, gx: int, gy: int, frame: int, axis: int, pick: int = 0):
    """A user seed for which the FIRST neighbour of pixel (gx, gy) in the spatial pass that the head of frame `frame`
    evaluates -- the pass of frame - 1 (f3d_shade.h frame_head) -- is drawn with u == 1.0 on `axis` (0: x, 1: y), given that
    the pixel itself draws nothing (its own m == 0).  Returns (seed, (rx, ry)).
    Stream: seed0 = (seed ^ (frame - 1)) + idx * 1664525 + 1013904223; draws rx then ry."""
    frame = frame - 1
    top = (M32 - pick) & M32  # one of the 128 values that round to 2^32 in f32
    assert np.float32(top) == np.float32(4294967296.0)
    if axis == 0:
        s1 = top
    else:
        s1 = xorshift_inverse(top)
    seed0 = xorshift_inverse(s1)
    s2 = xorshift(s1)
    rx, ry = offset_of(s1), offset_of(s2)
    assert (rx, ry)[axis] == 4
    idx = gy * width + gx
    seed = ((seed0 - idx * 1664525 - 1013904223) & M32) ^ frame
    assert xorshift(((seed ^ frame) + idx * 1664525 + 1013904223) & M32) == s1
    return seed, (rx, ry)


def planted_reservoirs(width, height, at, halo=HALO):
    """(height + 2 halo, width, 4) u32: every record empty except the one at pixel `at` = (x, y)."""
    res = np.zeros((height + 2 * halo, width, 4), np.uint32)
    f = res.view(np.float32)
    f[..., 3] = 1.0
    x, y = at
    f[y + halo, x, 0] = 5.0             # w_sum
    res[y + halo, x, 1] = 0x80000000 | 17  # m = 17, sun sample
    f[y + halo, x, 2] = 1.0             # weight
    return res


@pytest.mark.parametrize("axis", [0, 1])
@pytest.mark.parametrize("pick", [0, 77, 127])
def test_a_unit_draw_reaches_the_fourth_pixel_and_the_three_pixel_window_misses_it(axis, pick):
    """Round 4's window ([-3, +3]) called this pixel's neighbourhood empty; frame_head adds the +4 neighbour's m."""
    w, h, frame = 96, 64, 5
    gx, gy = 39, 23  # last column and last row of their 8x8 tile
    seed, (rx, ry) = unit_draw_case(w, gx, gy, frame, axis, pick)
    assert max(rx, ry) == 4 and min(rx, ry) >= -3
    res = planted_reservoirs(w, h, (gx + rx, gy + ry))
    g = np.zeros((h, w, 4), np.float32)
    g[..., 1] = 1.0
    g[..., 3] = 1.0
    wi = np.asarray([0.3, 0.8, 0.52], np.float32)
    fn = emul.lib().emul_head_shortcut_mismatches
    fn.restype = C.c_uint32
    applies = C.c_uint32(0)

    def mismatches(reach_hi):
        return fn(C.c_uint32(w), C.c_uint32(h), C.c_uint32(frame), res.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.POINTER(C.c_float)),
                  wi.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(seed), C.c_uint32(seed ^ 0x85EBCA6B), C.byref(applies),
                  C.c_uint32(reach_hi))

    assert mismatches(3) == 1, "the case is not the one it was constructed to be"
    assert mismatches(4) == 0


def planted_frame(make_session, zeros, sync, planted, width, height, halo=HALO):
    """Frame 0 as rendered, the planted records in place of its reservoir output, frame 1; -> (frame 1's reservoirs of the
    image rows as (H, W, 4) u32, the resolved images).  Shared with tests/test_gpu_head_reach.py."""
    import torch

    from forge3d_amd.session import reservoir_buffer_bytes

    res = [zeros(reservoir_buffer_bytes(height, width)) for _ in range(2)]
    s = make_session(res)
    s.enqueue_frames(0, 1, False)
    sync()
    res[0].copy_(torch.from_numpy(planted.view(np.uint8).reshape(-1)))  # frame 0 wrote buffer 0; frame 1's head reads it
    sync()
    s.enqueue_frames(1, 1, False)
    sync()
    out = res[1].cpu().numpy().view(np.uint32).reshape(height + 2 * halo, width, 4)[halo:halo + height].copy()
    img = s.resolve(2)
    s.close()
    return out, img


PLANT_W, PLANT_H, PLANT_PIXEL = 96, 64, (39, 47)  # a sun-facing terrain pixel of the golden scene, last column and row of its 8x8 tile


@pytest.mark.parametrize("axis", [0, 1])
def test_the_planted_frame_depends_on_the_record_four_pixels_out(axis):
    """The device test's premise, on the host: the kernel code (no shortcut here) merges the +4 neighbour's 17 samples into
    the pixel's history, so a head that skips it leaves a different reservoir behind."""
    import scenes

    dem = scenes.golden_dem(4)
    gx, gy = PLANT_PIXEL
    seed, (rx, ry) = unit_draw_case(PLANT_W, gx, gy, 1, axis)
    kw = scenes.fixed_frames(dict(scenes.scene_kwargs(dem), seed=seed), 2, spp=4)
    backend = emul.EmulBackend()

    def run(planted):
        return planted_frame(lambda res: backend.make_session(dem, PLANT_W, PLANT_H, scenes.CAM, 0, PLANT_H, res, backend.empty_i32(4), kw),
                             backend.empty_bytes, lambda: None, planted, PLANT_W, PLANT_H)[0]

    with_sample = run(planted_reservoirs(PLANT_W, PLANT_H, (gx + rx, gy + ry)))
    nothing = planted_reservoirs(PLANT_W, PLANT_H, (0, 0))
    nothing[HALO, 0, :3] = 0
    without = run(nothing)
    m = lambda r: int(r[gy, gx, 1] & 0x7FFFFFFF)
    assert m(with_sample) == m(without) + 17
