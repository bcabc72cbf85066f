"""The invariant behind k_head's shortcut (csrc/f3d_frame.h head_neighbourhood_empty, round 4): a pixel whose 7 x 7
neighbourhood holds no reservoir sample (m == 0 everywhere) gets, from the real frame head (csrc/f3d_shade.h frame_head /
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
                 wi.ctypes.data_as(C.POINTER(C.c_float)), C.c_uint32(0x9E3779B9 ^ seed), C.c_uint32(0x85EBCA6B), C.byref(applies))
        assert applies.value > 500  # a good part of the image, incl. pixels beside the blobs' 3-pixel reach
        assert bad == 0
