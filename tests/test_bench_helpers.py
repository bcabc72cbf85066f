"""Host logic of bench.py that only an N > 1 run exercises on hardware: the one re-cut of the strips after the warm-up frames
 and the identity fields.  No GPU."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


class _Renderer:
    def __init__(self, bounds, density=None):
        self.bounds, self.world, self.height = list(bounds), len(bounds) - 1, bounds[-1]
        if density is not None:
            self.cost_density = np.asarray(density, np.float64)
        self.closed = False

    def close(self):
        self.closed = True


def test_balanced_strips_are_left_alone():
    r = _Renderer([0, 400, 700, 1080])
    made = []
    got, report = bench.recut_after_warmup(r, [4.0, 4.1, 4.05], lambda b: made.append(b) or _Renderer(b))
    assert got is r and not r.closed and made == [] and report["recut"] is False and report["max_over_mean"] < 1.05


def test_an_imbalance_over_five_per_cent_recuts_once_towards_equal_cost():
    density = np.where(np.arange(1080) < 400, 0.2, 1.0)
    r = _Renderer([0, 360, 720, 1080], density)  # equal rows on a frame whose top is cheap sky
    times = [density[b0:b1].sum() / 100.0 for b0, b1 in zip(r.bounds, r.bounds[1:])]
    made = []
    got, report = bench.recut_after_warmup(r, times, lambda b: made.append(list(b)) or _Renderer(b, density))
    assert report["recut"] is True and r.closed and got is not r and made == [report["bounds_after"]]
    cost = [density[b0:b1].sum() for b0, b1 in zip(got.bounds, got.bounds[1:])]
    assert max(cost) / (sum(cost) / 3) < 1.03 and got.bounds[1] > 360  # the sky strip got fatter
    assert report["bounds_before"] == [0, 360, 720, 1080]


def test_a_new_cut_that_cannot_be_built_keeps_the_old_strips():
    r = _Renderer([0, 360, 720, 1080], np.where(np.arange(1080) < 400, 0.2, 1.0))

    def broken(bounds):
        raise RuntimeError("[Render] Render error: another rank of the strip job failed; this rank stops with it")

    got, report = bench.recut_after_warmup(r, [1.0, 3.0, 3.6], broken)
    assert got is r and not r.closed and report["recut"] is False and "another rank" in report["failed"]


def test_useless_times_never_recut():
    r = _Renderer([0, 540, 1080])
    for times in ([0.0, 1.0], [float("nan"), 1.0], [float("inf"), 1.0]):
        got, report = bench.recut_after_warmup(r, times, lambda b: (_ for _ in ()).throw(AssertionError("must not be called")))
        assert got is r and report["recut"] is False


def test_device_identity_reports_whatever_the_torch_build_exposes():
    class Props:
        name = "AMD Instinct MI355X"
        uuid = "abc"
        pci_bus_id = 5

    class Cuda:
        @staticmethod
        def get_device_properties(i):
            return Props()

    class Torch:
        cuda = Cuda()

    ident = bench.device_identity(Torch(), 3)
    assert ident == {"index": 3, "name": "AMD Instinct MI355X", "uuid": "abc", "pci_bus_id": "5"}
