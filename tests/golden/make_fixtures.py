#!/usr/bin/env python
"""Regenerates tests/golden/wrapper_contract.json from the reference checkout.

Runs ONLY in the build container (needs /root/reference); the JSON it writes is data --
the reference wrapper's signature model and (input -> exception type, message) pairs of its
pure-Python validation -- captured the way the reference's own test does it, with the
native module monkeypatched (reference tests/test_hybrid_terrain_pt.py:860-876).
Nothing from /root/reference travels to the GPU box.
"""
import inspect
import json
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference/python")
sys.path.insert(0, str(REF))
import forge3d.path_tracing as pt  # noqa: E402


class _Native:
    calls = []

    @staticmethod
    def hybrid_render_terrain_reference(*args, **kwargs):
        _Native.calls.append(kwargs)
        return {}


pt._NATIVE = _Native()
sig = inspect.signature(pt.hybrid_render_terrain_reference)
model = []
for p in sig.parameters.values():
    default = "<required>" if p.default is inspect._empty else p.default
    ann = "" if p.annotation is inspect._empty else str(p.annotation)
    if len(ann) >= 2 and ann[0] in "'\"" and ann[-1] == ann[0]:
        ann = ann[1:-1]
    model.append([p.name, p.kind.name, default if not isinstance(default, tuple) else list(default), ann])

dem = np.zeros((4, 4), np.float32)
CAM = {"origin": (0.0, 35.0, 90.0), "look_at": (0.0, 5.0, 0.0), "up": (0.0, 1.0, 0.0), "fov_y": 45.0}
cases = {
    "ndim": dict(heightmap="zeros3d"),
    "tiny": dict(heightmap="zeros1x1"),
    "nan": dict(heightmap="nan16"),
    "min_gt_max": dict(max_frames=4, min_frames=8),
    "spp0": dict(spp=0),
    "spp65": dict(spp=65),
    "spacing0": dict(spacing=(0.0, 1.0)),
    "sun_nan": dict(sun_color=(1.0, float("nan"), 1.0)),
    "sun_neg": dict(sun_color=(1.0, -0.1, 1.0)),
    "sun_two": dict(sun_color=(1.0, 1.0)),
    "sun_four": dict(sun_color=(1.0, 1.0, 1.0, 1.0)),
    "sun_scalar": dict(sun_color=0.5),
    "sun_str": dict(sun_color="abc"),
    "sun_strs": dict(sun_color=("0.5", "0.9", "0.8")),
    "mesh_alone": dict(mesh_vertices="zeros3x3"),
    "env_shape": dict(env_map="zeros4x4"),
    "mesh_shape": dict(mesh_vertices="zeros3x2", mesh_indices="idx1x3"),
}
arrays = {"zeros3d": np.zeros((2, 2, 2), np.float32), "zeros1x1": np.zeros((1, 1), np.float32),
          "nan16": np.full((16, 16), np.nan, np.float32), "zeros3x3": np.zeros((3, 3), np.float32),
          "zeros4x4": np.zeros((4, 4), np.float32), "zeros3x2": np.zeros((3, 2), np.float32),
          "idx1x3": np.zeros((1, 3), np.uint32)}
errors = {}
for name, kw in cases.items():
    kw = {k: (arrays[v] if isinstance(v, str) and v in arrays else v) for k, v in kw.items()}
    hm = kw.pop("heightmap", dem)
    try:
        pt.hybrid_render_terrain_reference(hm, 8, 8, CAM, **kw)
        errors[name] = None
    except Exception as exc:  # noqa: BLE001
        errors[name] = [type(exc).__name__, str(exc)]

# what the wrapper forwards to the native function for a default call
_Native.calls.clear()
pt.hybrid_render_terrain_reference(dem, 8, 8, CAM)
forwarded = {k: (list(v) if isinstance(v, tuple) else v) for k, v in _Native.calls[0].items()
             if isinstance(v, (int, float, str, tuple, bool, type(None)))}

out = {"signature": model, "errors": errors, "forwarded_defaults": forwarded}
Path(__file__).with_name("wrapper_contract.json").write_text(json.dumps(out, indent=1) + "\n")
print("wrote wrapper_contract.json:", len(model), "parameters,", len(errors), "error cases")
