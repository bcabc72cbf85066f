"""ctypes front-end of oracle/aether_bake_oracle.c -- TEST INFRASTRUCTURE ONLY (see the C file's header)."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SRC, _LIB = _HERE / "aether_bake_oracle.c", _HERE / "libaether_bake_oracle.so"

DEFAULT_DIMS = dict(transmittance_mu=32, transmittance_height=8, scattering_mu_view=17, scattering_mu_sun=17, scattering_height=8,
                    scattering_nu=16, aerial_distance=8, aerial_mu_view=8, aerial_height=8)          # bake.rs:44-58
DEFAULT_CONFIG = dict(turbidity=2.0, ozone_du=300.0, mie_g=0.8, bottom_radius_m=6_360_000.0, top_radius_m=6_460_000.0,
                      rayleigh_scale_height_m=8_000.0, mie_scale_height_m=1_200.0, max_aerial_distance_m=160_000.0,
                      ground_albedo=0.3, scattering_orders=4)                                          # bake.rs:146-162


class Config(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("turbidity", "ozone_du", "mie_g", "bottom_radius_m", "top_radius_m", "rayleigh_scale_height_m",
                                         "mie_scale_height_m", "max_aerial_distance_m", "ground_albedo")] + \
               [("scattering_orders", C.c_uint32)] + [(n, C.c_uint32) for n in DEFAULT_DIMS]


def build(force: bool = False) -> Path:
    if force or not _LIB.exists() or _LIB.stat().st_mtime < _SRC.stat().st_mtime:
        subprocess.run(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-fPIC", "-shared", str(_SRC), "-o", str(_LIB), "-lm"],
                       check=True, capture_output=True)
    return _LIB


def bake(**overrides):
    """-> dict(transmittance, single, accumulated, aerial: uint16 RGBA16F bit patterns shaped like the reference's tables,
    deltas: float32[scattering_orders]).  Keyword overrides of DEFAULT_CONFIG / DEFAULT_DIMS."""
    build()
    lib = C.CDLL(str(_LIB))
    cfg = Config()
    for k, v in {**DEFAULT_CONFIG, **DEFAULT_DIMS, **overrides}.items():
        setattr(cfg, k, v)
    d = cfg
    t = np.zeros((d.transmittance_height, d.transmittance_mu, 4), np.uint16)
    shape = (d.scattering_height, d.scattering_nu, d.scattering_mu_sun, d.scattering_mu_view, 4)
    single, accumulated = np.zeros(shape, np.uint16), np.zeros(shape, np.uint16)
    aerial = np.zeros((d.aerial_height, d.aerial_mu_view, d.aerial_distance, 4), np.uint16)
    deltas = np.zeros(d.scattering_orders, np.float32)
    rc = lib.abo_bake(C.byref(cfg), t.ctypes.data_as(C.c_void_p), single.ctypes.data_as(C.c_void_p), accumulated.ctypes.data_as(C.c_void_p),
                      aerial.ctypes.data_as(C.c_void_p), deltas.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError(f"aether bake oracle: status {rc}")
    return {"transmittance": t, "single": single, "accumulated": accumulated, "aerial": aerial, "deltas": deltas}
