"""AETHER atmosphere transport for the terrain path tracer: configuration, LUT payloads and their provenance.

Reference surface: ``AtmosphereConfig`` / ``LutDimensions`` (src/core/atmosphere/bake.rs:30-243), the shipped
five-anchor turbidity bank and its interpolation (src/core/atmosphere/precomputed.rs:5-140,
``load_precomputed_atmosphere_luts`` bake.rs:674-769) and the typed hand-off ``AtmosphereLutHandle``
(src/core/atmosphere/runtime.rs:38-89) that ``hybrid_render_terrain_reference(atmosphere=...)`` consumes
(src/py_functions/path_tracing/terrain_reference.rs:45-219).

Where the tables come from.  The reference compiles its bank (5 x 598 032 bytes, ``turbidity-{1,2,4,8,10}.bin``)
into the extension module.  This package installs the same five data files beside itself
(``forge3d_amd/data/aether_bank/``, round 5: a drop-in has to answer ``load_shipped`` with the shipped tables on a box
that holds no forge3d checkout): ``load_shipped`` reads the bank from ``bank_dir=``, ``$FORGE3D_AETHER_LUT_DIR``,
``$FORGE3D_REPO_ROOT/src/core/atmosphere/precomputed`` (a forge3d checkout) or that installed directory, in this order,
verifies every anchor it touches against the reference's locked SHA-256 (precomputed.rs:36-43) and
applies the reference's bracket interpolation.  When NO bank directory is found (the data directory was stripped from an
installation, or ``$FORGE3D_AETHER_NO_INSTALLED_BANK=1`` -- the tests' switch) the anchors are baked on the GPU by
this package's own baker (``atmosphere_bake_luts``, csrc/f3d_aether_bake.hip) -- which reproduces the shipped anchors
to the last f16 bit in 99.99 % of the values and within 1 f16 ulp in the rest (tests/test_aether_bake.py) -- and that
provenance is VISIBLE: the handle says ``precomputed=False`` / ``provenance="baked"`` and a ``RuntimeWarning`` is
emitted once per process.  ``load_shipped(..., require_bank=True)`` (or ``$FORGE3D_AETHER_REQUIRE_BANK=1``) turns the
missing bank into the error the reference's "no nearby or default LUT was substituted" rule asks for.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import math
import os
from dataclasses import dataclass, field, replace
from pathlib import Path

import numpy as np

TURBIDITY_BANK = (1.0, 2.0, 4.0, 8.0, 10.0)
INSTALLED_BANK = Path(__file__).resolve().parent / "data" / "aether_bank"  # the reference's five anchors (data), SHA-checked on load
ANCHOR_SHA256 = {  # precomputed.rs:36-43
    1.0: "9ead28087343283942d0bf834aecfb7b3a7b0ea513b830731c2cf9bc77a15f0b",
    2.0: "c6a77bd25241d6123078cace17d9a2181b520c44ac0e871274d755e092e565bc",
    4.0: "350a1d13863ac0f4a38a3be585e663a8e8c701c14cb5484760cb0d5ccbe772cd",
    8.0: "56594423699db4a644650e21f231824f19cb52c0abd718c88b7e4f21f00759cf",
    10.0: "633b77f0a6d8c31a4640e666ba45c7068fa31b7f623b1f711e5117583c1f51f5",
}
WAVELENGTHS_NM = (380.0, 420.0, 460.0, 500.0, 540.0, 580.0, 620.0, 660.0, 700.0, 740.0, 780.0)


@dataclass(frozen=True)
class LutDimensions:
    """bake.rs:30-58 (default = the shipped bank's, precomputed.rs:6-11)"""
    transmittance_mu: int = 32
    transmittance_height: int = 8
    scattering_mu_view: int = 17
    scattering_mu_sun: int = 17
    scattering_height: int = 8
    scattering_nu: int = 16
    aerial_distance: int = 8
    aerial_mu_view: int = 8
    aerial_height: int = 8

    def texel_counts(self):
        s = self.scattering_mu_view * self.scattering_mu_sun * self.scattering_height * self.scattering_nu
        return (self.transmittance_mu * self.transmittance_height, s, s,
                self.aerial_distance * self.aerial_mu_view * self.aerial_height)


@dataclass(frozen=True)
class AtmosphereConfig:
    """bake.rs:131-162"""
    turbidity: float = 2.0
    ozone_du: float = 300.0
    mie_g: float = 0.8
    bottom_radius_m: float = 6_360_000.0
    top_radius_m: float = 6_460_000.0
    rayleigh_scale_height_m: float = 8_000.0
    mie_scale_height_m: float = 1_200.0
    max_aerial_distance_m: float = 160_000.0
    ground_albedo: float = 0.3
    scattering_orders: int = 4
    dimensions: LutDimensions = field(default_factory=LutDimensions)

    def problem(self) -> "str | None":
        """AtmosphereConfig::validate, bake.rs:164-229: the first violated rule."""
        scalars = (self.turbidity, self.ozone_du, self.mie_g, self.bottom_radius_m, self.top_radius_m,
                   self.rayleigh_scale_height_m, self.mie_scale_height_m, self.max_aerial_distance_m, self.ground_albedo)
        if not all(math.isfinite(v) for v in scalars):
            return "all scalar parameters must be finite"
        if not 1.0 <= self.turbidity <= 10.0:
            return "turbidity must be in [1, 10]"
        if not 0.0 <= self.ozone_du <= 600.0:
            return "ozone must be in [0, 600] DU"
        if not 0.0 <= self.mie_g <= 0.99:
            return "mie_g must be in [0, 0.99]"
        if self.bottom_radius_m <= 0.0 or self.top_radius_m <= self.bottom_radius_m:
            return "top radius must exceed a positive bottom radius"
        if min(self.rayleigh_scale_height_m, self.mie_scale_height_m, self.max_aerial_distance_m) <= 0.0:
            return "scale heights and aerial distance must be positive"
        if not 0.0 <= self.ground_albedo <= 1.0:
            return "ground albedo must be in [0, 1]"
        if not 2 <= int(self.scattering_orders) <= 8:
            return "scattering_orders must be in [2, 8]"
        return None


def _f32(v) -> np.float32:
    return np.float32(v)


def _same_bits(a, b) -> bool:
    return _f32(a).tobytes() == _f32(b).tobytes()


def find_bank(bank_dir=None) -> Path:
    """Directory holding the reference's turbidity-*.bin anchors."""
    candidates = []
    if bank_dir:
        candidates.append(Path(bank_dir))
    if os.environ.get("FORGE3D_AETHER_LUT_DIR"):
        candidates.append(Path(os.environ["FORGE3D_AETHER_LUT_DIR"]))
    if os.environ.get("FORGE3D_REPO_ROOT"):
        candidates.append(Path(os.environ["FORGE3D_REPO_ROOT"]) / "src" / "core" / "atmosphere" / "precomputed")
    if os.environ.get("FORGE3D_AETHER_NO_INSTALLED_BANK", "") in ("", "0"):
        candidates.append(INSTALLED_BANK)
    for c in candidates:
        if c.is_dir() and any(c.glob("turbidity-*.bin")):
            return c
    raise FileNotFoundError(
        "no AETHER LUT bank found (looked in bank_dir, $FORGE3D_AETHER_LUT_DIR, $FORGE3D_REPO_ROOT/src/core/atmosphere/"
        "precomputed and the package's data/aether_bank)")


def _read_anchor(bank: Path, turbidity: float, dims: LutDimensions):
    name = bank / f"turbidity-{int(turbidity)}.bin"
    counts = dims.texel_counts()
    expected = sum(counts) * 8 + 4 * 4  # four RGBA16F tables + four f32 order deltas (precomputed.rs:13-25)
    raw = name.read_bytes()
    if len(raw) != expected:
        raise ValueError(f"{name}: {len(raw)} bytes, the shipped anchor layout has {expected}")
    digest = hashlib.sha256(raw).hexdigest()
    if digest != ANCHOR_SHA256[turbidity]:
        raise ValueError(f"{name}: SHA-256 {digest} is not the reference's locked anchor {ANCHOR_SHA256[turbidity]}")
    tables, off = [], 0
    for n in counts:
        tables.append(np.frombuffer(raw, dtype="<u2", count=n * 4, offset=off).copy())
        off += n * 8
    deltas = np.frombuffer(raw, dtype="<f4", count=4, offset=off).copy()
    return tables, deltas


@dataclass
class AtmosphereLutHandle:
    """Immutable LUT payload + the physical configuration it was baked for (runtime.rs:38-89).
    Tables are RGBA16F bit patterns (uint16), x fastest."""
    config: AtmosphereConfig
    transmittance: np.ndarray
    single_scattering: np.ndarray
    accumulated_scattering: np.ndarray
    aerial_perspective: np.ndarray
    order_deltas: np.ndarray
    precomputed: bool = True
    precomputed_turbidity_bracket: "tuple | None" = None
    provenance: str = "shipped"  # "shipped": the reference's SHA-locked anchors; "baked": this package's GPU baker

    @classmethod
    def load_shipped(cls, config: "AtmosphereConfig | None" = None, bank_dir=None, require_bank: "bool | None" = None) -> "AtmosphereLutHandle":
        """load_precomputed_atmosphere_luts, bake.rs:690-769: only the turbidity may differ from the shipped
        physical inputs; between anchors every f16 texel is interpolated in f32 and rounded back to f16.
        Without a bank directory the anchors are baked on the GPU and the handle says so (module docstring);
        require_bank=True raises FileNotFoundError instead."""
        config = config or AtmosphereConfig()
        problem = config.problem()
        if problem:
            raise ValueError(problem)
        defaults = AtmosphereConfig()
        if config.dimensions != LutDimensions():
            raise RuntimeError(f"precomputed atmosphere bank does not support dimensions={config.dimensions}; shipped dimensions are "
                               f"{LutDimensions()}")
        for name in ("ozone_du", "mie_g", "bottom_radius_m", "top_radius_m", "rayleigh_scale_height_m",
                     "mie_scale_height_m", "max_aerial_distance_m", "ground_albedo"):
            if not _same_bits(getattr(config, name), getattr(defaults, name)):
                raise RuntimeError(f"precomputed atmosphere bank does not support {name}={getattr(config, name)}; shipped value is "
                                   f"{getattr(defaults, name)}")
        if int(config.scattering_orders) != 4:
            raise RuntimeError(f"precomputed atmosphere bank does not support scattering_orders={config.scattering_orders}; shipped value is 4")
        t = _f32(config.turbidity)
        lower = upper = 4
        factor = _f32(0.0)
        for i in range(4):  # precomputed_bracket, bake.rs:674-688
            a, b = _f32(TURBIDITY_BANK[i]), _f32(TURBIDITY_BANK[i + 1])
            if t <= b:
                lower, upper, factor = i, i + 1, (t - a) / (b - a)
                break
        if require_bank is None:
            require_bank = os.environ.get("FORGE3D_AETHER_REQUIRE_BANK", "") not in ("", "0")
        try:
            bank = find_bank(bank_dir)
        except FileNotFoundError:
            if require_bank:
                raise
            bank = None  # the anchors are baked on the GPU instead (_anchor); the handle and a warning say so
            _warn_baked_once()
        if lower == upper or factor <= 0.0:
            pick = lower
        elif factor >= 1.0:
            pick = upper
        else:
            pick = None
        if pick is not None:
            tables, deltas = _anchor(bank, TURBIDITY_BANK[pick], config.dimensions)
        else:
            (ta, da), (tb, db) = (_anchor(bank, TURBIDITY_BANK[k], config.dimensions) for k in (lower, upper))
            tables = []
            for a, b in zip(ta, tb):
                fa, fb = a.view(np.float16).astype(np.float32), b.view(np.float16).astype(np.float32)
                tables.append((fa + (fb - fa) * factor).astype(np.float16).view(np.uint16))
            deltas = (da + (db - da) * factor).astype(np.float32)
        return cls(config, tables[0], tables[1], tables[2], tables[3], deltas, bank is not None,
                   (TURBIDITY_BANK[lower], TURBIDITY_BANK[upper]), "shipped" if bank is not None else "baked")

    def deterministic_sha256_hex(self) -> str:
        """A stable key of payload + configuration (the reference hashes its own serialisation; this key is this
        package's, used for cache lookups only)."""
        h = hashlib.sha256()
        for arr in (self.transmittance, self.single_scattering, self.accumulated_scattering, self.aerial_perspective):
            h.update(np.ascontiguousarray(arr, "<u2").tobytes())
        h.update(np.ascontiguousarray(self.order_deltas, "<f4").tobytes())
        h.update(repr(self.config).encode())
        return h.hexdigest()

    def byte_size(self) -> int:
        return 2 * (self.transmittance.size + self.single_scattering.size + self.accumulated_scattering.size
                    + self.aerial_perspective.size)


_BAKED_ANCHORS: dict = {}
_WARNED_BAKED = False


def _warn_baked_once() -> None:
    global _WARNED_BAKED
    if not _WARNED_BAKED:
        _WARNED_BAKED = True
        import warnings

        warnings.warn("forge3d_amd.atmosphere: no AETHER LUT bank directory found (bank_dir, $FORGE3D_AETHER_LUT_DIR, "
                      "$FORGE3D_REPO_ROOT, the package's data/aether_bank): the turbidity anchors are baked on the GPU by this package's baker instead of read "
                      "from the reference's SHA-locked files (<= 1 f16 ulp apart; handle.provenance == 'baked'). "
                      "Set FORGE3D_AETHER_REQUIRE_BANK=1 to make this an error.", RuntimeWarning, stacklevel=3)


def _anchor(bank, turbidity: float, dims: LutDimensions):
    """The anchor of the bank at `turbidity`: the reference's shipped file when a bank directory is known (verified
    against its locked SHA-256), otherwise the same tables baked on the GPU (bake_atmosphere_luts of the default
    configuration at that turbidity -- what the shipped files are; the bake agrees with them to the last f16 bit in all
    but a handful of values, tests/test_aether_bake.py) and kept for the process."""
    if bank is not None:
        return _read_anchor(bank, turbidity, dims)
    if turbidity not in _BAKED_ANCHORS:
        h = atmosphere_bake_luts(AtmosphereConfig(turbidity=float(turbidity), dimensions=dims))
        _BAKED_ANCHORS[turbidity] = ([h.transmittance, h.single_scattering, h.accumulated_scattering, h.aerial_perspective], h.order_deltas)
    return _BAKED_ANCHORS[turbidity]


def atmosphere_bake_luts(config: "AtmosphereConfig | None" = None, **settings) -> "AtmosphereLutHandle":
    """bake_atmosphere_luts (reference src/core/atmosphere/bake.rs:1481-1666, Python `atmosphere_bake_luts` of an
    atmosphere-bake build): the AETHER tables of ANY valid configuration, baked on the GPU (csrc/f3d_aether_bake.hip,
    ~0.1 s for the default dimensions; minutes of single-thread host code in the reference).  Keyword settings override
    fields of `config` (default AtmosphereConfig())."""
    import ctypes as C

    from . import _native

    config = replace(config or AtmosphereConfig(), **settings)
    problem = config.problem()
    if problem:
        raise ValueError(f"invalid atmosphere configuration: {problem}")
    d = config.dimensions

    class _Cfg(C.Structure):
        _fields_ = [(n, C.c_float) for n in ("turbidity", "ozone_du", "mie_g", "bottom_radius_m", "top_radius_m", "rayleigh_scale_height_m",
                                             "mie_scale_height_m", "max_aerial_distance_m", "ground_albedo")] + \
                   [("scattering_orders", C.c_uint32)] + \
                   [(n, C.c_uint32) for n in ("transmittance_mu", "transmittance_height", "scattering_mu_view", "scattering_mu_sun",
                                              "scattering_height", "scattering_nu", "aerial_distance", "aerial_mu_view", "aerial_height")]

    cfg = _Cfg()
    for name, _t in _Cfg._fields_:
        setattr(cfg, name, getattr(config, name) if hasattr(config, name) else getattr(d, name))
    counts = d.texel_counts()
    tables = [np.zeros(n * 4, np.uint16) for n in counts]
    deltas = np.zeros(int(config.scattering_orders), np.float32)
    seconds = C.c_double(0.0)
    err = C.create_string_buffer(512)
    rc = _native.lib().f3d_aether_bake(C.byref(cfg), tables[0].ctypes.data, tables[1].ctypes.data, tables[2].ctypes.data,
                                      tables[3].ctypes.data, deltas.ctypes.data, C.byref(seconds), err, len(err))
    if rc != 0:
        _native.raise_status(rc, err.value.decode(errors="replace"))
    handle = AtmosphereLutHandle(config, tables[0], tables[1], tables[2], tables[3], deltas, False, None)
    handle.bake_seconds = seconds.value  # device time of the bake kernels
    return handle


_SETTING_KEYS = ("enabled", "lut_handle", "turbidity", "ozone_du", "mie_g", "ground_albedo", "scattering_orders")


def resolve_setting(atmosphere, bank_dir=None) -> "AtmosphereLutHandle | None":
    """``atmosphere=`` of hybrid_render_terrain_reference -> handle or None
    (extract_atmosphere_lut_handle, terrain_reference.rs:45-219: same acceptance rules, messages and exception types)."""
    from collections.abc import Mapping

    if atmosphere is None:
        return None
    if isinstance(atmosphere, AtmosphereLutHandle):
        return atmosphere
    is_mapping = isinstance(atmosphere, Mapping)
    if is_mapping:
        for key in atmosphere.keys():
            if not isinstance(key, str):
                raise TypeError("atmosphere mapping keys must be strings")
            if key not in _SETTING_KEYS:
                raise ValueError(f'unknown atmosphere setting "{key}"; expected one of {", ".join(_SETTING_KEYS)}')
        item = atmosphere.get
    else:
        def item(name):
            return getattr(atmosphere, name, None)
        if all(not hasattr(atmosphere, k) for k in _SETTING_KEYS):
            raise TypeError("atmosphere must be an AtmosphereLutHandle, a mapping, or an object with recognized AETHER settings")
    if item("enabled") is False:
        return None
    handle = item("lut_handle")
    if handle is not None:
        if not isinstance(handle, AtmosphereLutHandle):
            raise TypeError("atmosphere.lut_handle must be an AtmosphereLutHandle returned by atmosphere_bake_luts()")
        for name in ("turbidity", "ozone_du", "mie_g", "ground_albedo"):
            supplied = item(name)
            if supplied is not None and not _same_bits(supplied, getattr(handle.config, name)):
                raise ValueError(f"atmosphere.{name}={float(_f32(supplied))} does not match the exact LUT handle value "
                                 f"{float(_f32(getattr(handle.config, name)))}; refusing to substitute or relabel transport")
        orders = item("scattering_orders")
        if orders is not None and int(orders) != int(handle.config.scattering_orders):
            raise ValueError(f"atmosphere.scattering_orders={int(orders)} does not match the exact LUT handle value "
                             f"{handle.config.scattering_orders}; refusing to substitute or relabel transport")
        return handle
    overrides = {k: (int(item(k)) if k == "scattering_orders" else float(item(k)))
                 for k in ("turbidity", "ozone_du", "mie_g", "ground_albedo", "scattering_orders") if item(k) is not None}
    config = replace(AtmosphereConfig(), **overrides)
    problem = config.problem()
    if problem:
        raise ValueError(f"invalid AETHER settings: invalid atmosphere configuration: {problem}")
    try:
        return AtmosphereLutHandle.load_shipped(config, bank_dir)
    except (RuntimeError, FileNotFoundError, ValueError) as error:
        raise RuntimeError(
            f"PROMETHEUS AETHER could not resolve the shipped LUT bank: {error}. Custom physical inputs require "
            "lut_handle=atmosphere_bake_luts(...) from an atmosphere-bake build; no nearby or default LUT was substituted.")


# ---- the stochastic spectral acceptance reference ---------------------------------------------------------------------
class _RefDesc(C.Structure):
    """f3d_aether_ref_desc"""
    _fields_ = [("struct_size", C.c_uint32), ("dem_width", C.c_uint32), ("dem_height", C.c_uint32), ("heights", C.c_void_p), ("spacing_x", C.c_float),
                ("spacing_z", C.c_float), ("exaggeration", C.c_float), ("cam_origin", C.c_float * 3), ("cam_look_at", C.c_float * 3),
                ("cam_up", C.c_float * 3), ("fov_y_deg", C.c_float), ("sun_azimuth_deg", C.c_float), ("sun_elevation_deg", C.c_float),
                ("sun_intensity", C.c_float), ("turbidity", C.c_float), ("ozone_du", C.c_float), ("mie_g", C.c_float), ("ground_albedo", C.c_float),
                ("width", C.c_uint32), ("height", C.c_uint32), ("seed", C.c_uint32), ("spp", C.c_uint32), ("enabled", C.c_int32),
                ("variance_threshold", C.c_float)]


class _RefOut(C.Structure):
    """f3d_aether_ref_out"""
    _fields_ = [("mean_xyz", C.c_void_p), ("linear_rgb", C.c_void_p), ("variance", C.c_float), ("converged", C.c_int32),
                ("terrain_primary_hits", C.c_uint64), ("gpu_resource_bytes", C.c_uint64), ("kernel_seconds", C.c_double)]


def hybrid_render_aether_spectral_reference(heightmap, width, height, cam, spacing=(1.0, 1.0), exaggeration=1.0, sun_azimuth_deg=90.0,
                                            sun_elevation_deg=10.0, sun_intensity=20.0, turbidity=2.0, ozone_du=300.0, mie_g=0.8,
                                            ground_albedo=0.3, spp=64, seed=7, enabled=True, variance_threshold=1e-3, certificate=None,
                                            cache=None) -> dict:
    """The independent GPU spectral atmosphere reference over a real DEM -- the acceptance check of the LUT post pass
    (reference `_forge3d.hybrid_render_aether_spectral_reference`, src/py_functions/path_tracing/aether_reference.rs:
    same signature, defaults, camera keys `origin` / `look_at` / `up` / `fov_y`, and result keys).  `mean_xyz` is the
    unclipped per-pixel estimator, `linear_rgb` its untonemapped non-negative form, `variance` the largest per-pixel
    estimated variance of the sample-mean luminance.  `certificate` and `cache` are accepted for signature parity (this
    build emits no render certificates; the reference is always recomputed); `kernel_seconds` is an extra key."""
    from . import _native

    del certificate, cache
    if not isinstance(cam, dict):
        raise TypeError("cam must be a dict")
    dem = np.ascontiguousarray(heightmap, dtype=np.float32)
    if dem.ndim != 2:
        raise TypeError("heightmap must be a 2-D float32 array")
    d = _RefDesc()
    d.struct_size = C.sizeof(_RefDesc)
    d.dem_height, d.dem_width = dem.shape
    d.heights = dem.ctypes.data
    d.spacing_x, d.spacing_z, d.exaggeration = float(spacing[0]), float(spacing[1]), float(exaggeration)
    d.cam_origin = (C.c_float * 3)(*cam.get("origin", (0.0, 1.0, 0.0)))
    d.cam_look_at = (C.c_float * 3)(*cam.get("look_at", (1.0, 1.0, 0.0)))
    d.cam_up = (C.c_float * 3)(*cam.get("up", (0.0, 1.0, 0.0)))
    d.fov_y_deg = float(cam.get("fov_y", 20.0))
    d.sun_azimuth_deg, d.sun_elevation_deg, d.sun_intensity = float(sun_azimuth_deg), float(sun_elevation_deg), float(sun_intensity)
    d.turbidity, d.ozone_du, d.mie_g, d.ground_albedo = float(turbidity), float(ozone_du), float(mie_g), float(ground_albedo)
    d.width, d.height, d.seed, d.spp = int(width), int(height), int(seed) & 0xFFFFFFFF, int(spp)
    d.enabled, d.variance_threshold = 1 if enabled else 0, float(variance_threshold)
    mean_xyz = np.zeros((max(0, int(height)), max(0, int(width)), 3), np.float32)
    rgb = np.zeros_like(mean_xyz)
    out = _RefOut()
    out.mean_xyz, out.linear_rgb = mean_xyz.ctypes.data, rgb.ctypes.data
    err = C.create_string_buffer(512)
    rc = _native.lib().f3d_aether_reference_render(C.byref(d), C.byref(out), err, len(err))
    if rc != 0:
        _native.raise_status(rc, err.value.decode("utf-8", "replace"))
    return {"mean_xyz": mean_xyz, "linear_rgb": rgb, "variance": float(out.variance), "converged": bool(out.converged), "seed": int(d.seed),
            "spp": int(d.spp), "terrain_primary_hits": int(out.terrain_primary_hits), "gpu_resource_bytes": int(out.gpu_resource_bytes),
            "environment": "black", "wavelength_count": 11, "max_depth": 6, "kernel_seconds": float(out.kernel_seconds)}
