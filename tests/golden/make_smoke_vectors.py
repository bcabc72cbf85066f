"""Generates tests/golden/smoke/marcher_vectors.npz: seeded inputs and the outputs of the REFERENCE's own NumPy smoke
helpers (python/forge3d/smoke.py:734-941) -- _smoke_sample_volume, _smoke_ray_box_intersection, _smoke_henyey_greenstein,
_smoke_smoothstep, _smoke_light_transmittance.

Run in the build container only (needs /root/reference and numpy); the .npz is the committed fixture (data: inputs and
expected outputs), this script is how it was made:
    python tests/golden/make_smoke_vectors.py
The reference's Rust marcher (src/smoke/render.rs, what oracle/smoke_oracle.c restates and csrc/f3d_smoke.hip runs) cannot
be executed here; these helpers are the reference's second statement of the marcher's primitives (its NumPy example
renderer), importable in pure-Python mode.  Where the two statements differ BY DESIGN the vectors stay inside the common
domain: the NumPy sampler returns 0 outside the grid where the Rust one clamps, so sample points are in bounds; the NumPy
ray-box treats |d| < 1e-6 as parallel where the Rust one uses 1e-12, so directions are either well away from both or
exactly 0.  round-4 verdict, "Next round" item 8.
"""
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, "/root/reference/python")
OUT = Path(__file__).resolve().parent / "smoke" / "marcher_vectors.npz"


def main():
    from forge3d import smoke as ref

    rng = np.random.default_rng(20260928)
    v = {}
    # ---- trilinear sampling: (depth, height, width) fields, in-bounds points incl. exact lattice points and the far faces ----
    for tag, (d, h, w) in (("a", (7, 5, 9)), ("b", (16, 12, 20))):
        field = rng.random((d, h, w), dtype=np.float32) * np.float32(3.0)
        n = 4000
        x = (rng.random(n, dtype=np.float32) * np.float32(w - 1)).astype(np.float32)
        y = (rng.random(n, dtype=np.float32) * np.float32(h - 1)).astype(np.float32)
        z = (rng.random(n, dtype=np.float32) * np.float32(d - 1)).astype(np.float32)
        # lattice points, the last cell and the far faces themselves
        x[:200] = np.floor(x[:200]); y[100:300] = np.floor(y[100:300]); z[150:400] = np.floor(z[150:400])  # [150, 200): all three integral
        x[400:420] = w - 1; y[420:440] = h - 1; z[440:460] = d - 1
        x[460:470] = 0; y[470:480] = 0; z[480:490] = 0
        v[f"sample_{tag}_field"] = field
        v[f"sample_{tag}_xyz"] = np.stack([x, y, z], axis=1)
        v[f"sample_{tag}_out"] = ref._smoke_sample_volume(field, x, y, z)
    # ---- ray / box: origins around and inside [0, upper]^3, one direction per case (the helper takes a scalar direction) ----
    dirs = [(0.3, -0.8, 0.52), (-0.7, 0.1, 0.7), (0.0, 1.0, 0.0), (0.0, 0.0, -1.0), (0.6, 0.0, 0.8), (-0.5, -0.5, 0.70710678), (1.0, 0.0, 0.0), (0.05, 0.99, -0.1)]
    upper = (23.0, 11.0, 31.0)
    for k, dvec in enumerate(dirs):
        dvec = np.asarray(dvec, np.float32)
        n = 1500
        o = (rng.random((n, 3), dtype=np.float32) * np.float32(3.0) - np.float32(1.0)) * np.asarray(upper, np.float32)
        o[:50] = np.round(o[:50])  # origins on lattice planes and box faces
        o[50:80, 0] = 0.0
        o[80:110, 1] = upper[1]
        t_enter, t_exit, valid = ref._smoke_ray_box_intersection(o[:, 0].copy(), o[:, 1].copy(), o[:, 2].copy(), dvec, upper)
        v[f"box_{k}_dir"] = dvec
        v[f"box_{k}_origins"] = o
        v[f"box_{k}_enter"] = np.asarray(t_enter, np.float32)
        v[f"box_{k}_exit"] = np.asarray(t_exit, np.float32)
        v[f"box_{k}_valid"] = np.asarray(valid, np.bool_)
    v["box_upper"] = np.asarray(upper, np.float32)
    # ---- Henyey-Greenstein and smoothstep: scalars ----
    cg = np.stack([rng.uniform(-1.0, 1.0, 600), rng.uniform(-0.95, 0.95, 600)], axis=1).astype(np.float32)
    cg[:6] = [(1.0, 0.9), (-1.0, 0.9), (1.0, -0.9), (0.0, 0.0), (1.0, 0.24), (-1.0, 0.24)]
    v["hg_in"] = cg
    v["hg_out"] = np.asarray([ref._smoke_henyey_greenstein(float(c), float(g)) for c, g in cg], np.float64)
    edges = [(1.6, 17.0), (0.045, 0.34), (0.0, 1.0), (2.0, 2.0)]
    xs = rng.uniform(-1.0, 20.0, 500).astype(np.float32)
    v["smoothstep_edges"] = np.asarray(edges, np.float32)
    v["smoothstep_x"] = xs
    v["smoothstep_out"] = np.stack([np.asarray(ref._smoke_smoothstep(e0, e1, xs), np.float64) for e0, e1 in edges])
    # ---- the sun march of the NumPy renderer: a composite of the sampler and exp() over a layer of start points ----
    d, h, w = 14, 10, 18
    yy = np.arange(h, dtype=np.float32)[None, :, None]
    density = (rng.random((d, h, w), dtype=np.float32) * np.exp(-yy / np.float32(4.0))).astype(np.float32)
    soot = (rng.random((d, h, w), dtype=np.float32) * np.float32(0.6)).astype(np.float32)
    sun = np.asarray([0.35, 0.85, -0.4], np.float64)
    sun = (sun / np.linalg.norm(sun)).astype(np.float32)
    for k, (layer, steps, step_size) in enumerate(((2, 12, 0.75), (6, 20, 0.0))):
        settings = SimpleNamespace(shadow_step_size=step_size, shadow_steps=steps, density_scale=1.3, extinction=2.6, soot_absorption=0.22)
        v[f"light_{k}_out"] = ref._smoke_light_transmittance(density, soot, layer, sun, settings)
        v[f"light_{k}_params"] = np.asarray([layer, steps, step_size, settings.density_scale, settings.extinction, settings.soot_absorption], np.float64)
    v["light_density"], v["light_soot"], v["light_sun"] = density, soot, sun
    OUT.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT, **v)
    print("wrote", OUT, OUT.stat().st_size, "bytes,", len(v), "arrays")


if __name__ == "__main__":
    main()
