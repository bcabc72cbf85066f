"""Generates tests/golden/smoke/composite_vectors.npz: inputs and the outputs of the REFERENCE's own smoke composites.

Run in the build container only (needs /root/reference, numpy 2.2 and Pillow 12.2.0); the .npz is the committed
fixture, this script is how it was made:
    python tests/golden/make_composite_vectors.py
The functions are imported from /root/reference/examples/california_cigar_smoke_demo.py the way the reference's own
tests/test_california_cigar_smoke_hybrid.py:25-40 loads that module.  Inputs: seeded random RGBA8 images (alpha in
bands, so that zero, thin, mid and dense smoke all occur) plus the literal images of the reference's three composite
tests (:235-283).
"""
import importlib.util
import sys
from pathlib import Path

import numpy as np

EXAMPLE = Path("/root/reference/examples/california_cigar_smoke_demo.py")
OUT = Path(__file__).resolve().parent / "smoke" / "composite_vectors.npz"


def load_module():
    sys.path.insert(0, str(EXAMPLE.parent))
    spec = importlib.util.spec_from_file_location("california_cigar_smoke_demo", EXAMPLE)
    module = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = module
    spec.loader.exec_module(module)
    return module


def random_rgba(rng, h, w):
    img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    band = rng.integers(0, 5, (h, w))
    a = img[..., 3]
    a[band == 0] = 0
    a[band == 1] = a[band == 1] % 25
    a[band == 2] = 255
    return img


def main():
    m = load_module()
    Image = m.Image
    rng = np.random.default_rng(20260927)
    h, w = 96, 128
    v = {}
    # composite_atmospheric_smoke: terrain-like bases (warm and cold), random smoke layers
    base = random_rgba(rng, h, w)
    base[..., 3] = 255
    base[: h // 2, :, 2] = base[: h // 2, :, 2] // 3  # warm half: drives the source_transmission / glow terms
    smoke = random_rgba(rng, h, w)
    v["atm_base"], v["atm_smoke"] = base, smoke
    v["atm_out"] = np.asarray(m.composite_atmospheric_smoke(Image.fromarray(base, "RGBA"), Image.fromarray(smoke, "RGBA")), dtype=np.uint8)
    # every (alpha, grey) pair once: the whole domain of optical depth against a neutral terrain
    aa, gg = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    sweep_smoke = np.stack([np.full_like(aa, 214), np.full_like(aa, 218), np.full_like(aa, 214), aa], axis=-1)
    sweep_base = np.stack([gg, gg, gg, np.full_like(gg, 255)], axis=-1)
    v["atm_sweep_base"], v["atm_sweep_smoke"] = sweep_base, sweep_smoke
    v["atm_sweep_out"] = np.asarray(m.composite_atmospheric_smoke(Image.fromarray(sweep_base, "RGBA"), Image.fromarray(sweep_smoke, "RGBA")), dtype=np.uint8)
    # reference test :235-243
    tb = np.empty((24, 24, 4), np.uint8)
    tb[:] = (34, 37, 38, 255)
    ts = np.empty((24, 24, 4), np.uint8)
    ts[:] = (214, 218, 214, 104)
    v["atm_test_base"], v["atm_test_smoke"] = tb, ts
    v["atm_test_out"] = np.asarray(m.composite_atmospheric_smoke(Image.fromarray(tb, "RGBA"), Image.fromarray(ts, "RGBA")), dtype=np.uint8)

    # composite_main_smoke_maps
    atmos, phys = random_rgba(rng, h, w), random_rgba(rng, h, w)
    v["maps_atmospheric"], v["maps_physical"] = atmos, phys
    v["maps_out"] = m.composite_main_smoke_maps(atmos, phys)
    v["maps_out_none"] = m.composite_main_smoke_maps(atmos, None)
    v["maps_out_scaled"] = m.composite_main_smoke_maps(atmos, phys, atmospheric_alpha=0.68, physical_alpha=0.58)
    ta = np.zeros((42, 64, 4), np.uint8)
    tp = np.zeros_like(ta)
    ta[10:34, 6:58, :3] = (190, 195, 190)
    ta[10:34, 6:58, 3] = 72
    tp[18:27, 18:44, :3] = (222, 220, 208)
    tp[18:27, 18:44, 3] = 132
    v["maps_test_atmospheric"], v["maps_test_physical"] = ta, tp  # reference test :267-283
    v["maps_test_out"] = m.composite_main_smoke_maps(ta, tp)
    v["max_alpha"] = np.int64(m.HYBRID_SMOKE_MAX_ALPHA)

    # PIL.Image.alpha_composite, full frame and a smaller layer placed at offsets (in-place form, dest = offset)
    dst, src = random_rgba(rng, h, w), random_rgba(rng, h, w)
    v["over_base"], v["over_layer"] = dst, src
    v["over_out"] = np.asarray(Image.alpha_composite(Image.fromarray(dst, "RGBA"), Image.fromarray(src, "RGBA")), dtype=np.uint8)
    small = random_rgba(rng, 40, 56)
    v["over_small"] = small
    for name, off in (("a", (0, 0)), ("b", (17, 9)), ("c", (72, 56))):
        canvas = Image.fromarray(dst, "RGBA")
        canvas.alpha_composite(Image.fromarray(small, "RGBA"), off)
        v[f"over_small_out_{name}"] = np.asarray(canvas, dtype=np.uint8)
        v[f"over_small_offset_{name}"] = np.asarray(off, dtype=np.int64)
    OUT.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT, **v)
    print(OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
