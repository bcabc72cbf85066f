"""I/O helpers and the ViewerHandle-shaped facade (parity unpinned by the
reference: its snapshot() is a raster viewer).  CPU: PNG round trips incl. the reference's own golden
PNG (written by another encoder, other filter types); GPU: snapshot() == direct call."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def test_png_round_trip_and_foreign_file(tmp_path):
    from forge3d_amd import io

    rng = np.random.default_rng(3)
    for shape in ((17, 23), (17, 23, 3), (9, 31, 4)):
        a = rng.integers(0, 256, shape, dtype=np.uint8)
        io.numpy_to_png(tmp_path / "a.png", a)
        assert np.array_equal(io.png_to_numpy(tmp_path / "a.png"), a)
    import scenes

    assert np.array_equal(io.png_to_numpy(scenes.GOLDEN_DIR / "mini_dem_reference.png"), scenes.golden_png())
    with pytest.raises(ValueError, match="uint8"):
        io.numpy_to_png(tmp_path / "b.png", np.zeros((4, 4), np.float32))
    np.save(tmp_path / "d.npy", np.ones((5, 7), np.float64))
    assert io.load_heightmap(tmp_path / "d.npy").dtype == np.float32
    with pytest.raises(ValueError, match="unsupported heightmap"):
        io.load_heightmap(tmp_path / "d.tif")


def test_orbit_mapping_of_the_facade():
    from forge3d_amd.offline import OfflineTerrainViewer

    v = OfflineTerrainViewer(64, 48)
    v.load_terrain(np.zeros((8, 8), np.float32), spacing=10.0)
    v.set_orbit_camera(0.0, 90.0, 100.0, fov_deg=30.0, target=(1.0, 2.0, 3.0))  # level with the horizon, along +x
    cam = v._camera
    assert np.allclose(cam["origin"], (101.0, 2.0, 3.0), atol=1e-9) and cam["look_at"] == (1.0, 2.0, 3.0)
    v.set_orbit_camera(90.0, 0.0, 50.0, target=(0.0, 0.0, 0.0))  # straight down
    assert np.allclose(v._camera["origin"], (0.0, 50.0, 0.0), atol=1e-9)
    with pytest.raises(RuntimeError, match="no terrain"):
        OfflineTerrainViewer().render()


@pytest.mark.gpu
def test_snapshot_writes_the_path_traced_frame(tmp_path):
    import forge3d_amd as f3d
    import scenes
    from forge3d_amd import io
    from forge3d_amd.offline import OfflineTerrainViewer

    dem = scenes.golden_dem()
    kw = scenes.scene_kwargs(dem)
    v = OfflineTerrainViewer(128, 96, spp=2, max_frames=4, min_frames=4, variance_threshold=1e30)
    v.load_terrain(dem, spacing=kw["spacing"])
    v.set_z_scale(kw["exaggeration"])
    v.set_sun(kw["sun_azimuth_deg"], kw["sun_elevation_deg"])
    v.set_fov(scenes.CAM["fov_y"])
    v.set_camera_lookat(scenes.CAM["origin"], scenes.CAM["look_at"], scenes.CAM["up"])
    v.snapshot(tmp_path / "snap.png")
    want = f3d.hybrid_render_terrain_reference(dem, 128, 96, scenes.CAM, spacing=kw["spacing"],
                                               exaggeration=kw["exaggeration"], sun_azimuth_deg=kw["sun_azimuth_deg"],
                                               sun_elevation_deg=kw["sun_elevation_deg"], spp=2, max_frames=4,
                                               min_frames=4, variance_threshold=1e30)
    assert np.array_equal(io.png_to_numpy(tmp_path / "snap.png"), want["rgba"])
