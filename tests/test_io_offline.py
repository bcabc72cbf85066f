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
        io.load_heightmap(tmp_path / "d.asc")


def _write_tiff(path, array, *, big_endian=False, tile=None, compression=1, predictor=1, scale=None, nodata=None):
    """A tiny independent TIFF writer for the reader's tests: strips or tiles, none / Deflate, predictors 1-3."""
    import struct
    import zlib

    e = ">" if big_endian else "<"
    a = np.ascontiguousarray(array)
    h, w = a.shape
    code = {"f": 3, "i": 2, "u": 1}[a.dtype.kind]
    bps = a.dtype.itemsize

    def pack(block):
        block = np.ascontiguousarray(block).astype(a.dtype.newbyteorder(e))
        if predictor == 2:
            d = block.astype(block.dtype.newbyteorder("=")).copy()
            d[:, 1:] = d[:, 1:] - d[:, :-1]
            block = d.astype(a.dtype.newbyteorder(e))
        raw = block.tobytes()
        if predictor == 3:
            rows, cols = block.shape
            b = np.frombuffer(raw, np.uint8).reshape(rows, cols, bps)
            if e == "<":
                b = b[:, :, ::-1]
            b = b.transpose(0, 2, 1).reshape(rows, cols * bps).astype(np.int16)
            b[:, 1:] = b[:, 1:] - b[:, :-1]
            raw = (b & 255).astype(np.uint8).tobytes()
        return zlib.compress(raw) if compression == 8 else raw

    chunks = []
    if tile:
        th, tw = tile
        for y0 in range(0, h, th):
            for x0 in range(0, w, tw):
                blk = np.zeros((th, tw), a.dtype)
                part = a[y0:y0 + th, x0:x0 + tw]
                blk[:part.shape[0], :part.shape[1]] = part
                chunks.append(pack(blk))
    else:
        rps = 8
        for y0 in range(0, h, rps):
            chunks.append(pack(a[y0:y0 + rps]))
    entries, extra = [], b""
    data_start = 8
    offsets, pos = [], data_start
    for c in chunks:
        offsets.append(pos)
        pos += len(c) + (len(c) & 1)
    body = b"".join(c + (b"\0" if len(c) & 1 else b"") for c in chunks)
    n_tags = 10 + (2 if tile else 1) + (1 if scale else 0) + (1 if nodata is not None else 0)
    ifd_pos = pos
    extra_pos = ifd_pos + 2 + 12 * n_tags + 4

    def tag(t, typ, values):
        nonlocal extra, extra_pos
        fmt = {3: "H", 4: "I", 12: "d", 2: "s"}[typ]
        if typ == 2:
            payload = values.encode() + b"\0"
            n = len(payload)
        else:
            payload = struct.pack(e + fmt * len(values), *values)
            n = len(values)
        if len(payload) <= 4:
            field = payload.ljust(4, b"\0")
        else:
            field = struct.pack(e + "I", extra_pos)
            extra += payload + (b"\0" if len(payload) & 1 else b"")
            extra_pos += len(payload) + (len(payload) & 1)
        entries.append((t, struct.pack(e + "HHI", t, typ, n) + field))

    tag(256, 4, [w]); tag(257, 4, [h]); tag(258, 3, [bps * 8]); tag(259, 3, [compression]); tag(262, 3, [1])
    tag(277, 3, [1]); tag(317, 3, [predictor]); tag(339, 3, [code])
    if tile:
        tag(322, 4, [tile[1]]); tag(323, 4, [tile[0]]); tag(324, 4, offsets); tag(325, 4, [len(c) for c in chunks])
    else:
        tag(273, 4, offsets); tag(278, 4, [8]); tag(279, 4, [len(c) for c in chunks])
    if scale:
        tag(33550, 12, [scale[0], scale[1], 0.0])
    if nodata is not None:
        tag(42113, 2, str(nodata))
    entries.sort()
    ifd = struct.pack(e + "H", len(entries)) + b"".join(x[1] for x in entries) + struct.pack(e + "I", 0)
    header = (b"MM" if big_endian else b"II") + struct.pack(e + "HI", 42, ifd_pos)
    Path(path).write_bytes(header + body + ifd + extra)


@pytest.mark.parametrize("kw", [dict(), dict(big_endian=True), dict(tile=(16, 16)), dict(compression=8, predictor=3),
                                dict(compression=8, predictor=3, big_endian=True, tile=(16, 32)),
                                dict(scale=(30.0, 30.0), nodata=-9999.0)])
def test_geotiff_reader_decodes_float_dems(tmp_path, kw):
    from forge3d_amd import io

    rng = np.random.default_rng(2)
    dem = rng.normal(1500.0, 400.0, (45, 70)).astype(np.float32)
    if "nodata" in kw:
        dem[3:6, 10:14] = -9999.0
    _write_tiff(tmp_path / "d.tif", dem, **kw)
    got, info = io.read_geotiff(tmp_path / "d.tif")
    assert np.array_equal(got, dem)
    if "scale" in kw:
        assert info["pixel_scale"] == (30.0, 30.0) and info["nodata"] == -9999.0
        filled = io.load_heightmap(tmp_path / "d.tif")
        assert np.isfinite(filled).all() and filled.min() == dem[dem > -9999.0].min()


def test_geotiff_reader_decodes_integer_dems_and_foreign_files(tmp_path):
    from forge3d_amd import io

    rng = np.random.default_rng(4)
    dem = rng.integers(-200, 4000, (33, 41)).astype(np.int16)
    _write_tiff(tmp_path / "i.tif", dem, compression=8, predictor=2)
    assert np.array_equal(io.read_geotiff(tmp_path / "i.tif")[0], dem.astype(np.float32))
    PIL = pytest.importorskip("PIL.Image")
    f = rng.normal(900.0, 200.0, (37, 53)).astype(np.float32)
    for comp in (None, "tiff_lzw", "tiff_adobe_deflate", "packbits"):  # files of another encoder (libtiff)
        PIL.fromarray(f).save(tmp_path / "p.tif", compression=comp)
        assert np.array_equal(io.read_geotiff(tmp_path / "p.tif")[0], f), comp
    with pytest.raises(ValueError, match="not a TIFF"):
        (tmp_path / "x.tif").write_bytes(b"nope")
        io.read_geotiff(tmp_path / "x.tif")


def test_viewer_handle_and_renderer_names():
    import forge3d_amd as f3d

    h = f3d.open_viewer_async(320, 200, fov_deg=50.0)
    assert isinstance(h, f3d.ViewerHandle) and h.is_running and h.get_stats()["backend"] == "hip-gfx950"
    h.load_terrain(np.zeros((8, 8), np.float32), spacing=10.0)
    h.set_orbit_camera(0.0, 90.0, 100.0, target=(0.0, 0.0, 0.0))
    h.set_sun(302.0, 24.0)
    h.set_z_scale(2.0)
    with pytest.raises(f3d.viewer.ViewerError, match="raster viewer"):
        h.add_label("x", (0, 0, 0))
    with pytest.raises(f3d.viewer.ViewerError):
        f3d.open_viewer_async(obj_path="a.obj")
    with h:
        pass
    assert not h.is_running
    r = f3d.Renderer(64, 48, exposure=1.5)
    assert r.get_config()["lighting"]["exposure"] == 1.5 and r.render_triangle_rgba().shape == (48, 64, 4)
    assert tuple(r.render_triangle_rgba()[24, 32]) == (128, 64, 32, 255) and tuple(r.render_triangle_rgba()[0, 0]) == (16, 16, 24, 255)
    with pytest.raises(TypeError, match="Unexpected arguments: bogus"):
        f3d.Renderer(8, 8, bogus=1)


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


@pytest.mark.gpu
def test_viewer_handle_animation_reuses_the_cached_scene(tmp_path):
    """ViewerHandle over a GeoTIFF DEM: a three-frame orbit; the DEM's tables are built once (scene cache) and every
    frame equals the direct call."""
    import forge3d_amd as f3d
    import scenes
    from forge3d_amd import _native, io

    dem = (scenes.golden_dem() * 20.0).astype(np.float32)
    spacing = scenes.SPAN / (dem.shape[1] - 1)
    _write_tiff(tmp_path / "dem.tif", dem, compression=8, predictor=3, scale=(spacing, spacing))
    L = _native.lib()
    L.f3d_scene_cache_limit(0)
    L.f3d_scene_cache_limit(2)
    h = f3d.open_viewer_async(96, 64, terrain_path=tmp_path / "dem.tif", fov_deg=45.0)
    h._render.update(spp=2, max_frames=3, min_frames=3, variance_threshold=1e30)
    h.set_sun(225.0, 35.0)
    keys = [dict(phi_deg=p, theta_deg=60.0, radius=110.0, target=(0.0, 5.0, 0.0)) for p in (20.0, 50.0, 80.0)]
    h.render_animation(keys, tmp_path / "frames", width=96, height=64)
    assert L.f3d_scene_cache_entries() == 1
    from forge3d_amd.datasets import orbit_camera

    for i, k in enumerate(keys):
        cam = orbit_camera(k["target"], k["radius"], k["phi_deg"], k["theta_deg"], 45.0)
        want = f3d.hybrid_render_terrain_reference(dem, 96, 64, cam, spacing=(spacing, spacing), sun_azimuth_deg=225.0,
                                                   sun_elevation_deg=35.0, spp=2, max_frames=3, min_frames=3,
                                                   variance_threshold=1e30)
        assert np.array_equal(io.png_to_numpy(tmp_path / "frames" / f"frame_{i:04d}.png"), want["rgba"]), i
    L.f3d_scene_cache_limit(0)
    assert L.f3d_scene_cache_entries() == 0
    L.f3d_scene_cache_limit(2)


# ---- Radiance .hdr environment maps (reference src/formats/hdr.rs; its unit tests :290-378 restated) ----
def _hdr_file(tmp_path, body: bytes, header=b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n", res=b"-Y 2 +X 4\n", name="e.hdr"):
    p = tmp_path / name
    p.write_bytes(header + res + body)
    return p


def test_hdr_pixel_rule_and_flat_files(tmp_path):
    from forge3d_amd import io

    # load_hdr_uncompressed_round_trip_dims_and_values (:345-378): 4 x 2 pixels of (128, 64, 32, 129)
    img = io.read_hdr(_hdr_file(tmp_path, bytes([128, 64, 32, 129]) * 8))
    assert img.shape == (2, 4, 3) and img.dtype == np.float32
    exp = 2.0 ** (129 - 128 - 8)
    assert np.array_equal(img, np.broadcast_to(np.float32([128 * exp, 64 * exp, 32 * exp]), (2, 4, 3)))
    # test_rgbe_to_rgb_zero / _nonzero / _bright (:292-326)
    px = io.read_hdr(_hdr_file(tmp_path, bytes([10, 20, 30, 0, 128, 128, 128, 128, 255, 128, 64, 140]), res=b"-Y 1 +X 3\n"))
    assert np.array_equal(px[0, 0], [0.0, 0.0, 0.0])
    assert np.array_equal(px[0, 1], [0.5, 0.5, 0.5])
    assert np.array_equal(px[0, 2], [255 * 16.0, 128 * 16.0, 64 * 16.0])
    # `#?RGBE` magic and the xyze format name are accepted too; the resolution tokens are positional
    io.read_hdr(_hdr_file(tmp_path, bytes(4 * 8), header=b"#?RGBE\nEXPOSURE=1\nFORMAT=32-bit_rle_xyze\n\n", res=b"+Y 2 +X 4\n"))


def test_hdr_run_length_scanlines(tmp_path):
    from forge3d_amd import io

    w, h = 40, 3
    rng = np.random.default_rng(4)
    rgbe = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    rgbe[1, 5:30, 0] = 77  # a run worth encoding
    body = b""
    for y in range(h):
        if y == 2:  # a flat row between run-length rows (its first two bytes must not look like the marker)
            rgbe[y, 0, 0] = 9
            body += rgbe[y].tobytes()
            continue
        body += bytes([2, 2, w >> 8, w & 255])
        for c in range(4):
            col = rgbe[y, :, c]
            x = 0
            while x < w:
                run = 1
                while x + run < w and run < 127 and col[x + run] == col[x]:
                    run += 1
                if run >= 3:
                    body += bytes([128 + run, col[x]])
                else:
                    run = min(w - x, 7)
                    body += bytes([run]) + col[x:x + run].tobytes()
                x += run
    got = io.read_hdr(_hdr_file(tmp_path, body, res=f"-Y {h} +X {w}\n".encode()))
    scale = np.where(rgbe[..., 3] == 0, 0.0, np.exp2(rgbe[..., 3].astype(np.float64) - 136.0)).astype(np.float32)
    assert np.array_equal(got, rgbe[..., :3].astype(np.float32) * scale[..., None])


def test_hdr_errors_and_round_trip(tmp_path):
    from forge3d_amd import io

    for kw, message in ((dict(header=b"P6\n\n"), "missing magic header"), (dict(header=b"#?RADIANCE\n\n"), "missing FORMAT"),
                        (dict(header=b"#?RADIANCE\nFORMAT=32-bit_rle_foo\n\n"), "Unsupported HDR format"),
                        (dict(res=b"-Y 2\n"), "Invalid HDR resolution line"), (dict(res=b"-Y two +X 4\n"), "Invalid HDR height"),
                        (dict(res=b"-Y 0 +X 4\n"), "cannot be zero"), (dict(body=bytes(12)), "Failed to read pixel data at row 0"),
                        (dict(body=bytes([2, 2, 0, 4, 200, 1])), "RLE run exceeds scanline width")):
        body = kw.pop("body", bytes(32))
        with pytest.raises(io.HdrError, match=message):
            io.read_hdr(_hdr_file(tmp_path, body, **kw))
    rng = np.random.default_rng(2)
    env = (rng.random((6, 12, 3)) * np.float32(40.0)).astype(np.float32)
    env[0, 0] = 0.0
    io.write_hdr(tmp_path / "env.hdr", env)
    back = io.read_hdr(tmp_path / "env.hdr")
    assert np.all(np.abs(back - env) <= np.max(env, axis=2, keepdims=True) / 128.0 + 1e-30)  # 8-bit mantissa of the largest channel
    assert np.array_equal(back[0, 0], [0.0, 0.0, 0.0])


def test_viewer_set_ibl_reads_hdr_files(tmp_path):
    from forge3d_amd import io, viewer

    env = np.full((4, 8, 3), 0.5, np.float32)
    io.write_hdr(tmp_path / "sky.hdr", env)
    v = viewer.ViewerHandle.__new__(viewer.ViewerHandle)
    v._revision = 0
    v.set_ibl(tmp_path / "sky.hdr", intensity=2.0)
    assert np.array_equal(v._env, env) and v._env_intensity == 2.0
    with pytest.raises(viewer.ViewerError, match="Unsupported environment map format"):
        v.set_ibl(tmp_path / "sky.png")
    with pytest.raises(viewer.ViewerError):
        v.set_ibl(tmp_path / "missing.hdr")


def test_hdr_header_cannot_ask_for_more_than_the_file_holds(tmp_path):
    """dimensions parse as Rust's str::parse::<u32> does (reference src/formats/hdr.rs:137-143: an optional
    leading '+', decimal digits with any number of leading zeros, value < 2^32); a tiny file that promises a huge image is
    refused before anything is allocated for it, in the words of the reference's read_exact failure."""
    from forge3d_amd import io

    head = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n"
    for res in (b"-Y +2 +X 0000000000004\n", b"-Y 2 +X +4\n"):  # what u32::from_str accepts
        p = tmp_path / "plus.hdr"
        p.write_bytes(head + res + bytes([1, 2, 3, 128] * 8))
        assert io.read_hdr(p).shape == (2, 4, 3)
    for res in (b"-Y ++5 +X 4\n", b"-Y -0 +X 4\n", b"-Y + +X 4\n", b"-Y 1_0 +X 4\n", b"-Y 4 +X 4294967296\n", b"-Y 4 +X 00000000099999999999\n"):
        p = tmp_path / "bad.hdr"
        p.write_bytes(head + res + b"\0" * 64)
        with pytest.raises(io.HdrError, match="Invalid HDR"):
            io.read_hdr(p)
    p = tmp_path / "huge.hdr"
    p.write_bytes(head + b"-Y 60000 +X 60000\n" + b"\0" * 32)
    with pytest.raises(io.HdrError, match="failed to fill whole buffer"):
        io.read_hdr(p)


def test_real_dem_hook_of_the_bench(tmp_path, monkeypatch):
    """a real DEM supplied through FORGE3D_REPO_ROOT (or --dem-path) is rendered with the headline
    camera mapping and labelled; without one fetch_dem says where it looked (no download)."""
    from forge3d_amd import datasets

    monkeypatch.delenv("FORGE3D_REPO_ROOT", raising=False)
    with pytest.raises(FileNotFoundError, match="FORGE3D_REPO_ROOT"):
        datasets.fetch_dem("rainier")
    with pytest.raises(KeyError):
        datasets.fetch_dem("nowhere")
    (tmp_path / "assets" / "tif").mkdir(parents=True)
    heights = (np.linspace(400.0, 4392.0, 300 * 200).reshape(200, 300)).astype(np.float32)
    np.save(tmp_path / "dem.npy", heights)
    dem, cam, kw, what = datasets.real_dem_scene(tmp_path / "dem.npy")
    assert dem.shape == (200, 300) and dem.min() == 0.0 and kw["spacing"] == (10.0, 10.0) and "real DEM" in what
    span = 299 * 10.0
    assert abs(np.linalg.norm(np.subtract(cam["origin"], cam["look_at"])) - 1.25 * span) < 1e-6 * span and cam["fov_y"] == 42.0
    big = np.zeros((2, 20000), np.float32)
    np.save(tmp_path / "wide.npy", big)
    dem2, _, kw2, _ = datasets.real_dem_scene(tmp_path / "wide.npy")
    assert dem2.shape[1] <= 8193 and kw2["spacing"][0] == 10.0 * (20000 // dem2.shape[1] + (1 if 20000 % dem2.shape[1] else 0) if False else kw2["spacing"][0] / 10.0)
