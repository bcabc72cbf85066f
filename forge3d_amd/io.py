"""Minimal image / DEM I/O around the offline render path: PNG out, PNG in,
.npy heightmaps.  Mirrors the helper names of the reference (``forge3d.numpy_to_png`` /
``forge3d.png_to_numpy``, python/forge3d/__init__.py:427-460); written against the PNG specification
(8-bit grey / RGB / RGBA, filter type 0 on write, all five filter types on read, no interlace)."""
from __future__ import annotations

import struct
import zlib
from pathlib import Path

import numpy as np

_SIGNATURE = b"\x89PNG\r\n\x1a\n"
_CHANNELS = {0: 1, 2: 3, 4: 2, 6: 4}


def _chunk(kind: bytes, payload: bytes) -> bytes:
    return struct.pack(">I", len(payload)) + kind + payload + struct.pack(">I", zlib.crc32(kind + payload) & 0xFFFFFFFF)


def numpy_to_png(path, array: np.ndarray) -> None:
    """Write an (H, W), (H, W, 3) or (H, W, 4) uint8 array as a PNG file."""
    arr = np.ascontiguousarray(array)
    if arr.dtype != np.uint8:
        raise ValueError("PNG encoder requires uint8 data")
    if arr.ndim == 2:
        color_type = 0
    elif arr.ndim == 3 and arr.shape[2] in (3, 4):
        color_type = 2 if arr.shape[2] == 3 else 6
    else:
        raise ValueError(f"Unsupported array shape: {arr.shape}")
    h, w = arr.shape[:2]
    rows = arr.reshape(h, -1)
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rows], axis=1).tobytes()  # filter type 0 per scanline
    data = (_SIGNATURE + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0))
            + _chunk(b"IDAT", zlib.compress(raw, 6)) + _chunk(b"IEND", b""))
    Path(path).write_bytes(data)


def png_to_numpy(path) -> np.ndarray:
    """Read an 8-bit, non-interlaced grey / grey+alpha / RGB / RGBA PNG into a uint8 array."""
    data = Path(path).read_bytes()
    if data[:8] != _SIGNATURE:
        raise ValueError("not a PNG file")
    pos, idat, header = 8, [], None
    while pos < len(data):
        (length,), kind = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        payload = data[pos + 8:pos + 8 + length]
        pos += 12 + length
        if kind == b"IHDR":
            header = struct.unpack(">IIBBBBB", payload)
        elif kind == b"IDAT":
            idat.append(payload)
        elif kind == b"IEND":
            break
    if header is None:
        raise ValueError("PNG without IHDR")
    w, h, depth, color_type, _, _, interlace = header
    if depth != 8 or interlace != 0 or color_type not in _CHANNELS:
        raise ValueError("only 8-bit non-interlaced grey/RGB/RGBA PNGs are supported")
    bpp = _CHANNELS[color_type]
    stride = w * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ftype, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if ftype == 0:
            cur = line
        elif ftype == 2:
            cur = (line + prev) & 255
        else:  # 1 (sub), 3 (average), 4 (Paeth): left neighbour of the SAME row -> sequential per pixel
            cur = np.zeros(stride, np.int32)
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ftype == 1:
                    pred = a
                elif ftype == 3:
                    pred = (a + b) >> 1
                elif ftype == 4:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if pa <= pb and pa <= pc else (b if pb <= pc else c)
                else:
                    raise ValueError(f"bad PNG filter type {ftype}")
                cur[i] = (line[i] + pred) & 255
        out[y] = cur
        prev = cur
    return out.reshape(h, w) if bpp == 1 else out.reshape(h, w, bpp)


def load_heightmap(path) -> np.ndarray:
    """A 2-D float32 heightmap from a ``.npy`` file (GeoTIFF readers are outside this repository)."""
    p = Path(path)
    if p.suffix.lower() != ".npy":
        raise ValueError(f"unsupported heightmap format {p.suffix!r}: convert the DEM to a 2-D float32 .npy")
    dem = np.load(p)
    if dem.ndim != 2:
        raise ValueError("heightmap must be a 2-D array")
    return np.ascontiguousarray(dem, np.float32)
