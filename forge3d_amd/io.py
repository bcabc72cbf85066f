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


# ---- float / integer GeoTIFF DEMs (TIFF 6.0 baseline + the GeoTIFF scale tag) ------------------------------------
_TIFF_TYPES = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("II", 8), 6: ("b", 1), 8: ("h", 2), 9: ("i", 4),
               11: ("f", 4), 12: ("d", 8), 16: ("Q", 8), 17: ("q", 8)}


def _lzw_decode(data: bytes) -> bytes:
    """TIFF LZW (MSB-first codes, 9..12 bits, ClearCode 256, EOI 257, early change)."""
    out = bytearray()
    table = [bytes([i]) for i in range(256)] + [b"", b""]
    bits, nbits, width, prev, pos = 0, 0, 9, None, 0
    n = len(data)
    while True:
        while nbits < width and pos < n:
            bits = (bits << 8) | data[pos]
            pos += 1
            nbits += 8
        if nbits < width:
            break
        code = (bits >> (nbits - width)) & ((1 << width) - 1)
        nbits -= width
        if code == 257:
            break
        if code == 256:
            table = table[:258]
            width, prev = 9, None
            continue
        if prev is None:
            entry = table[code]
        elif code < len(table):
            entry = table[code]
            table.append(prev + entry[:1])
        else:
            entry = prev + prev[:1]
            table.append(entry)
        out += entry
        prev = entry
        if len(table) + 1 >= (1 << width) and width < 12:  # early change: widen one code early
            width += 1
    return bytes(out)


def _packbits_decode(data: bytes) -> bytes:
    out, i = bytearray(), 0
    while i < len(data):
        n = data[i]
        i += 1
        if n < 128:
            out += data[i:i + n + 1]
            i += n + 1
        elif n > 128:
            out += data[i:i + 1] * (257 - n)
            i += 1
    return bytes(out)


def read_geotiff(path):
    """(heights float32 (H, W), info) from a single-band TIFF / GeoTIFF DEM: little or big endian, classic TIFF,
    strips or tiles, uncompressed / LZW / Deflate / PackBits, predictors 1, 2 (integers) and 3 (floating point),
    8/16/32-bit integer or 32/64-bit float samples.  info: {"pixel_scale": (sx, sy) or None, "tiepoint": ...,
    "nodata": float or None}.  (The reference reads DEMs through rasterio, python/forge3d/io.py.)"""
    raw = Path(path).read_bytes()
    if raw[:2] == b"II":
        e = "<"
    elif raw[:2] == b"MM":
        e = ">"
    else:
        raise ValueError("not a TIFF file")
    if struct.unpack(e + "H", raw[2:4])[0] != 42:
        raise ValueError("only classic TIFF (version 42) is supported, not BigTIFF")
    (ifd,) = struct.unpack(e + "I", raw[4:8])
    (count,) = struct.unpack(e + "H", raw[ifd:ifd + 2])
    tags = {}
    for k in range(count):
        tag, typ, n, _ = struct.unpack(e + "HHI4s", raw[ifd + 2 + 12 * k:ifd + 14 + 12 * k])
        fmt, size = _TIFF_TYPES.get(typ, ("B", 1))
        field = raw[ifd + 10 + 12 * k:ifd + 14 + 12 * k]
        if size * n > 4:
            (off,) = struct.unpack(e + "I", field)
            field = raw[off:off + size * n]
        if typ == 2:
            tags[tag] = field[:n].split(b"\0")[0].decode("latin-1")
        elif typ == 5:
            v = struct.unpack(e + "II" * n, field[:8 * n])
            tags[tag] = tuple(v[2 * i] / max(v[2 * i + 1], 1) for i in range(n))
        else:
            tags[tag] = struct.unpack(e + fmt * n, field[:size * n])
    one = lambda t, d=None: tags[t][0] if t in tags else d  # noqa: E731
    w, h = one(256), one(257)
    bits, fmt_code, spp = one(258, 1), one(339, 1), one(277, 1)
    compression, predictor = one(259, 1), one(317, 1)
    if spp != 1:
        raise ValueError(f"DEM TIFFs have one band, this file has {spp}")
    dtype = {(8, 1): "u1", (16, 1): "u2", (32, 1): "u4", (8, 2): "i1", (16, 2): "i2", (32, 2): "i4", (32, 3): "f4",
             (64, 3): "f8"}.get((bits, fmt_code))
    if dtype is None:
        raise ValueError(f"unsupported TIFF sample layout: {bits} bits, SampleFormat {fmt_code}")
    bps = bits // 8

    def decode(chunk: bytes, rows: int, cols: int) -> np.ndarray:
        if compression in (8, 32946):
            chunk = zlib.decompress(chunk)
        elif compression == 5:
            chunk = _lzw_decode(chunk)
        elif compression == 32773:
            chunk = _packbits_decode(chunk)
        elif compression != 1:
            raise ValueError(f"unsupported TIFF compression {compression}")
        chunk = chunk[:rows * cols * bps]
        if predictor == 3:  # floating-point predictor: bytes differenced, then planes of significance per row
            b = np.frombuffer(chunk, np.uint8).reshape(rows, cols * bps).astype(np.uint8)
            b = np.cumsum(b, axis=1, dtype=np.uint8)
            b = b.reshape(rows, bps, cols)  # most significant byte plane first
            b = b.transpose(0, 2, 1)
            if e == "<":
                b = b[:, :, ::-1]
            return np.ascontiguousarray(b).view(e + dtype).reshape(rows, cols)
        a = np.frombuffer(chunk, e + dtype).reshape(rows, cols)
        if predictor == 2:
            a = np.cumsum(a, axis=1, dtype=a.dtype.newbyteorder("="))
        return a

    out = np.zeros((h, w), np.float64 if dtype == "f8" else np.float32)
    if 322 in tags:  # tiles
        tw, th = one(322), one(323)
        offsets, counts = tags[324], tags[325]
        across = (w + tw - 1) // tw
        for i, (off, n) in enumerate(zip(offsets, counts)):
            ty, tx = divmod(i, across)
            tile = decode(raw[off:off + n], th, tw)
            y0, x0 = ty * th, tx * tw
            out[y0:y0 + th, x0:x0 + tw] = tile[: h - y0, : w - x0]
    else:
        rps = min(one(278, h), h)
        for i, (off, n) in enumerate(zip(tags[273], tags[279])):
            y0 = i * rps
            rows = min(rps, h - y0)
            out[y0:y0 + rows] = decode(raw[off:off + n], rows, w)
    nodata = None
    if 42113 in tags:
        try:
            nodata = float(tags[42113])
        except ValueError:
            nodata = None
    info = {"pixel_scale": tuple(tags[33550][:2]) if 33550 in tags else None,
            "tiepoint": tuple(tags[33922][:6]) if 33922 in tags else None, "nodata": nodata}
    return np.ascontiguousarray(out, np.float32), info


def load_heightmap(path, fill_nodata: bool = True) -> np.ndarray:
    """A 2-D float32 heightmap from a ``.npy`` file or a single-band (Geo)TIFF; GDAL nodata cells are replaced by
    the smallest valid height when fill_nodata (the path tracer rejects non-finite samples)."""
    p = Path(path)
    suffix = p.suffix.lower()
    if suffix == ".npy":
        dem = np.load(p)
    elif suffix in (".tif", ".tiff"):
        dem, info = read_geotiff(p)
        bad = ~np.isfinite(dem)
        if info["nodata"] is not None:
            bad |= dem == np.float32(info["nodata"])
        if fill_nodata and bad.any():
            dem = np.where(bad, dem[~bad].min() if (~bad).any() else 0.0, dem)
    else:
        raise ValueError(f"unsupported heightmap format {p.suffix!r}: use a 2-D float32 .npy or a GeoTIFF")
    if dem.ndim != 2:
        raise ValueError("heightmap must be a 2-D array")
    return np.ascontiguousarray(dem, np.float32)


# ---- Radiance .hdr / .rgbe environment maps ---------------------------------------------------------------------------
class HdrError(ValueError):
    """A file that is not a readable Radiance picture."""


def _hdr_rows(buf: memoryview, pos: int, width: int, height: int) -> np.ndarray:
    """(height, width, 4) uint8 RGBE from the scanline section: per row either the adaptive run-length form (2, 2,
    width hi, width lo, then the four components one after the other as runs: count > 128 repeats one byte count - 128
    times, otherwise `count` literal bytes) or `width` flat pixels."""
    out = np.empty((height, width, 4), np.uint8)
    n = len(buf)
    for y in range(height):
        if pos + 4 > n:
            raise HdrError(f"Failed to read scanline header at row {y}: failed to fill whole buffer")
        head = bytes(buf[pos:pos + 4])
        if head[0] == 2 and head[1] == 2 and head[2] == ((width >> 8) & 0xFF) and head[3] == (width & 0xFF):
            pos += 4
            for c in range(4):
                x = 0
                row = out[y, :, c]
                while x < width:
                    if pos >= n:
                        raise HdrError("Failed to read RLE run info: failed to fill whole buffer")
                    count = buf[pos]
                    pos += 1
                    if count > 128:
                        count -= 128
                        if x + count > width:
                            raise HdrError("HDR RLE run exceeds scanline width")
                        if pos >= n:
                            raise HdrError("Failed to read RLE repeat value: failed to fill whole buffer")
                        row[x:x + count] = buf[pos]
                        pos += 1
                    else:
                        if x + count > width:
                            raise HdrError("HDR literal run exceeds scanline width")
                        if pos + count > n:
                            raise HdrError("Failed to read literal value: failed to fill whole buffer")
                        row[x:x + count] = np.frombuffer(buf[pos:pos + count], np.uint8)
                        pos += count
                    x += count
        else:
            if pos + 4 * width > n:
                raise HdrError(f"Failed to read pixel data at row {y}: failed to fill whole buffer")
            out[y] = np.frombuffer(buf[pos:pos + 4 * width], np.uint8).reshape(width, 4)
            pos += 4 * width
    return out


def read_hdr(path) -> np.ndarray:
    """A Radiance picture as (H, W, 3) float32 linear RGB, rows in file order.  Behaviour of the reference's loader
    (src/formats/hdr.rs:49-288): magic `#?RADIANCE` or `#?RGBE`; a FORMAT= line is required and must be
    32-bit_rle_rgbe or 32-bit_rle_xyze; the resolution line is four tokens of which the second is the height and the
    fourth the width (orientation flags are not interpreted); a pixel (r, g, b, e) is (r, g, b) * 2^(e - 136), all
    zero when e == 0 (no half-step offset)."""
    data = Path(path).read_bytes()
    end = data.find(b"\n")
    first = data[:end + 1 if end >= 0 else len(data)]
    if not (first.startswith(b"#?RADIANCE") or first.startswith(b"#?RGBE")):
        raise HdrError("Invalid HDR file: missing magic header")
    pos, fmt = len(first), False
    while pos < len(data):
        end = data.find(b"\n", pos)
        end = len(data) if end < 0 else end + 1
        line = data[pos:end].decode("latin-1").strip()
        pos = end
        if not line:
            break
        if line.startswith("FORMAT="):
            if line not in ("FORMAT=32-bit_rle_rgbe", "FORMAT=32-bit_rle_xyze"):
                raise HdrError(f"Unsupported HDR format: {line}")
            fmt = True
    if not fmt:
        raise HdrError("HDR file missing FORMAT specification")
    end = data.find(b"\n", pos)
    end = len(data) if end < 0 else end + 1
    resolution = data[pos:end].decode("latin-1").strip()
    parts = resolution.split()
    if len(parts) != 4:
        raise HdrError(f"Invalid HDR resolution line: {resolution}")
    # the reference parses both with str::parse::<u32> (src/formats/hdr.rs:137-143): an optional single leading '+', then
    # ASCII decimal digits -- any number of them, leading zeros included -- with a value that fits 32 bits.  int() alone
    # would also take "1_0", " 5" and "-0".
    def parse_u32(text):
        digits = text[1:] if text.startswith("+") else text
        if not digits or not digits.isascii() or not digits.isdigit():
            return None
        value = int(digits.lstrip("0") or "0") if len(digits.lstrip("0")) <= 10 else 1 << 32
        return value if value <= 0xFFFFFFFF else None

    height, width = parse_u32(parts[1]), parse_u32(parts[3])
    if height is None:
        raise HdrError(f"Invalid HDR height: {parts[1]}")
    if width is None:
        raise HdrError(f"Invalid HDR width: {parts[3]}")
    if width <= 0 or height <= 0:
        raise HdrError("HDR image dimensions cannot be zero")
    # a scanline is at least 4 bytes in either form: a header that promises more rows than the file can hold is refused
    # BEFORE (height, width, 4) bytes are allocated for it (a 30-byte file could ask for gigabytes)
    remaining = len(data) - end
    # (run-length rows hold at most 127 pixels per 2 bytes and component: < 16 pixels per byte)
    # (only where the allocation would matter: small pictures fail row by row, in the reference's order and words)
    if width * height * 4 > (64 << 20) and (height > remaining // 4 or width * height > 64 * max(remaining, 1)):
        raise HdrError(f"Failed to read scanline header at row {min(height, remaining // 4)}: failed to fill whole buffer")
    try:
        rgbe = _hdr_rows(memoryview(data), end, width, height)
    except (MemoryError, OverflowError):
        raise HdrError(f"HDR image of {width} x {height} pixels does not fit in memory") from None
    scale = np.ldexp(np.float32(1.0), rgbe[..., 3].astype(np.int32) - 136).astype(np.float32)
    scale[rgbe[..., 3] == 0] = 0.0
    return rgbe[..., :3].astype(np.float32) * scale[..., None]


def write_hdr(path, rgb: np.ndarray) -> None:
    """(H, W, 3) float32 -> flat (uncompressed) Radiance picture; the shared exponent is that of the largest channel,
    mantissas truncated (what `read_hdr` inverts up to the 8-bit mantissa)."""
    rgb = np.ascontiguousarray(rgb, np.float32)
    if rgb.ndim != 3 or rgb.shape[2] != 3:
        raise ValueError("rgb must be (H, W, 3)")
    top = np.max(rgb, axis=2)
    mant, exp = np.frexp(top)  # top = mant * 2^exp, mant in [0.5, 1)
    valid = top > 1e-32
    scale = np.where(valid, np.ldexp(np.float32(1.0), 8 - exp), 0.0).astype(np.float32)
    out = np.zeros(rgb.shape[:2] + (4,), np.uint8)
    out[..., :3] = np.clip(rgb * scale[..., None], 0.0, 255.0).astype(np.uint8)
    out[..., 3] = np.where(valid, exp + 128, 0).astype(np.uint8)
    h, w = rgb.shape[:2]
    Path(path).write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {h} +X {w}\n".encode() + out.tobytes())
