"""A scanline OpenEXR parser written for the tests, from the file layout alone.

It shares no code with the writer under test (csrc/exr.cpp) and does not call its reader: every field
is unpacked with `struct` and checked against what the OpenEXR file layout prescribes for a
single-part scanline image -- the magic number, the version field, typed header attributes, the
offset table, and one chunk per scanline for uncompressed data, channels stored in alphabetical
order within a scanline.  The reference writes its files with the `exr` crate's write_rgb_file
(src/main.rs:420-435): RGB, 32-bit float, linear, value * exposure.
"""
import struct

import numpy as np

MAGIC = 20000630  # 0x76 0x2f 0x31 0x01
PIXEL_BYTES = {0: 4, 1: 2, 2: 4}  # UINT, HALF, FLOAT
PIXEL_DTYPE = {0: "<u4", 1: "<f2", 2: "<f4"}
# attributes every OpenEXR header must carry, with their types
REQUIRED = {"channels": "chlist", "compression": "compression", "dataWindow": "box2i", "displayWindow": "box2i",
            "lineOrder": "lineOrder", "pixelAspectRatio": "float", "screenWindowCenter": "v2f",
            "screenWindowWidth": "float"}
FIXED_SIZE = {"box2i": 16, "compression": 1, "lineOrder": 1, "float": 4, "v2f": 8, "int": 4, "double": 8, "box2f": 16,
              "v2i": 8, "v3f": 12, "v3i": 12}


def _cstr(raw, at, limit=255):
    end = raw.index(b"\0", at)
    assert 0 < end - at <= limit, "attribute and type names are 1..255 bytes"
    return raw[at:end].decode("ascii"), end + 1


def parse_header(raw):
    """-> (attributes {name: (type, bytes)}, offset of the first byte after the header)."""
    magic, version = struct.unpack_from("<iI", raw, 0)
    assert magic == MAGIC, "not an OpenEXR file"
    assert version & 0xff == 2, "file format version 2"
    flags = version >> 8
    assert flags & ~0x04 == 0, "single-part scanline image: no tiled / deep / multipart bits"  # 0x04: long names
    at, attrs = 8, {}
    while raw[at] != 0:
        name, at = _cstr(raw, at)
        kind, at = _cstr(raw, at)
        (size,) = struct.unpack_from("<i", raw, at)
        at += 4
        assert size >= 0 and at + size <= len(raw)
        assert name not in attrs, f"attribute {name} twice"
        if kind in FIXED_SIZE:
            assert size == FIXED_SIZE[kind], (name, kind, size)
        attrs[name] = (kind, raw[at:at + size])
        at += size
    for name, kind in REQUIRED.items():
        assert name in attrs, f"required attribute {name} missing"
        assert attrs[name][0] == kind, (name, attrs[name][0])
    return attrs, at + 1


def parse_channels(value):
    """chlist -> [(name, pixel type, x sampling, y sampling)], as stored."""
    at, out = 0, []
    while value[at] != 0:
        name, at = _cstr(value, at)
        ptype, plinear, r0, r1, r2, xs, ys = struct.unpack_from("<iB3Bii", value, at)
        at += 16
        assert ptype in PIXEL_BYTES and plinear in (0, 1) and (r0, r1, r2) == (0, 0, 0)
        out.append((name, ptype, xs, ys))
    assert at + 1 == len(value), "channel list ends with one null byte"
    assert [c[0] for c in out] == sorted(c[0] for c in out), "channels are stored in alphabetical order"
    return out


def read_scanline_exr(path):
    """-> {channel name: (h, w) array} and the parsed attributes, of an uncompressed single-part scanline file."""
    raw = open(path, "rb").read()
    attrs, at = parse_header(raw)
    channels = parse_channels(attrs["channels"][1])
    assert attrs["compression"][1] == b"\0", "NO_COMPRESSION"
    line_order = attrs["lineOrder"][1][0]
    assert line_order in (0, 1), "INCREASING_Y or DECREASING_Y"
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    assert w > 0 and h > 0
    assert all(xs == 1 and ys == 1 for _, _, xs, ys in channels)
    # offset table: one entry per chunk; an uncompressed chunk is one scanline
    offsets = struct.unpack_from(f"<{h}Q", raw, at)
    table_end = at + 8 * h
    line_bytes = sum(PIXEL_BYTES[t] for _, t, _, _ in channels) * w
    planes = {name: np.zeros((h, w), PIXEL_DTYPE[t]) for name, t, _, _ in channels}
    seen = set()
    for k, off in enumerate(offsets):
        assert table_end <= off <= len(raw) - 8 - line_bytes, "chunk inside the file, after the offset table"
        y, size = struct.unpack_from("<ii", raw, off)
        assert size == line_bytes, "uncompressed: the chunk holds exactly one scanline of every channel"
        assert y == (y0 + k if line_order == 0 else y1 - k), "offset table entries follow the line order"
        assert y not in seen
        seen.add(y)
        p = off + 8
        for name, t, _, _ in channels:
            n = PIXEL_BYTES[t] * w
            planes[name][y - y0] = np.frombuffer(raw, PIXEL_DTYPE[t], w, p)
            p += n
    assert len(seen) == h
    # the chunks fill the rest of the file exactly
    assert table_end + h * (8 + line_bytes) == len(raw)
    return planes, attrs


def read_rgb(path):
    """(h, w, 3) float32 image of a file with float R, G, B channels; also returns the attributes."""
    planes, attrs = read_scanline_exr(path)
    assert set(planes) == {"R", "G", "B"}, sorted(planes)
    assert all(p.dtype == np.dtype("<f4") for p in planes.values()), "32-bit float channels"
    return np.stack([planes["R"], planes["G"], planes["B"]], axis=-1), attrs
