"""An OpenEXR scanline reader with nothing but struct, zlib and numpy, independent of the library's own reader: the
tests parse what the C++ host writes with it."""
import struct
import zlib

import numpy as np


def read_exr_scanlines(path):
    """Single-part scanline file, FLOAT channels, no compression or ZIP / ZIPS: returns ({channel name: [h, w] float32},
    channel names in file order, header attributes as {name: (type, bytes)})."""
    data = open(path, "rb").read()
    magic, version = struct.unpack_from("<II", data, 0)
    assert magic == 20000630 and version == 2
    pos, attrs = 8, {}

    def cstr(at):
        end = data.index(b"\0", at)
        return data[at:end].decode(), end + 1
    while data[pos] != 0:
        name, pos = cstr(pos)
        kind, pos = cstr(pos)
        size = struct.unpack_from("<I", data, pos)[0]
        attrs[name] = (kind, data[pos + 4:pos + 4 + size])
        pos += 4 + size
    pos += 1
    assert attrs["compression"][0] == "compression" and attrs["lineOrder"] == ("lineOrder", b"\0")
    compression = attrs["compression"][1][0]
    assert compression in (0, 2, 3)
    lines = 16 if compression == 3 else 1
    xmin, ymin, xmax, ymax = struct.unpack("<iiii", attrs["dataWindow"][1])
    w, h = xmax - xmin + 1, ymax - ymin + 1
    names, cl, at = [], attrs["channels"][1], 0
    while cl[at] != 0:
        end = cl.index(b"\0", at)
        names.append(cl[at:end].decode())
        ptype, _, xs, ys = struct.unpack_from("<IIii", cl, end + 1)
        assert ptype == 2 and xs == 1 and ys == 1              # FLOAT, no subsampling
        at = end + 1 + 16
    blocks = (h + lines - 1) // lines
    offsets = struct.unpack_from("<%dQ" % blocks, data, pos)
    planes = {nm: np.empty((h, w), np.float32) for nm in names}
    for k, off in enumerate(offsets):
        y, size = struct.unpack_from("<iI", data, off)
        rows = min(lines, h - k * lines)
        want = 4 * w * len(names) * rows
        assert y == ymin + k * lines and size <= want
        raw = data[off + 8:off + 8 + size]
        if size < want:                                        # a block as long as its pixels is stored as it is
            d = np.frombuffer(zlib.decompress(raw), np.uint8).astype(np.int64)
            assert len(d) == want
            d = (np.cumsum(d) - 128 * np.arange(want, dtype=np.int64)) & 0xFF  # each byte was written as its difference from the last, + 128
            half = (want + 1) // 2
            out = np.empty(want, np.uint8)
            out[0::2] = d[:half]
            out[1::2] = d[half:]
            raw = out.tobytes()
        for r in range(rows):
            for c, nm in enumerate(names):
                planes[nm][k * lines + r] = np.frombuffer(raw, "<f4", w, 4 * w * (r * len(names) + c))
    return planes, names, attrs
