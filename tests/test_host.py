"""CPU tests of the C++ host layer (csrc/host/): Wavefront ingest against the numpy restatement,
EXR reader/writer, camera matrix."""
import os
import struct
import zlib

import numpy as np
import pytest

from gpuraytracer_amd import host
from gpuraytracer_amd.camera import camera_matrix
from oracle import scene_np

OBJ = """# test scene
mtllib t.mtl
o Lamp
v 0.0 2.0 0.0
v 1.0 2.0 0.0
v 0.0 2.0 1.0
vt 0.0 0.0
vt 1.0 0.0
vt 0.0 1.0
vn 0.0 -1.0 0.0
usemtl Wall
usemtl Light
s off
f 1/1/1 2/2/1 3/3/1
o Floor
v -4.0 0.0 -4.0
v 4.0 0.0 -4.0
v 4.0 0.0 4.0
v -4.0 0.0 4.0
v 0.5 0.5 0.5
vt 0.25 0.75
vn 0.0 1.0 0.0
vn 0.577350 0.577350 0.577350
usemtl Wall
f 4/1/2 6/3/2 5/2/2
f 4/1/2 7/4/2 6/3/2
f 4/4/3 5/1/2 8/2/3
f 8/2/3 5/1/2 4/4/3
o Ball
usemtl Glass
f 5/1/2 6/2/3 8/3/3
"""
MTL = """# materials
newmtl Light
Ns 0.0
Kd 0.0 0.0 0.0
Ke 10.0 12.5 15.0
Ni 1.0

newmtl Wall
Ns 500.0
Kd 0.8 0.7 0.6
Ks 0.1 0.2 0.3
Ni 1.00029

newmtl Glass
Ns 50
Kd 0.1 0.1 0.1
Ks 1 1 1
Ni 1.474
"""


@pytest.fixture()
def tiny_scene(tmp_path):
    base = str(tmp_path / "t")
    open(base + ".obj", "w").write(OBJ)
    open(base + ".mtl", "w").write(MTL)
    return base


def test_ingest_conventions(tiny_scene):
    models, shadings, lights, materials = host.load_wavefront(tiny_scene)
    assert len(models) == 6 and lights == 1 and len(materials) == 3
    # last usemtl of an object wins (Scene.cpp:633-634): the Lamp object is emissive and comes first
    assert shadings["material_id"][0] == 0
    assert np.allclose(materials["emission_ior"][0], [10.0, 12.5, 15.0, 1.0])
    # Ns / 1000 -> roughness, Ni -> ior
    assert materials["specular_roughness"][1][3] == np.float32(500.0) / np.float32(1000.0)
    assert materials["emission_ior"][2][3] == np.float32(1.474)
    # the rest is ordered by descending |cross(u, v)|; the two equal floor halves keep file order
    c = np.cross(models["u"][1:], models["v"][1:])
    area = np.linalg.norm(c, axis=1)
    assert (np.diff(area) <= 0).all()
    assert np.array_equal(models["true_normal"][0], np.cross(models["u"][0], models["v"][0]))


def test_ingest_matches_numpy_restatement(tiny_scene):
    models, shadings, lights, materials = host.load_wavefront(tiny_scene)
    m2, s2, l2, mat2, _ = scene_np.load_wavefront(tiny_scene)
    assert lights == l2
    assert models.tobytes() == m2.tobytes()
    assert materials.tobytes() == mat2.tobytes()
    for f in ("material_id", "normals", "tex_p", "tex_u", "tex_v"):
        assert np.array_equal(shadings[f], s2[f]), f
    assert np.array_equal(shadings["tangents"], s2["tangents"], equal_nan=True)


@pytest.mark.skipif(not os.path.isdir("/root/reference/res/models"), reason="reference models not present")
@pytest.mark.parametrize("name", ["cubes", "box", "sphere", "_ObiWan"])
def test_ingest_of_reference_models_matches_fixtures(name):
    from conftest import load_scene
    models, shadings, lights, materials = host.load_wavefront("/root/reference/res/models/" + name)
    fm, fs, fl, fmat = load_scene(name)
    assert lights == fl
    assert models.tobytes() == fm.tobytes() and materials.tobytes() == fmat.tobytes()
    assert np.array_equal(shadings["material_id"], fs["material_id"])
    assert np.array_equal(shadings["normals"], fs["normals"])
    assert np.array_equal(shadings["tangents"], fs["tangents"], equal_nan=True)


def test_missing_model_is_an_error(tmp_path):
    with pytest.raises(ValueError):
        host.load_wavefront(str(tmp_path / "nothing"))


def test_exr_roundtrip(tmp_path):
    rng = np.random.default_rng(1)
    img = rng.uniform(0, 4, size=(13, 17, 3)).astype(np.float32)
    p = str(tmp_path / "a.exr")
    host.write_exr(p, img)
    back = host.read_exr(p)
    assert back.shape == (13, 17, 4)
    assert np.array_equal(back[..., :3], img) and (back[..., 3] == 1).all()
    raw = open(p, "rb").read()
    assert raw[:4] == bytes([0x76, 0x2F, 0x31, 0x01])            # OpenEXR magic
    rgba = rng.uniform(0, 1, size=(5, 4, 4)).astype(np.float32)
    host.write_exr(p, rgba)
    assert np.array_equal(host.read_exr(p), rgba)


def test_exr_writer_zip(tmp_path):
    """ZIP blocks of 16 lines, read back by the library's reader and by the tests' own parser (struct + zlib)."""
    from exr_plain import read_exr_scanlines
    rng = np.random.default_rng(10)
    y, x = np.mgrid[0:45, 0:37].astype(np.float32)
    img = np.stack([x / 8, (x + y) / 16, np.floor(y / 4)], axis=-1).astype(np.float32)
    img[30:, :] = rng.uniform(0, 1, size=(15, 37, 3)).astype(np.float32)     # noise in the last block
    p = str(tmp_path / "z.exr")
    host.write_exr(p, img, compression=3)
    q = str(tmp_path / "n.exr")
    host.write_exr(q, img)
    assert os.path.getsize(p) < os.path.getsize(q)
    assert np.array_equal(host.read_exr(p)[..., :3], img)
    planes, names, attrs = read_exr_scanlines(p)
    assert names == ["B", "G", "R"] and attrs["compression"][1] == b"\x03"
    assert np.array_equal(np.stack([planes["R"], planes["G"], planes["B"]], axis=-1), img)
    rgba = rng.uniform(0, 1, size=(17, 5, 4)).astype(np.float32)
    host.write_exr(p, rgba, compression=-1)                                  # the host program's choice: ZIP from 16 pixels on
    assert read_exr_scanlines(p)[2]["compression"][1] == b"\x03" and np.array_equal(host.read_exr(p), rgba)
    host.write_exr(p, rgba[:15], compression=-1)
    assert read_exr_scanlines(p)[2]["compression"][1] == b"\x00" and np.array_equal(host.read_exr(p), rgba[:15])


def _rle(data):
    """OpenEXR's run-length code: count byte n, n < 0: -n literal bytes follow, n >= 0: the next byte n + 1 times."""
    out, i, n = bytearray(), 0, len(data)
    while i < n:
        run = 1
        while i + run < n and data[i + run] == data[i] and run < 128:
            run += 1
        if run >= 3:
            out += struct.pack("b", run - 1) + bytes([data[i]])
            i += run
        else:
            j = i
            while j < n and j - i < 127 and not (j + 2 < n and data[j] == data[j + 1] == data[j + 2]):
                j += 1
            out += struct.pack("b", -(j - i)) + bytes(data[i:j])
            i = j
    return bytes(out)


def _zip_exr(path, img_half_rgb, compression=3):
    """Writes a compressed HALF RGB scanline EXR with Python, to test the reader: 3 = ZIP (16-line blocks),
    2 = ZIPS (one line per block), 1 = RLE (one line per block)."""
    h, w, _ = img_half_rgb.shape
    lines = 16 if compression == 3 else 1

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(val)) + val
    ch = b""
    for n in ("B", "G", "R"):
        ch += n.encode() + b"\0" + struct.pack("<IBBBBII", 1, 0, 0, 0, 0, 1, 1)
    ch += b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    hdr = struct.pack("<II", 20000630, 2) + attr("channels", "chlist", ch) + attr("compression", "compression", bytes([compression]))
    hdr += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0")
    hdr += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    hdr += b"\0"
    blocks = []
    for y0 in range(0, h, lines):
        raw = b""
        for y in range(y0, min(h, y0 + lines)):
            for c in (2, 1, 0):
                raw += img_half_rgb[y, :, c].astype("<f2").tobytes()
        a = np.frombuffer(raw, np.uint8)
        inter = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
        pred = inter.copy()
        pred[1:] = (inter[1:] - inter[:-1] + 128 + 256) % 256
        z = _rle(pred.astype(np.uint8).tobytes()) if compression == 1 else zlib.compress(pred.astype(np.uint8).tobytes())
        blocks.append((y0, z if len(z) < len(raw) else raw))      # EXR stores a block raw when zlib does not shrink it
    off = len(hdr) + 8 * len(blocks)
    table = b""
    for _, z in blocks:
        table += struct.pack("<Q", off)
        off += 8 + len(z)
    with open(path, "wb") as fh:
        fh.write(hdr + table)
        for y0, z in blocks:
            fh.write(struct.pack("<ii", y0, len(z)) + z)


@pytest.mark.parametrize("compression", [1, 2, 3])
def test_exr_reader_compressed_half(tmp_path, compression):
    rng = np.random.default_rng(2)
    img = rng.uniform(0, 2, size=(37, 21, 3)).astype(np.float16)
    img[5:9, 3:15] = np.float16(0.75)                              # flat areas: runs for the run-length code
    img[20:, :4] = np.float16(0.0)
    p = str(tmp_path / "z.exr")
    _zip_exr(p, img, compression)
    back = host.read_exr(p)
    assert np.array_equal(back[..., :3], img.astype(np.float32))


def _tiled_exr(path, img_rgb, tile_w, tile_h, compression, mip_levels=False):
    """Writes a tiled (one level) FLOAT RGB EXR with Python: version flag 0x200, a `tiles` attribute, one offset per
    tile in row-major tile order, each tile with its coordinates, level (0, 0), size and data (rows of the tile,
    channels B, G, R within a row), compressed like a scanline block."""
    h, w, _ = img_rgb.shape

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(val)) + val
    ch = b""
    for n in ("B", "G", "R"):
        ch += n.encode() + b"\0" + struct.pack("<IBBBBII", 2, 0, 0, 0, 0, 1, 1)
    ch += b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    hdr = struct.pack("<II", 20000630, 2 | 0x200) + attr("channels", "chlist", ch)
    hdr += attr("compression", "compression", bytes([compression]))
    hdr += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0")
    hdr += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    hdr += attr("tiles", "tiledesc", struct.pack("<IIB", tile_w, tile_h, 1 if mip_levels else 0)) + b"\0"
    tiles = []
    for ty in range((h + tile_h - 1) // tile_h):
        for tx in range((w + tile_w - 1) // tile_w):
            part = img_rgb[ty * tile_h:(ty + 1) * tile_h, tx * tile_w:(tx + 1) * tile_w]
            raw = b"".join(part[y, :, c].astype("<f4").tobytes() for y in range(part.shape[0]) for c in (2, 1, 0))
            data = raw
            if compression:
                a = np.frombuffer(raw, np.uint8)
                inter = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
                pred = inter.copy()
                pred[1:] = (inter[1:] - inter[:-1] + 128 + 256) % 256
                z = _rle(pred.astype(np.uint8).tobytes()) if compression == 1 else zlib.compress(pred.astype(np.uint8).tobytes())
                data = z if len(z) < len(raw) else raw
            tiles.append(struct.pack("<iiiii", tx, ty, 0, 0, len(data)) + data)
    if mip_levels:
        # the smaller levels follow level 0 in the offset table, down to 1 x 1; their pixels do not matter here
        lw, lh = w, h
        while lw > 1 or lh > 1:
            lw, lh = max(1, lw // 2), max(1, lh // 2)
            lvl = len(bin(w // lw)) - 3
            for ty in range((lh + tile_h - 1) // tile_h):
                for tx in range((lw + tile_w - 1) // tile_w):
                    cw, chh = min(tile_w, lw - tx * tile_w), min(tile_h, lh - ty * tile_h)
                    raw = bytes(12 * cw * chh)
                    tiles.append(struct.pack("<iiiii", tx, ty, lvl, lvl, len(raw)) + raw)
    off = len(hdr) + 8 * len(tiles)
    table = b""
    for t in tiles:
        table += struct.pack("<Q", off)
        off += len(t)
    with open(path, "wb") as fh:
        fh.write(hdr + table + b"".join(tiles))


@pytest.mark.parametrize("compression", [0, 1, 3])
def test_exr_reader_tiled(tmp_path, compression):
    rng = np.random.default_rng(4)
    img = rng.uniform(0, 3, size=(45, 70, 3)).astype(np.float32)     # 32 x 16 tiles: ragged on both edges
    img[10:20, 5:60] = np.float32(1.5)
    p = str(tmp_path / "t.exr")
    _tiled_exr(p, img, 32, 16, compression)
    back = host.read_exr(p)
    assert back.shape == (45, 70, 4)
    assert np.array_equal(back[..., :3], img) and (back[..., 3] == 1).all()


def _piz_exr(path, img, types, tile=None, stats=None):
    """Writes a PIZ-compressed EXR with Python (tests/exr_piz.py): img is (h, w, 3 or 4) RGB(A), `types` gives each
    channel's sample type (1 = HALF, 2 = FLOAT) by name; scanline blocks of 32 lines, or tiles of `tile` = (w, h)."""
    from exr_piz import piz_block
    h, w, nc = img.shape
    names = ("A", "B", "G", "R") if nc == 4 else ("B", "G", "R")
    src = {"R": 0, "G": 1, "B": 2, "A": 3}

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(val)) + val
    ch = b"".join(n.encode() + b"\0" + struct.pack("<IBBBBII", types[n], 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    hdr = struct.pack("<II", 20000630, 2 | (0x200 if tile else 0)) + attr("channels", "chlist", ch)
    hdr += attr("compression", "compression", bytes([4]))
    hdr += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0")
    hdr += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    if tile:
        hdr += attr("tiles", "tiledesc", struct.pack("<IIB", tile[0], tile[1], 0))
    hdr += b"\0"

    def block(part):
        planes = []
        for n in names:
            v = np.ascontiguousarray(part[:, :, src[n]])
            if types[n] == 1:
                planes.append((v.astype("<f2").view(np.uint16), 1))
            else:
                planes.append((v.astype("<f4").view(np.uint16), 2))
        return piz_block(planes, stats)
    chunks = []
    if tile:
        for ty in range((h + tile[1] - 1) // tile[1]):
            for tx in range((w + tile[0] - 1) // tile[0]):
                data = block(img[ty * tile[1]:(ty + 1) * tile[1], tx * tile[0]:(tx + 1) * tile[0]])
                chunks.append(struct.pack("<iiiii", tx, ty, 0, 0, len(data)) + data)
    else:
        for y0 in range(0, h, 32):
            data = block(img[y0:y0 + 32])
            chunks.append(struct.pack("<ii", y0, len(data)) + data)
    off = len(hdr) + 8 * len(chunks)
    table = b""
    for c in chunks:
        table += struct.pack("<Q", off)
        off += len(c)
    with open(path, "wb") as fh:
        fh.write(hdr + table + b"".join(chunks))


def _piz_test_image(rng, h, w, nc):
    """Smooth ramps, flat areas and a patch of noise: few distinct words per block, long runs after the wavelet."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([(x + 2 * y) / 64, np.sin(x / 9) * np.cos(y / 7) + 1, (x * y) / 512, 0.5 + 0 * x][:nc], axis=-1)
    img = img.astype(np.float16).astype(np.float32)                  # the same values whether stored HALF or FLOAT
    img[5:9, 3:15] = 0.75
    img[h // 2:, :w // 3] = 0.0
    img[2:12, w - 12:w - 2] = rng.uniform(0, 2, size=(10, 10, nc)).astype(np.float16)
    return img


def test_exr_reader_piz_half_scanlines(tmp_path):
    rng = np.random.default_rng(5)
    img = _piz_test_image(rng, 71, 53, 3)                            # blocks of 32, 32 and 7 lines, odd sizes
    stats = {}
    p = str(tmp_path / "p.exr")
    _piz_exr(p, img, {"R": 1, "G": 1, "B": 1}, stats=stats)
    assert stats["coded_blocks"] == 3 and stats["repeats"] > 0 and stats["max_value"] < (1 << 14)
    back = host.read_exr(p)
    assert back.shape == (71, 53, 4)
    assert np.array_equal(back[..., :3], img) and (back[..., 3] == 1).all()


def test_exr_reader_piz_mixed_types_and_one_line(tmp_path):
    rng = np.random.default_rng(6)
    img = _piz_test_image(rng, 33, 40, 4)                            # second block is a single line: no wavelet level
    p = str(tmp_path / "p.exr")
    _piz_exr(p, img, {"R": 2, "G": 2, "B": 2, "A": 1})
    assert np.array_equal(host.read_exr(p), img)
    flat = np.zeros((3, 5, 3), np.float32)                           # only the word 0 in use: empty bitmap
    _piz_exr(p, flat, {"R": 1, "G": 1, "B": 1})
    assert np.array_equal(host.read_exr(p)[..., :3], flat)


def test_exr_reader_piz_many_values_tiled(tmp_path):
    """More than 2^14 distinct words in a tile: the wavelet's modulo-2^16 form, and codes longer than 14 bits."""
    rng = np.random.default_rng(7)
    img = np.zeros((70, 200, 3), np.float32)
    img[:, :110] = rng.uniform(0, 3, size=(70, 110, 3)).astype(np.float32)
    img[40:, 150:] = 2.5
    stats = {}
    p = str(tmp_path / "p.exr")
    _piz_exr(p, img, {"R": 2, "G": 2, "B": 2}, tile=(128, 64), stats=stats)
    assert stats["coded_blocks"] >= 1 and stats["max_value"] >= (1 << 14) and stats["longest_code"] > 14
    back = host.read_exr(p)
    assert np.array_equal(back[..., :3], img)


def test_exr_reader_piz_rejects_damaged_blocks(tmp_path):
    """Damaged files end in an error or in different pixels, never in a crash."""
    rng = np.random.default_rng(8)
    img = _piz_test_image(rng, 64, 90, 3)
    stats = {}
    p = str(tmp_path / "p.exr")
    _piz_exr(p, img, {"R": 1, "G": 2, "B": 1}, stats=stats)
    assert stats["coded_blocks"] == 2
    good = open(p, "rb").read()
    start = good.index(b"screenWindowWidth") + 40                    # past the header: offsets and blocks
    q = str(tmp_path / "q.exr")
    errors = 0
    for k in range(300):
        bad = bytearray(good)
        if k % 3 == 0:
            bad = bad[:int(rng.integers(start, len(good)))]
        else:
            for _ in range(int(rng.integers(1, 4))):
                bad[int(rng.integers(start, len(bad)))] = int(rng.integers(0, 256))
        open(q, "wb").write(bytes(bad))
        try:
            host.read_exr(q)
        except (ValueError, OSError, RuntimeError):
            errors += 1
    assert errors > 150


def test_exr_reader_takes_level_0_of_mip_mapped_tiles(tmp_path):
    rng = np.random.default_rng(9)
    img = rng.uniform(0, 3, size=(40, 64, 3)).astype(np.float32)
    p = str(tmp_path / "m.exr")
    _tiled_exr(p, img, 32, 16, 0, mip_levels=True)
    back = host.read_exr(p)
    assert back.shape == (40, 64, 4) and np.array_equal(back[..., :3], img)


def test_exr_reader_refuses_absurd_headers(tmp_path):
    p = str(tmp_path / "a.exr")
    host.write_exr(p, np.zeros((2, 2, 3), np.float32))
    raw = bytearray(open(p, "rb").read())
    at = raw.index(b"dataWindow\0box2i\0") + len(b"dataWindow\0box2i\0") + 4
    for box in ((0, 0, 2 ** 31 - 1, 2 ** 31 - 1), (-2 ** 31, -2 ** 31, 2 ** 31 - 1, 1), (0, 0, 65535, 65535), (0, 0, 60000, 3)):
        raw[at:at + 16] = struct.pack("<iiii", *box)
        open(p, "wb").write(bytes(raw))
        with pytest.raises(IOError):
            host.read_exr(p)


def test_exr_reader_refuses_huge_tiles_and_truncated_channel_lists(tmp_path):
    # a tile size of 2^31 or more would turn negative in the block arithmetic; a channel list that ends in the middle of a
    # channel would leave a channel with made-up fields: both are refused with a message, no exception, no crash
    rng = np.random.default_rng(8)
    img = rng.uniform(0, 1, size=(20, 24, 3)).astype(np.float32)
    p = str(tmp_path / "t.exr")
    _tiled_exr(p, img, 16, 16, 0)
    raw = bytearray(open(p, "rb").read())
    at = raw.index(b"tiles\0tiledesc\0") + len(b"tiles\0tiledesc\0") + 4
    for tw, th in ((2 ** 31, 16), (16, 2 ** 32 - 1), (2 ** 31 + 5, 2 ** 31 + 5), (70000, 16)):
        raw[at:at + 8] = struct.pack("<II", tw, th)
        open(p, "wb").write(bytes(raw))
        with pytest.raises(IOError):
            host.read_exr(p)
    q = str(tmp_path / "c.exr")
    _plain_exr(q, {"R": img[..., 0], "G": img[..., 1], "B": img[..., 2]})
    raw = bytearray(open(q, "rb").read())
    at = raw.index(b"channels\0chlist\0") + len(b"channels\0chlist\0")
    size = struct.unpack("<I", raw[at:at + 4])[0]
    # the list loses its last 9 bytes (the end marker and half of the last channel) but keeps its place in the file
    raw[at:at + 4] = struct.pack("<I", size - 9)
    cut = bytes(raw[:at + 4 + size - 9]) + bytes(raw[at + 4 + size:])
    open(q, "wb").write(cut)
    with pytest.raises(IOError):
        host.read_exr(q)


def _plain_exr(path, planes):
    """An uncompressed FLOAT scanline file of the given channels ({name: [h, w] float32})."""
    names = sorted(planes)
    h, w = planes[names[0]].shape

    def attr(name, typ, val):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(val)) + val
    ch = b"".join(n.encode() + b"\0" + struct.pack("<IBBBBII", 2, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<iiii", 0, 0, w - 1, h - 1)
    hdr = struct.pack("<II", 20000630, 2) + attr("channels", "chlist", ch) + attr("compression", "compression", b"\0")
    hdr += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0")
    hdr += attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += attr("screenWindowCenter", "v2f", struct.pack("<ff", 0, 0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    hdr += b"\0"
    line = 8 + 4 * w * len(names)
    table = b"".join(struct.pack("<Q", len(hdr) + 8 * h + y * line) for y in range(h))
    rows = b"".join(struct.pack("<ii", y, line - 8) + b"".join(planes[n][y].astype("<f4").tobytes() for n in names)
                    for y in range(h))
    with open(path, "wb") as fh:
        fh.write(hdr + table + rows)


def test_exr_reader_grey_and_missing_channels(tmp_path):
    """tinyexr's LoadEXR (Scene.cpp:81): one channel of any name fills all four components; otherwise R, G and B
    must be there, further channels are skipped."""
    rng = np.random.default_rng(12)
    v = {n: rng.uniform(0, 1, size=(6, 9)).astype(np.float32) for n in ("R", "G", "B", "Y", "Z")}
    p = str(tmp_path / "g.exr")
    _plain_exr(p, {"Y": v["Y"]})
    assert np.array_equal(host.read_exr(p), np.repeat(v["Y"][..., None], 4, axis=-1))
    _plain_exr(p, {n: v[n] for n in ("B", "G", "R", "Z")})
    back = host.read_exr(p)
    assert np.array_equal(back[..., :3], np.stack([v["R"], v["G"], v["B"]], axis=-1)) and (back[..., 3] == 1).all()
    _plain_exr(p, {"R": v["R"], "G": v["G"]})
    with pytest.raises(IOError):
        host.read_exr(p)
    # version flags: long attribute names are fine, deep data and several parts are refused
    _plain_exr(p, {"Y": v["Y"]})
    raw = bytearray(open(p, "rb").read())
    for flag, ok in ((0x400, True), (0x800, False), (0x1000, False)):
        raw[4:8] = struct.pack("<I", 2 | flag)
        open(p, "wb").write(bytes(raw))
        if ok:
            assert np.array_equal(host.read_exr(p)[..., 0], v["Y"])
        else:
            with pytest.raises(IOError):
                host.read_exr(p)


def test_camera_matrix_matches_python_recipe():
    for t, r in (((0.0, 1.5, -3.0), (-np.pi / 4, 0.0)), ((0.3, 1.2, -2.5), (-0.5, 0.4))):
        assert np.allclose(host.camera_matrix(t, r), camera_matrix(t, r), atol=1e-7)
