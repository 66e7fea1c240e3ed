"""Reader/writer of .rtscene files: the ordered TriangleModel / TriangleShading / Material arrays
that Scene::finalizeObjects (Scene.cpp:382-395 of the reference) uploads, stored little-endian as
magic "RTSC0001", uint32 count, lights, n_materials, reserved, then the three record arrays."""
import struct

import numpy as np

from .api import MATERIAL_DTYPE, TRI_MODEL_DTYPE, TRI_SHADING_DTYPE

MAGIC = b"RTSC0001"


def write_rtscene(path, models, shadings, lights, materials):
    with open(path, "wb") as fh:
        fh.write(MAGIC)
        fh.write(struct.pack("<IIII", len(models), int(lights), len(materials), 0))
        fh.write(np.ascontiguousarray(models, dtype=TRI_MODEL_DTYPE).tobytes())
        fh.write(np.ascontiguousarray(shadings, dtype=TRI_SHADING_DTYPE).tobytes())
        fh.write(np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE).tobytes())


def read_rtscene(path):
    """Returns (models, shadings, lights, materials)."""
    with open(path, "rb") as fh:
        data = fh.read()
    if data[:8] != MAGIC:
        raise ValueError("not an rtscene file: %s" % path)
    count, lights, n_mat, _ = struct.unpack_from("<IIII", data, 8)
    need = 24 + 148 * count + 48 * n_mat
    if len(data) < need:
        raise ValueError("truncated rtscene file: %s" % path)
    off = 24
    models = np.frombuffer(data, dtype=TRI_MODEL_DTYPE, count=count, offset=off).copy()
    off += 48 * count
    shadings = np.frombuffer(data, dtype=TRI_SHADING_DTYPE, count=count, offset=off).copy()
    off += 100 * count
    materials = np.frombuffer(data, dtype=MATERIAL_DTYPE, count=n_mat, offset=off).copy()
    return models, shadings, lights, materials
