"""ctypes access to the CPU-only entry points of libgpurt_host.so (the C++ host layer in csrc/host/:
Wavefront ingest with the reference's conventions, EXR reader/writer, camera matrix)."""
import ctypes
import os

import numpy as np

from .api import MATERIAL_DTYPE, TRI_MODEL_DTYPE, TRI_SHADING_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgpurt_host.so")
        if not os.path.exists(path):
            raise ImportError("%s is missing: run __graft_entry__.build()" % path)
        lib = ctypes.CDLL(path)
        c = ctypes
        lib.rth_load_wavefront.argtypes = [c.c_char_p, c.POINTER(c.c_void_p), c.POINTER(c.c_void_p),
                                           c.POINTER(c.c_uint32), c.POINTER(c.c_uint32), c.POINTER(c.c_void_p),
                                           c.POINTER(c.c_uint32)]
        lib.rth_free.argtypes = [c.c_void_p]
        lib.rth_write_exr.argtypes = [c.c_char_p, c.c_int, c.c_int, c.c_int, c.c_void_p]
        lib.rth_write_exr_compressed.argtypes = [c.c_char_p, c.c_int, c.c_int, c.c_int, c.c_void_p, c.c_int]
        lib.rth_read_exr.argtypes = [c.c_char_p, c.POINTER(c.c_int), c.POINTER(c.c_int), c.POINTER(c.c_void_p)]
        lib.rth_camera_matrix.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p]
        _LIB = lib
    return _LIB


def load_wavefront(base_path):
    """Scene::addWavefrontModel + the buffer construction of Scene::finalizeObjects.
    Returns (models, shadings, lights, materials)."""
    lib = _lib()
    pm, ps, pmat = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    count, lights, n_mat = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
    rc = lib.rth_load_wavefront(base_path.encode(), ctypes.byref(pm), ctypes.byref(ps), ctypes.byref(count),
                                ctypes.byref(lights), ctypes.byref(pmat), ctypes.byref(n_mat))
    if rc != 0:
        raise ValueError("cannot load %s (.obj/.mtl), code %d" % (base_path, rc))
    try:
        models = np.frombuffer(ctypes.string_at(pm, 48 * count.value), dtype=TRI_MODEL_DTYPE).copy()
        shadings = np.frombuffer(ctypes.string_at(ps, 100 * count.value), dtype=TRI_SHADING_DTYPE).copy()
        materials = np.frombuffer(ctypes.string_at(pmat, 48 * n_mat.value), dtype=MATERIAL_DTYPE).copy()
    finally:
        lib.rth_free(pm), lib.rth_free(ps), lib.rth_free(pmat)
    return models, shadings, int(lights.value), materials


def write_exr(path, pixels, compression=0):
    """compression: 0 = none, 3 = ZIP, -1 = ZIP unless both sides are under 16 pixels (what the host program writes)."""
    pixels = np.ascontiguousarray(pixels, dtype=np.float32)
    h, w, ch = pixels.shape
    if _lib().rth_write_exr_compressed(path.encode(), w, h, ch, pixels.ctypes.data_as(ctypes.c_void_p), compression) != 0:
        raise IOError("cannot write " + path)


def read_exr(path):
    """Returns RGBA float32 [h, w, 4], row 0 = first scanline."""
    lib = _lib()
    w, h, p = ctypes.c_int(), ctypes.c_int(), ctypes.c_void_p()
    if lib.rth_read_exr(path.encode(), ctypes.byref(w), ctypes.byref(h), ctypes.byref(p)) != 0:
        raise IOError("cannot read " + path)
    try:
        out = np.frombuffer(ctypes.string_at(p, 16 * w.value * h.value), dtype=np.float32).reshape(h.value, w.value, 4).copy()
    finally:
        lib.rth_free(p)
    return out


def camera_matrix(translation, rotation):
    t = (ctypes.c_double * 3)(*translation)
    r = (ctypes.c_double * 2)(*rotation)
    out = (ctypes.c_float * 16)()
    _lib().rth_camera_matrix(t, r, out)
    return np.array(out[:], dtype=np.float32)
