"""ctypes binding of the CPU checker (oracle/rt_oracle.cpp).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build():
    """Compile both back ends with the Makefile in this directory."""
    subprocess.check_call(["make", "-s", "-C", HERE])


def _load(libm):
    key = bool(libm)
    if key in _LIBS:
        return _LIBS[key]
    name = "librt_oracle_libm.so" if libm else "librt_oracle.so"
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        build()
    lib = ctypes.CDLL(path)
    c = ctypes
    fp = c.POINTER(c.c_float)
    lib.orc_set_flush_denormals.argtypes = [c.c_int]
    lib.orc_pcg_hash.restype = c.c_uint32
    lib.orc_pcg_hash.argtypes = [c.c_uint32]
    lib.orc_unit_float.restype = c.c_float
    lib.orc_unit_float.argtypes = [c.POINTER(c.c_uint32)]
    lib.orc_random_sphere.argtypes = [c.POINTER(c.c_uint32), fp]
    lib.orc_pack_r11g11b10.restype = c.c_uint32
    lib.orc_pack_r11g11b10.argtypes = [c.c_float] * 3
    lib.orc_unpack_r11g11b10.argtypes = [c.c_uint32, fp]
    lib.orc_math.restype = c.c_float
    lib.orc_math.argtypes = [c.c_int, c.c_float, c.c_float]
    lib.orc_math_batch.argtypes = [c.c_int, c.c_void_p, c.c_void_p, c.c_uint64, c.c_void_p]
    lib.orc_sample_env.argtypes = [c.c_void_p, c.c_void_p, c.c_uint64, c.c_void_p]
    lib.orc_irradiance.argtypes = [c.c_void_p, c.c_uint32, c.c_uint32, c.c_void_p]
    lib.orc_scene_create.restype = c.c_void_p
    lib.orc_scene_create.argtypes = [c.c_void_p, c.c_void_p, c.c_uint32, c.c_uint32, c.c_void_p, c.c_uint32]
    lib.orc_scene_destroy.argtypes = [c.c_void_p]
    lib.orc_scene_set_atlas_size.argtypes = [c.c_void_p, c.c_uint32]
    lib.orc_scene_upload_atlas_layer.argtypes = [c.c_void_p, c.c_uint32, c.c_uint32, c.c_uint32, c.c_uint32,
                                                 c.c_void_p]
    lib.orc_scene_set_has_texture.argtypes = [c.c_void_p, c.c_void_p, c.c_uint32]
    lib.orc_scene_upload_env.argtypes = [c.c_void_p, c.c_uint32, c.c_uint32, c.c_void_p]
    lib.orc_closest_hit_batch.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_uint64, c.c_void_p, c.c_void_p]
    lib.orc_shadow_batch.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_uint64,
                                     c.c_void_p]
    lib.orc_pixel_samples.argtypes = [c.c_void_p, c.c_void_p, c.c_uint32, c.c_int32, c.c_float, c.c_uint32,
                                      c.c_uint32, c.c_void_p, c.c_void_p, c.c_void_p, c.c_uint64, c.c_void_p]
    lib.orc_render.argtypes = [c.c_void_p, c.c_void_p, c.c_uint32, c.c_int32, c.c_float, c.c_uint32, c.c_uint32,
                               c.c_uint32, c.c_uint32, c.c_uint32, c.c_int, c.c_void_p, c.c_void_p]
    lib.orc_linear_to_srgb.restype = c.c_float
    lib.orc_linear_to_srgb.argtypes = [c.c_float]
    _LIBS[key] = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def set_flush_denormals(on, libm=False):
    """on: the batch entry points read denormal inputs as zero and flush denormal results to zero, as llvmpipe's
    compiled shaders do (off by default: the product keeps denormal numbers)."""
    _load(libm).orc_set_flush_denormals(1 if on else 0)


def pcg_hash(k, libm=False):
    return int(_load(libm).orc_pcg_hash(ctypes.c_uint32(k & 0xFFFFFFFF)))


def unit_float(seed, libm=False):
    """Returns (value, new_seed)."""
    s = ctypes.c_uint32(seed & 0xFFFFFFFF)
    v = _load(libm).orc_unit_float(ctypes.byref(s))
    return float(v), int(s.value)


def random_sphere(seed, libm=False):
    s = ctypes.c_uint32(seed & 0xFFFFFFFF)
    out = (ctypes.c_float * 3)()
    _load(libm).orc_random_sphere(ctypes.byref(s), out)
    return np.array(out[:], dtype=np.float32), int(s.value)


_MATH_CODES = {"sin": 0, "cos": 1, "tan": 2, "acos": 3, "asin": 4, "atan2": 5, "pow5": 6}


def math_fn(which, x, y=0.0, libm=False):
    return np.float32(_load(libm).orc_math(_MATH_CODES[which], ctypes.c_float(x), ctypes.c_float(y)))


def irradiance(env_rgba, libm=False):
    """Irradiance pre-filter (Scene.cpp:150-186 + irradiance.glsl): RGBA float32 (h, w, 4) in, RGB (h//4, w//4, 3) out."""
    env = np.ascontiguousarray(env_rgba, dtype=np.float32)
    h, w = env.shape[:2]
    out = np.zeros((h // 4, w // 4, 3), dtype=np.float32)
    _load(libm).orc_irradiance(_ptr(env), w, h, _ptr(out))
    return out


def math_batch(which, x, y=None, libm=False):
    """Vector form of math_fn: float32 arrays in, float32 array out."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    yy = None if y is None else np.ascontiguousarray(np.broadcast_to(np.asarray(y, dtype=np.float32), x.shape))
    out = np.empty_like(x)
    _load(libm).orc_math_batch(_MATH_CODES[which], _ptr(x), None if yy is None else _ptr(yy), x.size, _ptr(out))
    return out


def pack_r11g11b10(r, g, b):
    return int(_load(False).orc_pack_r11g11b10(r, g, b))


def unpack_r11g11b10(p):
    out = (ctypes.c_float * 3)()
    _load(False).orc_unpack_r11g11b10(ctypes.c_uint32(p), out)
    return np.array(out[:], dtype=np.float32)


def linear_to_srgb(c):
    return float(_load(True).orc_linear_to_srgb(ctypes.c_float(c)))


class OracleScene:
    """Holds a scene inside the checker library; mirrors the uploads of the C ABI under test."""

    def __init__(self, models, shadings, lights, materials, libm=False):
        self.lib = _load(libm)
        self.count = len(models)
        self.lights = int(lights)
        models = np.ascontiguousarray(models)
        shadings = np.ascontiguousarray(shadings)
        materials = np.ascontiguousarray(materials)
        assert models.dtype.itemsize == 48 and shadings.dtype.itemsize == 100 and materials.dtype.itemsize == 48
        self.h = self.lib.orc_scene_create(_ptr(models), _ptr(shadings), self.count, self.lights, _ptr(materials),
                                           len(materials))

    def close(self):
        if self.h:
            self.lib.orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_atlas_size(self, size):
        self.lib.orc_scene_set_atlas_size(self.h, size)

    def upload_atlas_layer(self, material, slot, rgba):
        rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        h, w = rgba.shape[:2]
        rc = self.lib.orc_scene_upload_atlas_layer(self.h, material, slot, w, h, _ptr(rgba))
        if rc:
            raise ValueError("bad atlas layer")

    def set_has_texture(self, flags):
        flags = np.ascontiguousarray(flags, dtype=np.int32)
        self.lib.orc_scene_set_has_texture(self.h, _ptr(flags), len(flags))

    def upload_env(self, rgba):
        rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        h, w = rgba.shape[:2]
        self.lib.orc_scene_upload_env(self.h, w, h, _ptr(rgba))

    def sample_env(self, st):
        """Bilinear environment lookups: st[n, 2] -> rgb[n, 3]."""
        st = np.ascontiguousarray(st, dtype=np.float32)
        rgb = np.empty((len(st), 3), dtype=np.float32)
        self.lib.orc_sample_env(self.h, _ptr(st), len(st), _ptr(rgb))
        return rgb

    def closest_hit_batch(self, o, d):
        o = np.ascontiguousarray(o, dtype=np.float32)
        d = np.ascontiguousarray(d, dtype=np.float32)
        n = len(o)
        ids = np.empty(n, dtype=np.int32)
        uvt = np.empty((n, 3), dtype=np.float32)
        self.lib.orc_closest_hit_batch(self.h, _ptr(o), _ptr(d), n, _ptr(ids), _ptr(uvt))
        return ids, uvt

    def shadow_batch(self, o, d, max_t, current):
        o = np.ascontiguousarray(o, dtype=np.float32)
        d = np.ascontiguousarray(d, dtype=np.float32)
        max_t = np.ascontiguousarray(max_t, dtype=np.float32)
        current = np.ascontiguousarray(current, dtype=np.int32)
        occ = np.empty(len(o), dtype=np.int32)
        self.lib.orc_shadow_batch(self.h, _ptr(o), _ptr(d), _ptr(max_t), _ptr(current), len(o), _ptr(occ))
        return occ

    def pixel_samples(self, camera, clock, recursion, exposure, width, height, xs, ys, samples):
        camera = np.ascontiguousarray(camera, dtype=np.float32).reshape(16)
        xs = np.ascontiguousarray(xs, dtype=np.uint32)
        ys = np.ascontiguousarray(ys, dtype=np.uint32)
        samples = np.ascontiguousarray(samples, dtype=np.uint32)
        rgb = np.empty((len(xs), 3), dtype=np.float32)
        self.lib.orc_pixel_samples(self.h, _ptr(camera), clock, recursion, exposure, width, height, _ptr(xs),
                                   _ptr(ys), _ptr(samples), len(xs), _ptr(rgb))
        return rgb

    def render(self, camera, clock, recursion, exposure, width, height, sample_begin, sample_count,
               sample_stride=1, mode="sum", rgb=None):
        """Returns (rgb[h, w, 3], stats dict).  mode 'sum' adds to `rgb`; 'mean' is the reference's running mean."""
        camera = np.ascontiguousarray(camera, dtype=np.float32).reshape(16)
        if rgb is None:
            rgb = np.zeros((height, width, 3), dtype=np.float32)
        assert rgb.flags["C_CONTIGUOUS"] and rgb.dtype == np.float32
        stats = np.zeros(6, dtype=np.uint64)
        self.lib.orc_render(self.h, _ptr(camera), clock, recursion, exposure, width, height, sample_begin,
                            sample_count, sample_stride, 0 if mode == "sum" else 1, _ptr(rgb), _ptr(stats))
        keys = ["pixel_samples", "rays", "shadow_probes", "glossy_probes", "closest_tests", "probe_tests"]
        return rgb, dict(zip(keys, (int(v) for v in stats)))
