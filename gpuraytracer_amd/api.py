"""ctypes binding of the C ABI declared in include/gpurt.h."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

TRI_MODEL_DTYPE = np.dtype([("true_normal", "<f4", 3), ("position", "<f4", 3), ("u", "<f4", 3), ("v", "<f4", 3)])
TRI_SHADING_DTYPE = np.dtype(
    [("material_id", "<u4"), ("normals", "<f4", (3, 3)), ("tangents", "<f4", (3, 3)),
     ("tex_p", "<f4", 2), ("tex_u", "<f4", 2), ("tex_v", "<f4", 2)])
MATERIAL_DTYPE = np.dtype([("albedo", "<f4", 4), ("specular_roughness", "<f4", 4), ("emission_ior", "<f4", 4)])

EXPORTS = [
    "rt_create", "rt_destroy", "rt_last_error", "rt_set_stream", "rt_synchronize", "rt_upload_scene",
    "rt_set_atlas_size", "rt_upload_atlas_layer", "rt_set_has_texture", "rt_upload_env", "rt_prefilter_irradiance", "rt_resize",
    "rt_set_sum_buffer", "rt_set_frame", "rt_trace", "rt_read_sum", "rt_read_sum_static", "rt_closest_hit_batch",
    "rt_closest_hit_batch_device", "rt_read_srgb8", "rt_get_stats", "rt_reset_stats", "rt_set_option", "rt_accumulate_from",
    "rt_select_target", "rt_clear",
]


class RtStats(ctypes.Structure):
    _fields_ = [("pixel_samples", ctypes.c_uint64), ("rays", ctypes.c_uint64), ("shadow_probes", ctypes.c_uint64),
                ("glossy_probes", ctypes.c_uint64), ("closest_tests", ctypes.c_uint64),
                ("probe_tests", ctypes.c_uint64), ("trace_ms", ctypes.c_double), ("extend_ms", ctypes.c_double),
                ("probe_ms", ctypes.c_double), ("shade_ms", ctypes.c_double), ("extend_launches", ctypes.c_uint64),
                ("probe_launches", ctypes.c_uint64), ("extend0_ms", ctypes.c_double),
                ("extend0_launches", ctypes.c_uint64), ("bvh_extend_nodes", ctypes.c_uint64),
                ("bvh_extend_tris", ctypes.c_uint64), ("bvh_probe_nodes", ctypes.c_uint64),
                ("bvh_probe_tris", ctypes.c_uint64), ("depth1_paths", ctypes.c_uint64), ("queue_bytes", ctypes.c_uint64),
                ("deferred_probes", ctypes.c_uint64), ("parked_probes", ctypes.c_uint64),
                ("skipped_samples", ctypes.c_uint64)]


class GpuRtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("gpurt error %d: %s" % (code, message))
        self.code = code


def lib_path():
    """Path of the HIP library; GPURT_LIB selects another build of it (used for A/B timing of variants)."""
    return os.environ.get("GPURT_LIB") or os.path.join(_HERE, "libgpurt.so")


_LIB = None


def load_library():
    """Loads libgpurt.so; there is no fallback path when it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError("%s is missing: build it with __graft_entry__.build() (hipcc, gfx950)" % path)
    lib = ctypes.CDLL(path)
    c = ctypes
    vp = c.c_void_p
    lib.rt_create.argtypes = [c.c_int, c.POINTER(vp)]
    lib.rt_destroy.argtypes = [vp]
    lib.rt_last_error.argtypes = [vp]
    lib.rt_last_error.restype = c.c_char_p
    lib.rt_set_stream.argtypes = [vp, vp]
    lib.rt_synchronize.argtypes = [vp]
    lib.rt_upload_scene.argtypes = [vp, vp, vp, c.c_uint32, c.c_uint32, vp, c.c_uint32]
    lib.rt_set_atlas_size.argtypes = [vp, c.c_uint32]
    lib.rt_upload_atlas_layer.argtypes = [vp, c.c_uint32, c.c_uint32, c.c_uint32, c.c_uint32, vp]
    lib.rt_set_has_texture.argtypes = [vp, vp, c.c_uint32]
    lib.rt_upload_env.argtypes = [vp, c.c_uint32, c.c_uint32, vp]
    lib.rt_prefilter_irradiance.argtypes = [vp, vp, c.POINTER(c.c_uint32), c.POINTER(c.c_uint32)]
    lib.rt_resize.argtypes = [vp, c.c_uint32, c.c_uint32]
    lib.rt_set_sum_buffer.argtypes = [vp, vp]
    lib.rt_select_target.argtypes = [vp, c.c_int]
    lib.rt_clear.argtypes = [vp]
    lib.rt_set_frame.argtypes = [vp, vp, c.c_uint32, c.c_int32, c.c_float]
    lib.rt_trace.argtypes = [vp, c.c_uint32, c.c_uint32, c.c_uint32]
    lib.rt_read_sum.argtypes = [vp, vp, c.POINTER(c.c_uint64)]
    lib.rt_read_sum_static.argtypes = [vp, vp, c.POINTER(c.c_uint64)]
    lib.rt_accumulate_from.argtypes = [vp, vp]
    lib.rt_closest_hit_batch.argtypes = [vp, vp, vp, c.c_uint64, vp, vp]
    lib.rt_closest_hit_batch_device.argtypes = [vp, vp, vp, c.c_uint64, vp, vp]
    lib.rt_read_srgb8.argtypes = [vp, vp]
    lib.rt_get_stats.argtypes = [vp, c.POINTER(RtStats)]
    lib.rt_reset_stats.argtypes = [vp]
    lib.rt_set_option.argtypes = [vp, c.c_char_p, c.c_int64]
    for name in EXPORTS:
        if name != "rt_last_error":
            getattr(lib, name).restype = c.c_int
    _LIB = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Context:
    """One rendering context on one GPU (rt_context of the C ABI)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        rc = self.lib.rt_create(device, ctypes.byref(h))
        if rc != 0:
            raise GpuRtError(rc, "rt_create(device=%d) failed" % device)
        self.h = h
        self.width = self.height = 0
        self._parked_size = (0, 0)
        self.target = 0

    def _check(self, rc):
        if rc != 0:
            raise GpuRtError(rc, self.lib.rt_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.rt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_option(self, name, value):
        self._check(self.lib.rt_set_option(self.h, name.encode(), int(value)))

    def set_stream(self, stream_ptr):
        self._check(self.lib.rt_set_stream(self.h, ctypes.c_void_p(stream_ptr)))

    def synchronize(self):
        self._check(self.lib.rt_synchronize(self.h))

    def upload_scene(self, models, shadings, lights, materials):
        models = np.ascontiguousarray(models, dtype=TRI_MODEL_DTYPE)
        shadings = np.ascontiguousarray(shadings, dtype=TRI_SHADING_DTYPE)
        materials = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        if len(models) != len(shadings):
            raise ValueError("models and shadings differ in length")
        self._check(self.lib.rt_upload_scene(self.h, _ptr(models), _ptr(shadings), len(models), int(lights),
                                             _ptr(materials), len(materials)))

    def set_atlas_size(self, size):
        self._check(self.lib.rt_set_atlas_size(self.h, size))

    def upload_atlas_layer(self, material, slot, rgba):
        rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        if rgba.ndim != 3 or rgba.shape[2] != 4:
            raise ValueError("rgba must be [h, w, 4]")
        self._check(self.lib.rt_upload_atlas_layer(self.h, material, slot, rgba.shape[1], rgba.shape[0], _ptr(rgba)))

    def set_has_texture(self, flags):
        flags = np.ascontiguousarray(flags, dtype=np.int32)
        self._check(self.lib.rt_set_has_texture(self.h, _ptr(flags), len(flags)))

    def upload_env(self, rgba):
        rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        if rgba.ndim != 3 or rgba.shape[2] != 4:
            raise ValueError("rgba must be [h, w, 4]")
        self._check(self.lib.rt_upload_env(self.h, rgba.shape[1], rgba.shape[0], _ptr(rgba)))

    def prefilter_irradiance(self):
        """Scene.cpp:150-186: the (h//4, w//4, 3) irradiance image of the uploaded environment."""
        w, h = ctypes.c_uint32(0), ctypes.c_uint32(0)
        self._check(self.lib.rt_prefilter_irradiance(self.h, None, ctypes.byref(w), ctypes.byref(h)))
        out = np.zeros((h.value, w.value, 3), dtype=np.float32)
        self._check(self.lib.rt_prefilter_irradiance(self.h, _ptr(out), ctypes.byref(w), ctypes.byref(h)))
        return out

    def resize(self, width, height):
        self._check(self.lib.rt_resize(self.h, width, height))
        self.width, self.height = width, height

    def select_target(self, target):
        """0 = the still image, 1 = the low-resolution preview (Scene::prepare's choice of render target)."""
        self._check(self.lib.rt_select_target(self.h, int(target)))
        if int(target) != self.target:
            (self.width, self.height), self._parked_size = self._parked_size, (self.width, self.height)
            self.target = int(target)

    def clear(self):
        """The selected target starts over (what SAMPLE 0 does to the reference's running mean)."""
        self._check(self.lib.rt_clear(self.h))

    def set_sum_buffer(self, device_ptr):
        self._check(self.lib.rt_set_sum_buffer(self.h, ctypes.c_void_p(device_ptr)))

    def set_frame(self, camera, clock, recursion=6, exposure=1.0):
        camera = np.ascontiguousarray(camera, dtype=np.float32).reshape(16)
        self._check(self.lib.rt_set_frame(self.h, _ptr(camera), clock, recursion, exposure))

    def trace(self, sample_begin, sample_count, sample_stride=1):
        self._check(self.lib.rt_trace(self.h, sample_begin, sample_count, sample_stride))

    def read_sum(self):
        """Returns (rgb[h, w, 3] float32 sums, n_samples)."""
        rgb = np.empty((self.height, self.width, 3), dtype=np.float32)
        n = ctypes.c_uint64()
        self._check(self.lib.rt_read_sum(self.h, _ptr(rgb), ctypes.byref(n)))
        return rgb, int(n.value)

    def accumulate_from(self, other):
        """Adds another context's sums and sample count to this one's (same target size; any GPU of the node)."""
        self._check(self.lib.rt_accumulate_from(self.h, other.h))

    def read_sum_static(self):
        """read_sum through the library's built-in staging buffer (no device allocation)."""
        rgb = np.empty((self.height, self.width, 3), dtype=np.float32)
        n = ctypes.c_uint64()
        self._check(self.lib.rt_read_sum_static(self.h, _ptr(rgb), ctypes.byref(n)))
        return rgb, int(n.value)

    def read_srgb8(self):
        rgb = np.empty((self.height, self.width, 3), dtype=np.uint8)
        self._check(self.lib.rt_read_srgb8(self.h, _ptr(rgb)))
        return rgb

    def closest_hit_batch(self, o, d):
        o = np.ascontiguousarray(o, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, dtype=np.float32).reshape(-1, 3)
        if len(o) != len(d):
            raise ValueError("origins and directions differ in length")
        ids = np.empty(len(o), dtype=np.int32)
        uvt = np.empty((len(o), 3), dtype=np.float32)
        self._check(self.lib.rt_closest_hit_batch(self.h, _ptr(o), _ptr(d), len(o), _ptr(ids), _ptr(uvt)))
        return ids, uvt

    def closest_hit_batch_device(self, o_ptr, d_ptr, n, id_ptr, uvt_ptr):
        vp = ctypes.c_void_p
        self._check(self.lib.rt_closest_hit_batch_device(self.h, vp(o_ptr), vp(d_ptr), n, vp(id_ptr), vp(uvt_ptr)))

    def stats(self):
        s = RtStats()
        self._check(self.lib.rt_get_stats(self.h, ctypes.byref(s)))
        return {k: getattr(s, k) for k, _ in RtStats._fields_}

    def reset_stats(self):
        self._check(self.lib.rt_reset_stats(self.h))
