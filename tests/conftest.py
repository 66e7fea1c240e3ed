import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(GOLDEN, "scenes")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def constant_env():
    env = np.zeros((4, 8, 4), dtype=np.float32)
    env[..., 0], env[..., 1], env[..., 2], env[..., 3] = 0.5, 0.7, 1.0, 1.0
    return env


def gradient_env(w=64, h=32):
    """Analytic environment that varies in both directions, to exercise the bilinear lookup."""
    y = (np.arange(h, dtype=np.float32) + 0.5) / h
    x = (np.arange(w, dtype=np.float32) + 0.5) / w
    env = np.zeros((h, w, 4), dtype=np.float32)
    env[..., 0] = 0.2 + 0.8 * y[:, None]
    env[..., 1] = 0.3 + 0.5 * np.abs(2.0 * x[None, :] - 1.0)
    env[..., 2] = 1.0 - 0.6 * y[:, None]
    env[..., 3] = 1.0
    return env.astype(np.float32)


def load_scene(name):
    from gpuraytracer_amd.scene_io import read_rtscene
    return read_rtscene(os.path.join(SCENES, name + ".rtscene"))


# Values at the edges of the GL_R11F_G11F_B10F upload conversion: zero, negative, below the smallest packed denormal,
# around the smallest packed normal (2^-14), around the largest packed values (65024 for the 6-bit mantissas, 64512 for
# the 5-bit one), far above them
PACK_EDGE_VALUES = (0.0, -1.0, -1e-3, 1e-8, 3e-5, 6.0e-5, 6.1e-5, 6.2e-5, 1.2e-4, 64512.0, 64513.0, 65024.0, 65025.0, 70000.0, 1e9)


def textured_cubes(atlas_size=32, edge_texels=False, nonfinite_texels=False):
    """The `cubes` scene with synthetic texture coordinates and tangents on every triangle and synthetic albedo /
    normal / ARM layers for material 2 (the ARM layer smaller than the atlas: it fills a corner only).
    edge_texels: every fifth texel of the albedo layer carries one of PACK_EDGE_VALUES in one channel;
    nonfinite_texels: a few texels carry Inf / NaN as well.
    Returns (scene tuple, layers {slot: rgba}, hasTexture flags)."""
    models, shadings, lights, materials = load_scene("cubes")
    rng = np.random.default_rng(5)
    S = atlas_size
    shadings = shadings.copy()
    shadings["tex_p"] = rng.uniform(0, 1, size=(len(shadings), 2)).astype(np.float32)
    shadings["tex_u"] = rng.uniform(-2, 2, size=(len(shadings), 2)).astype(np.float32)
    shadings["tex_v"] = rng.uniform(-2, 2, size=(len(shadings), 2)).astype(np.float32)
    t = rng.normal(size=(len(shadings), 3, 3)).astype(np.float32)
    shadings["tangents"] = t / np.linalg.norm(t, axis=2, keepdims=True)
    albedo = rng.uniform(0.1, 0.9, size=(S, S, 4)).astype(np.float32)
    nrm = rng.uniform(0.3, 0.7, size=(S, S, 4)).astype(np.float32)
    nrm[..., 2] = rng.uniform(0.8, 1.0, size=(S, S))
    arm = rng.uniform(0.05, 1.0, size=(S * 3 // 4, S * 5 // 8, 4)).astype(np.float32)
    if edge_texels:
        flat = albedo.reshape(-1, 4)
        idx = np.arange(0, len(flat), 5)
        flat[idx, idx % 3] = np.array(PACK_EDGE_VALUES, dtype=np.float32)[(idx // 5) % len(PACK_EDGE_VALUES)]
    if nonfinite_texels:
        flat = albedo.reshape(-1, 4)
        flat[7::97, 0] = np.inf
        flat[11::101, 1] = np.nan
        flat[13::103, 2] = -np.inf
    return (models, shadings, lights, materials), {0: albedo, 1: nrm, 2: arm}, np.array([0, 0, 1, 0], np.int32)


def tiny_cubes(scale):
    """The `cubes` scene scaled by a power of two so small that the triangle normals (pre-computed cross products of
    two edges, i.e. scaled by its square) fall among the denormal numbers.  All scalings are exact."""
    models, shadings, lights, materials = load_scene("cubes")
    scale = np.float32(scale)
    tiny = models.copy()
    for field in ("position", "u", "v"):
        tiny[field] = models[field] * scale
    tiny["true_normal"] = models["true_normal"] * scale * scale
    return tiny, shadings, lights, materials


@pytest.fixture(scope="session")
def cubes():
    return load_scene("cubes")


@pytest.fixture(scope="session")
def obiwan():
    return load_scene("_ObiWan")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gpu_ctx():
    import gpuraytracer_amd as g
    ctx = g.Context(0)
    yield ctx
    ctx.close()


class HipRuntime:
    """The few HIP runtime calls some GPU tests need (streams, events, graph capture), bound with ctypes to the runtime
    library the HIP extension itself is linked against, so that the test and the extension share one runtime."""

    NOT_READY = 600          # hipErrorNotReady

    def __init__(self):
        import ctypes
        self.c = ctypes
        self.lib = ctypes.CDLL("libamdhip64.so.7")
        vp = ctypes.c_void_p
        self.lib.hipStreamCreate.argtypes = [ctypes.POINTER(vp)]
        self.lib.hipStreamDestroy.argtypes = [vp]
        self.lib.hipStreamSynchronize.argtypes = [vp]
        self.lib.hipEventCreate.argtypes = [ctypes.POINTER(vp)]
        self.lib.hipEventRecord.argtypes = [vp, vp]
        self.lib.hipEventQuery.argtypes = [vp]
        self.lib.hipEventDestroy.argtypes = [vp]
        self.lib.hipStreamBeginCapture.argtypes = [vp, ctypes.c_int]
        self.lib.hipStreamEndCapture.argtypes = [vp, ctypes.POINTER(vp)]
        self.lib.hipGraphInstantiate.argtypes = [ctypes.POINTER(vp), vp, vp, vp, ctypes.c_size_t]
        self.lib.hipGraphLaunch.argtypes = [vp, vp]
        self.lib.hipGraphExecDestroy.argtypes = [vp]
        self.lib.hipGraphDestroy.argtypes = [vp]

    def check(self, rc):
        assert rc == 0, "HIP call failed with code %d" % rc

    def stream(self):
        h = self.c.c_void_p()
        self.check(self.lib.hipStreamCreate(self.c.byref(h)))
        return h

    def event(self):
        h = self.c.c_void_p()
        self.check(self.lib.hipEventCreate(self.c.byref(h)))
        return h
