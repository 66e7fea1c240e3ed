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


def textured_cubes(atlas_size=32):
    """The `cubes` scene with synthetic texture coordinates and tangents on every triangle and synthetic albedo /
    normal / ARM layers for material 2 (the ARM layer smaller than the atlas: it fills a corner only).
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
    return (models, shadings, lights, materials), {0: albedo, 1: nrm, 2: arm}, np.array([0, 0, 1, 0], np.int32)


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
