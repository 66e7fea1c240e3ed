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
