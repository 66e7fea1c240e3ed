"""GPU test of rt_read_sum_static, the read-back entry point that allocates nothing: same bits as rt_read_sum on
two calls in a row (the second on a smaller target, so stale words of the first call sit behind its image in the
shared buffer), and the documented size limit."""
import numpy as np
import pytest

from conftest import constant_env, load_scene
from gpuraytracer_amd.camera import camera_matrix, DEFAULT_CLOCK

pytestmark = pytest.mark.gpu


def test_read_sum_static_equals_read_sum():
    import gpuraytracer_amd as g
    cubes = load_scene("cubes")
    ctx = g.Context(0)
    try:
        ctx.upload_scene(*cubes)
        ctx.upload_env(constant_env())
        ctx.set_frame(camera_matrix(), DEFAULT_CLOCK, 6, 1.0)
        for (w, h, spp) in ((96, 64, 5), (40, 24, 3)):       # the second call reads a smaller image
            ctx.resize(w, h)
            ctx.trace(0, spp, 1)
            want, n_want = ctx.read_sum()
            got, n_got = ctx.read_sum_static()
            assert n_got == n_want == spp
            assert got.shape == (h, w, 3) and np.array_equal(got.view(np.uint32), want.view(np.uint32))
            assert np.abs(got).max() > 0
        # a second context on the same GPU uses the same array, one call after the other
        other = g.Context(0)
        try:
            other.upload_scene(*cubes)
            other.upload_env(constant_env())
            other.set_frame(camera_matrix(), 77, 4, 1.0)
            other.resize(56, 16)
            other.trace(2, 2, 1)
            a, _ = other.read_sum()
            b, _ = other.read_sum_static()
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        finally:
            other.close()
        again, _ = ctx.read_sum_static()
        assert np.array_equal(again.view(np.uint32), want.view(np.uint32))
        # beyond the built-in buffer: a clear error, and rt_read_sum still serves that size
        ctx.resize(2048, 1366)
        with pytest.raises(g.GpuRtError) as err:
            ctx.read_sum_static()
        assert err.value.code == -5 and "2796202" in str(err.value)
        big, _ = ctx.read_sum()
        assert big.shape == (1366, 2048, 3) and not big.any()
        ctx.resize(1920, 1080)
        ctx.trace(0, 1, 1)
        a, _ = ctx.read_sum()
        b, _ = ctx.read_sum_static()
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    finally:
        ctx.close()
