"""A slice of the soak run (tools/soak_parity.py) inside the GPU tests: random triangle soups, bundled scenes with
random cameras, cameras with exact zeros in the matrix and soups of 30 000 to 160 000 triangles; the library's default
forms and three other hierarchy forms against the loops over all triangles (the reference's own algorithm), bit for bit
with equal counters, every fourth scene also against the CPU checker.  SEED_BASE is moved whenever the kernels change
in a larger way, so that the slice does not keep testing the scenes the code was last debugged on."""
import os
import sys

import pytest

from conftest import ROOT, gradient_env

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(ROOT, "tools"))

SEED_BASE = 53000        # >= 10000: all five families of soak_parity.case_for, by seed % 5
N_SEEDS = 50


@pytest.mark.parametrize("seed", range(SEED_BASE, SEED_BASE + N_SEEDS), ids=lambda s: "base%d-seed%d" % (SEED_BASE, s))
def test_soak_slice(gpu_ctx, oracle_mod, tmp_path, seed):
    import soak_parity
    import test_gpu_parity as T
    line, diffs = soak_parity.check_seed(gpu_ctx, seed, str(tmp_path), gradient_env(), T, oracle_mod, oracle_budget=6e8)
    assert not diffs, line + " " + " ".join(diffs)
