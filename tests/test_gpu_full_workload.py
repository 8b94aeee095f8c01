"""BASELINE.json configs[1] and configs[2] at full size, every voxel and every pixel against the oracle (the C oracle
fills 512^3 in about a second on the box's cores): textures and the pre-shading march record bit for bit, RGBA within
the north-star tolerance of 1e-4."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RGBA_TOL = 1e-4


@pytest.mark.parametrize("side", [256, 512])
def test_whole_grid_and_whole_frame(pkg, oracle, side):
    spec = importlib.util.spec_from_file_location("full_parity", os.path.join(ROOT, "tools", "full_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad_words, aux_diff, rgba_err = mod.check(side, log=lambda m: None)
    assert bad_words == 0
    assert all(v == 0 for v in aux_diff.values()), aux_diff
    assert rgba_err <= RGBA_TOL
