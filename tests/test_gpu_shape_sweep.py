"""Boundary shapes of every numeric entry point of the C ABI against the oracle (tools/shape_sweep.py): vectSize 1 .. 81 around every
multiple of 4 / 16 / 32 / 64 and the MFMA instantiation bounds (16, 32, 60, 64, 80), 1 .. 65 (and 2048, 4097) Gaussians, 1 .. 257 frames,
ranks 1 .. 129 around the 16 / 32-wide tiles of the Cholesky family, one-utterance / one-speaker / one-trial calls, float32 and float64
features.  Round 6 added it after ONE fixture that happened to have vectSize 1 uncovered a shape on which every statistic was wrong."""
import os
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("which,at_least", [("gmm", 1500), ("gmm_paths", 1400), ("tv", 1500), ("score", 500), ("backend", 100)])
def test_boundary_shapes_match_the_oracle(which, at_least):
    import shape_sweep
    n, failed = shape_sweep.run([which])
    assert n >= at_least and not failed, failed[:10]
