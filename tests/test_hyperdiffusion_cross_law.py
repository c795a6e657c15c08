"""DryBiharmonic of the dry atmosphere against the (reference-pinned) scalar hyperdiffusion law:
see tests/hs_crosslaw.py.  CPU: the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import cm  # noqa: E402
from hs_crosslaw import crosslaw_residual  # noqa: E402


def test_atmos_hyperdiffusion_equals_scalar_law_sum_oracle():
    from oracle import oracle as O
    O.build()

    def make_dg(law, grid, d, dd, nf):
        return O.OracleDGModel(law, grid, nf_first=nf, direction=d, diffusion_direction=dd)
    res, cond = crosslaw_residual(cm, make_dg)
    assert res[0] == 0.0                                    # no hyperdiffusive mass flux
    for s in range(1, 5):
        # the hyperdiffusive part must be there at all (this check once passed vacuously with a
        # horizontal length scale of 1e-5 m) ...
        assert cond[s] < 1e7, (s, cond)
        # ... and equal the scalar-law sum to the rounding of the two full tendencies it is the
        # difference of (observed: 3e-11 ... 8e-11 at cond = 2e5)
        assert res[s] < 2e-15 * cond[s], (s, res, cond)
