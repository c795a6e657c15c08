"""Different polynomial orders in the horizontal and the vertical (polynomialorder = (N_h, N_v))
against the reference's stored errors: variable_degree_advection_diffusion.jl, dim = 3,
orders (4,2) and (2,4), level 1 in the oracle (the GPU runs levels 1-3 in
tests/test_gpu_variable_degree.py)."""
import json
import os

import numpy as np
import pytest

from helpers import variable_degree_setup

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_values.json")))[
    "variable_degree_advection_diffusion"]


def expected(orders, field, level):
    deg = orders[0] if field == "horizontal" else orders[1]
    return GOLD[field + "_field"][str(deg)][level - 1]


@pytest.mark.parametrize("orders", [(4, 2), (2, 4)])
@pytest.mark.parametrize("field", ["horizontal", "vertical"])
def test_variable_degree_matches_reference_error(oracle, orders, field):
    law, grid, dt = variable_degree_setup(1, orders, field)
    dg = oracle.OracleDGModel(law, grid)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    oracle.solve(dg, Q, dt, 1.0)
    Qe = law.init_state_prognostic(grid, dg.state_auxiliary, 1.0)
    err = np.sqrt(oracle.weighted_norm2_local(grid, Q, Qe))
    exp = expected(orders, field, 1)
    assert abs(err - exp) <= GOLD["rtol"] * exp, (err, exp)
