"""The reference's dim = 2 golden rows (pseudo1D_advection_diffusion.jl:242-253,
periodic_3D_hyperdiffusion.jl:231-239) on the 3-D restatement: the 2-D problem is run as the
y-invariant slice of a 3-D one (helpers.pseudo1d_dim2_setup), see DESIGN.md "dim = 2"."""
import json
import os

import numpy as np
import pytest

from helpers import periodic_hyperdiffusion_dim2_setup, pseudo1d_dim2_setup

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_values.json")))
DIRS = ["EveryDirection", "HorizontalDirection", "VerticalDirection"]


@pytest.mark.parametrize("flux_bc", [False, True])
@pytest.mark.parametrize("direction", [0, 1, 2])
@pytest.mark.parametrize("level", [1, 2])
def test_pseudo1d_dim2_l2_error(oracle, level, direction, flux_bc):
    Ne = 4 * 2 ** (level - 1)
    law, grid, dt, scale = pseudo1d_dim2_setup(Ne=Ne, direction=direction, flux_bc=flux_bc)
    dg = oracle.OracleDGModel(law, grid, nf_first=0, direction=direction)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    t, nsteps = oracle.solve(dg, Q, dt, 1.0)
    assert t == 1.0 and nsteps == 64 * Ne
    # the slice stays y-invariant
    q = Q[:grid.nreal, 0, :].reshape(grid.nreal, 5, 5, 5)
    assert np.max(np.abs(q - q[:, :, :1, :])) < 1e-11      # [e][k][j][i]: j runs across y
    Qe = law.init_state_prognostic(grid, dg.state_auxiliary, 1.0)
    err = np.sqrt(oracle.weighted_norm2_local(grid, Q, Qe)) / scale
    g = GOLD["pseudo1D_advection_diffusion"]
    exp = g["dim2"][DIRS[direction]][level - 1]
    assert abs(err - exp) <= g["rtol"] * exp, (err, exp)
    assert abs(err - exp) <= 1e-10 * exp, (err, exp)


@pytest.mark.parametrize("direction", [0, 1, 2])
def test_periodic_hyperdiffusion_dim2_level1(oracle, direction):
    law, grid, dt, scale = periodic_hyperdiffusion_dim2_setup(Ne=4, direction=direction)
    dg = oracle.OracleDGModel(law, grid, nf_first=1, direction=direction)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    t, _ = oracle.solve(dg, Q, dt, 1.0)
    assert t == 1.0
    Qe = law.init_state_prognostic(grid, dg.state_auxiliary, 1.0)
    err = np.sqrt(oracle.weighted_norm2_local(grid, Q, Qe)) / scale
    g = GOLD["periodic_3D_hyperdiffusion"]
    exp = g["dim2"][DIRS[direction]][0]
    assert abs(err - exp) <= g["rtol"] * exp, (err, exp)
