"""The reference's dim = 2 golden rows on the device, every refinement level
(pseudo1D_advection_diffusion.jl:242-253 levels 1-4, periodic_3D_hyperdiffusion.jl:231-239
levels 1-3): the 2-D problem is the y-invariant slice of a 3-D one (helpers.pseudo1d_dim2_setup,
DESIGN.md "dim = 2"); level 1 is also compared with the oracle state by state."""
import json
import os

import numpy as np
import pytest

from helpers import periodic_hyperdiffusion_dim2_setup, pseudo1d_dim2_setup, rel_linf

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_values.json")))
DIRS = ["EveryDirection", "HorizontalDirection", "VerticalDirection"]


@pytest.mark.parametrize("flux_bc", [False, True])
@pytest.mark.parametrize("direction", [0, 1, 2])
@pytest.mark.parametrize("level", [1, 2, 3, 4])
def test_pseudo1d_dim2_gpu(cm, oracle, torch, level, direction, flux_bc):
    Ne = 4 * 2 ** (level - 1)
    law, grid, dt, scale = pseudo1d_dim2_setup(Ne=Ne, direction=direction, flux_bc=flux_bc)
    dg = cm.dgmodel.DGModel(law, grid, direction=direction)
    Q = dg.init_ode_state(0.0)
    solver = cm.odesolvers.LSRK54CarpenterKennedy(dg, Q, dt=dt, t0=0.0)
    tend = cm.odesolvers.solve(Q, solver, timeend=1.0)
    assert tend == 1.0 and solver.steps == 64 * Ne
    err = dg.euclidean_distance(Q, dg.init_ode_state(1.0)) / scale
    g = GOLD["pseudo1D_advection_diffusion"]
    exp = g["dim2"][DIRS[direction]][level - 1]
    assert abs(err - exp) <= g["rtol"] * exp, (err, exp)
    assert abs(err - exp) <= 1e-9 * exp, (err, exp)
    if level == 1:
        odg = oracle.OracleDGModel(law, grid, nf_first=0, direction=direction)
        Qo = law.init_state_prognostic(grid, odg.state_auxiliary, 0.0)
        oracle.solve(odg, Qo, dt, 1.0)
        assert rel_linf(Q.cpu().numpy()[:grid.nreal], Qo[:grid.nreal]) < 1e-11
    dg.close()


@pytest.mark.parametrize("direction", [0, 1, 2])
@pytest.mark.parametrize("level", [1, 2, 3])
def test_periodic_hyperdiffusion_dim2_gpu(cm, oracle, torch, level, direction):
    Ne = 4 * 2 ** (level - 1)
    law, grid, dt, scale = periodic_hyperdiffusion_dim2_setup(Ne=Ne, direction=direction)
    dg = cm.dgmodel.DGModel(law, grid, numerical_flux_first_order=1, direction=direction)
    Q = dg.init_ode_state(0.0)
    solver = cm.odesolvers.LSRK54CarpenterKennedy(dg, Q, dt=dt)
    cm.odesolvers.solve(Q, solver, timeend=1.0)
    err = dg.euclidean_distance(Q, dg.init_ode_state(1.0)) / scale
    g = GOLD["periodic_3D_hyperdiffusion"]
    exp = g["dim2"][DIRS[direction]][level - 1]
    assert abs(err - exp) <= g["rtol"] * exp, (err, exp)
    if level == 1:
        odg = oracle.OracleDGModel(law, grid, nf_first=1, direction=direction)
        Qo = law.init_state_prognostic(grid, odg.state_auxiliary, 0.0)
        oracle.solve(odg, Qo, dt, 1.0)
        assert rel_linf(Q.cpu().numpy()[:grid.nreal], Qo[:grid.nreal]) < 1e-10
    dg.close()
