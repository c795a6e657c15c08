"""The ShallowWaterModel on its own (test/Ocean/ShallowWater/test_2D_spindown.jl: 5 x 5 elements,
N = 4, ConstantViscosity(5e3), no advection, c = 1, central fluxes, 288 LSRK54 steps of 300 s)
against the reference's StateCheck rows and the analytic error it prints.  The 2-D law runs on
the one-layer extrusion of the 2-D grid (five nodes, or two with N_extrusion = 1); statistics are
taken on the k = 0 plane.  CPU only (~5 s)."""
import json
import os

import numpy as np
import pytest

from cmdg_loader import cm
from helpers import check_split_explicit_table

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ocean_2d_spindown_refvals.json")))


def shallow_spindown_setup(N_extrusion=None):
    O = cm.ocean
    problem = O.SimpleBox(1e6, 1e6, 400.0)
    law = O.ShallowWaterModel(problem, 5e3, advection=False, coupled=False, c=1.0, f_o=0.0, beta=0.0)
    x = np.linspace(0.0, 1e6, 6)
    grid = O.extruded_barotropic_grid(x, x, 4, N_extrusion=N_extrusion)
    return law, grid, 300.0, 288


def plane_fields(Q, grid):
    Nqh = grid.Nq[0] * grid.Nq[1]
    q0 = np.asarray(Q).reshape(grid.nelem, 3, grid.Nq[2], Nqh)[:grid.nreal, :, 0, :]
    return {("2D state", "η"): q0[:, 0], ("2D state", "U[1]"): q0[:, 1], ("2D state", "U[2]"): q0[:, 2]}


@pytest.mark.parametrize("N_extrusion", [None, 1])
def test_shallow_water_spindown_matches_reference(oracle, N_extrusion):
    law, grid, dt, nsteps = shallow_spindown_setup(N_extrusion)
    dg = oracle.OracleDGModel(law, grid, nf_first=1)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    dQ = np.zeros_like(Q)
    for i in range(nsteps):
        oracle.lsrk54_step(dg, Q, dQ, i * dt, dt)
    Qe = law.init_state_prognostic(grid, dg.state_auxiliary, 86400.0)
    err = np.sqrt(oracle.weighted_norm2_local(grid, Q, Qe) / oracle.weighted_norm2_local(grid, Qe))
    assert err < 0.005                                             # test_2D_spindown.jl:103
    # the value quoted in the comment at :219 (1.1328e-4) is not reproduced (1.0329e-4 here) although
    # every checked statistic of the table is, to 1e-13: the comment is taken to be stale
    assert abs(err - GOLD["error_printed_by_reference"]) < 2e-5
    check_split_explicit_table(GOLD["explicit"], GOLD["parr"], plane_fields(Q, grid), slack=2.0)
