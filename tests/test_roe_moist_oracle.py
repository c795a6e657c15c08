"""RoeNumericalFluxMoist (src/Atmos/Model/AtmosModel.jl:1276-1513) with its four optional fixes on
the moist isentropic vortex: the only numbers the reference stores that run through the moist law
(EquilMoist at q_tot = 0: PhaseEquil construction, moist gas constants, the six-variable Roe
dissipation).  test/Numerics/DGMethods/Euler/isentropicvortex.jl:120-142, level 1 on the oracle
(levels 1 - 4 on the device: tests/test_gpu_roe_moist.py)."""
import json
import os

import numpy as np
import pytest

from helpers import cm, isentropic_vortex_moist_setup

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_values.json")))["isentropicvortex"]
FLUXES = [(5, "RoeMoist"), (6, "RoeMoistLM"), (7, "RoeMoistHH"), (8, "RoeMoistLV"), (9, "RoeMoistLVPP")]


@pytest.mark.parametrize("nf,name", FLUXES)
def test_moist_vortex_level_one(oracle, nf, name):
    law, grid, dt, timeend, nsteps = isentropic_vortex_moist_setup(1)
    dg = oracle.OracleDGModel(law, grid, nf_first=nf, direction=0)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    assert not Q[:, 5].any()
    dQ = np.zeros_like(Q)
    for s in range(nsteps):
        oracle.lsrk54_step(dg, Q, dQ, s * dt, dt)
    Qe = law.init_state_prognostic(grid, dg.state_auxiliary, timeend)
    err = np.sqrt(oracle.weighted_norm2_local(grid, Q, Qe))
    exp = GOLD["dim3"][name][0]
    assert abs(err - exp) <= GOLD["rtol"] * exp, (err, exp)
    assert not Q[:, 5].any()          # no moisture appears
