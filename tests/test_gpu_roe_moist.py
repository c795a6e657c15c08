"""RoeNumericalFluxMoist on the device: the reference's error table of the moist isentropic vortex
(test/Numerics/DGMethods/Euler/isentropicvortex.jl:120-142), five variants x four refinement
levels, and the device tendency against the oracle's."""
import numpy as np
import pytest

from helpers import isentropic_vortex_moist_setup, rel_linf
from test_roe_moist_oracle import FLUXES, GOLD

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nf,name", FLUXES)
def test_tendency_matches_oracle(cm, oracle, torch, nf, name):
    law, grid, dt, _, _ = isentropic_vortex_moist_setup(1)
    odg = oracle.OracleDGModel(law, grid, nf_first=nf, direction=0)
    dg = cm.dgmodel.DGModel(law, grid, numerical_flux_first_order=nf, direction=0)
    Q0 = law.init_state_prognostic(grid, odg.state_auxiliary, 0.0)
    rng = np.random.default_rng(nf)
    Q0[:, :5] *= 1 + 1e-3 * rng.standard_normal(Q0[:, :5].shape)
    To = np.zeros_like(Q0)
    odg(To, Q0.copy(), 0.0, 1.0, 0.0)
    Q = torch.from_numpy(Q0.copy()).cuda()
    T = dg.create_state()
    torch.cuda.synchronize()
    dg(T, Q, 0.0, 1.0, 0.0)
    Tg = T.cpu().numpy()
    for s in range(5):
        assert rel_linf(Tg[:, s], To[:, s]) < 1e-11, (name, s)
    assert np.abs(Tg[:, 5]).max() <= 1e-12 * np.abs(Tg[:, 0]).max()
    dg.close()


@pytest.mark.parametrize("level", [1, 2, 3, 4])
@pytest.mark.parametrize("nf,name", FLUXES)
def test_moist_vortex_error_table(cm, torch, nf, name, level):
    law, grid, dt, timeend, nsteps = isentropic_vortex_moist_setup(level)
    dg = cm.dgmodel.DGModel(law, grid, numerical_flux_first_order=nf, direction=0)
    Q = dg.init_ode_state(0.0)
    solver = cm.odesolvers.LSRK54CarpenterKennedy(dg, Q, dt=dt)
    cm.odesolvers.solve(Q, solver, timeend=timeend)
    assert solver.steps in (nsteps, nsteps + 1)
    err = dg.euclidean_distance(Q, dg.init_ode_state(timeend))
    exp = GOLD["dim3"][name][level - 1]
    assert abs(err - exp) <= GOLD["rtol"] * exp, (name, level, err, exp)
    dg.close()
