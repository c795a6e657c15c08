"""mms_bc_atmos.jl (dim = 3) on the GPU: right-hand side and LSRK steps against the oracle, and
the reference's expected errors at refinement levels 1-3 run end to end on the device.
``-m gpu``."""
import numpy as np
import pytest

from helpers import mms_atmos_setup, rel_linf
from test_mms_oracle import GOLD

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "no HIP device: the product path has no CPU fallback"
    return t


def _gpu(torch, a):
    x = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    torch.cuda.synchronize()
    return x


def test_mms_tendency_matches_oracle(cm, oracle, torch):
    law, grid, dt, _ = mms_atmos_setup(level=1)
    odg = oracle.OracleDGModel(law, grid)
    dg = cm.dgmodel.DGModel(law, grid)
    rng = np.random.default_rng(2)
    Q0 = law.init_state_prognostic(grid, odg.state_auxiliary, 0.2)
    Q0 *= 1 + 1e-2 * rng.standard_normal(Q0.shape)
    T0 = rng.standard_normal(Q0.shape)
    for alpha, beta, t in ((1.0, 0.0, 0.2), (0.5, 2.0, 0.7)):
        To = T0.copy()
        odg(To, Q0.copy(), t, alpha, beta)
        Tg = _gpu(torch, T0)
        dg(Tg, _gpu(torch, Q0), t, alpha, beta)
        Tn = Tg.cpu().numpy()
        for s in range(5):
            assert rel_linf(Tn[:, s], To[:, s]) < TOL, s
        assert rel_linf(dg.state_gradient_flux.cpu().numpy()[:, 3:], odg.state_gradient_flux[:, 3:]) < TOL
        # total_specific_enthalpy == 0: the enthalpy gradient is identically zero
        assert not dg.state_gradient_flux.cpu().numpy()[:, :3].any()
    Qo, dQo = Q0.copy(), np.zeros_like(Q0)
    for i in range(3):
        oracle.lsrk54_step(odg, Qo, dQo, i * dt, dt)
    Q = _gpu(torch, Q0)
    dQ = torch.zeros_like(Q)
    dg.lsrk_run(Q, dQ, 0.0, dt, 3, oracle.RKA, oracle.RKB, oracle.RKC)
    dg.synchronize()
    assert rel_linf(Q.cpu().numpy(), Qo) < TOL
    dg.close()


@pytest.mark.parametrize("level", [1, 2, 3])
def test_mms_reference_errors_on_the_device(cm, oracle, torch, level):
    law, grid, dt, nsteps = mms_atmos_setup(level=level)
    dg = cm.dgmodel.DGModel(law, grid)
    Q = dg.init_ode_state(0.0)
    dQ = torch.zeros_like(Q)
    dg.lsrk_run(Q, dQ, 0.0, dt, nsteps, oracle.RKA, oracle.RKB, oracle.RKC)
    dg.synchronize()
    Qe = law.init_state_prognostic(grid, dg.state_auxiliary.cpu().numpy(), 1.0)
    err = np.sqrt(oracle.weighted_norm2_local(grid, Q.cpu().numpy(), Qe))
    ref = GOLD["dim3"][level - 1]
    assert abs(err - ref) <= GOLD["rtol"] * ref, (err, ref)
    dg.close()


def test_driver_mms3_value_on_the_device(cm, oracle, torch):
    """test/Driver/mms3.jl through solve(..., adjustfinalstep=False): 801 steps (see
    tests/test_mms_oracle.py)."""
    law, grid, dt, _ = mms_atmos_setup(level=1)
    dg = cm.dgmodel.DGModel(law, grid)
    Q = dg.init_ode_state(0.0)
    solver = cm.odesolvers.LSRK54CarpenterKennedy(dg, Q, dt=dt)
    cm.odesolvers.solve(Q, solver, timeend=1.0, adjustfinalstep=False)
    assert solver.steps == 801
    Qe = law.init_state_prognostic(grid, dg.state_auxiliary.cpu().numpy(), 1.0)
    err = np.sqrt(oracle.weighted_norm2_local(grid, Q.cpu().numpy(), Qe))
    ref = GOLD["driver_mms3"]
    assert abs(err - ref) <= GOLD["rtol"] * ref, (err, ref)
    dg.close()
