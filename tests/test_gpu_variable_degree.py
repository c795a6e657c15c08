"""polynomialorder = (N_h, N_v) on the GPU: the kernels templated on separate horizontal and
vertical point counts against the oracle, the reference's stored errors of
variable_degree_advection_diffusion.jl (levels 1-3), and the split-explicit ocean with its
barotropic model on a two-node extrusion.  ``-m gpu``."""
import numpy as np
import pytest

from helpers import (check_split_explicit_table, rel_linf, split_explicit_fields,
                     split_explicit_schedule, split_explicit_setup, variable_degree_setup)
from test_split_explicit_oracle import GOLD as SE_GOLD, relaxed
from test_variable_degree import GOLD, expected

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


@pytest.mark.parametrize("orders", [(4, 2), (2, 4)])
@pytest.mark.parametrize("direction", [0, 1, 2])
def test_mixed_order_tendency_matches_oracle(cm, oracle, torch, orders, direction):
    law, grid, dt = variable_degree_setup(1, orders, "horizontal")
    law.problem.n = np.ones(3) / np.sqrt(3)          # flow across every face
    odg = oracle.OracleDGModel(law, grid, direction=direction)
    dg = cm.dgmodel.DGModel(law, grid, direction=direction)
    rng = np.random.default_rng(orders[0])
    Q0 = law.init_state_prognostic(grid, odg.state_auxiliary, 0.1)
    Q0 = Q0 + 1e-2 * rng.standard_normal(Q0.shape)
    T0 = rng.standard_normal(Q0.shape)
    To = T0.copy()
    odg(To, Q0.copy(), 0.2, 0.5, 2.0)
    Tg = _gpu(torch, T0)
    dg(Tg, _gpu(torch, Q0), 0.2, 0.5, 2.0)
    assert rel_linf(Tg.cpu().numpy(), To) < TOL
    assert rel_linf(dg.state_gradient_flux.cpu().numpy(), odg.state_gradient_flux) < TOL
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
@pytest.mark.parametrize("orders", [(4, 2), (2, 4)])
@pytest.mark.parametrize("field", ["horizontal", "vertical"])
def test_variable_degree_reference_errors_on_the_device(cm, torch, orders, field, level):
    law, grid, dt = variable_degree_setup(level, orders, field)
    dg = cm.dgmodel.DGModel(law, grid)
    Q = dg.init_ode_state(0.0)
    solver = cm.odesolvers.LSRK54CarpenterKennedy(dg, Q, dt=dt)
    cm.odesolvers.solve(Q, solver, timeend=1.0)
    Qe = _gpu(torch, law.init_state_prognostic(grid, dg.state_auxiliary.cpu().numpy(), 1.0))
    err = dg.euclidean_distance(Q, Qe)
    exp = expected(orders, field, level)
    assert abs(err - exp) <= GOLD["rtol"] * exp, (err, exp)
    dg.close()


def test_unsupported_mixed_order_fails_loudly(cm):
    law, grid, _ = variable_degree_setup(1, (3, 2), "horizontal")
    with pytest.raises(cm._lib.CmdgError):
        cm.dgmodel.DGModel(law, grid)


@pytest.mark.parametrize("name,dt_slow", [("coupled", 300.0), ("ninety_minutes", 5400.0)])
def test_split_explicit_with_two_node_extrusion(cm, oracle, torch, name, dt_slow):
    """The barotropic model on (N, N, 1): same tables as with the five-node extrusion."""
    O = cm.ocean
    law3, g3, law2, g2 = split_explicit_setup(True, N_extrusion=1)
    assert g2.Nq[2] == 2
    dg3 = cm.dgmodel.DGModel(law3, g3)
    keep = O.install_hydrostatic_boussinesq_hooks(dg3)
    dg2 = cm.dgmodel.DGModel(law2, g2,
                             numerical_flux_first_order=cm.balancelaws.CentralNumericalFluxFirstOrder)
    Q3g, Q2g = dg3.init_ode_state(0.0), dg2.init_ode_state(0.0)
    dt, nsteps = split_explicit_schedule(dt_slow)
    se = O.SplitExplicitSolver(dg3, dg2, Q3g, Q2g, dt, 300.0)
    se.dostep(Q3g, Q2g, nsteps)
    fields = split_explicit_fields(Q3g.cpu().numpy(), dg3.state_auxiliary.cpu().numpy(),
                                   Q2g.cpu().numpy(), dg2.state_auxiliary.cpu().numpy(), g2)
    check_split_explicit_table(SE_GOLD[name], relaxed(SE_GOLD["parr"]), fields, slack=3.0)
    dg3.set_rhs_hooks()
    for f in keep:
        f.close()
    dg3.close()
    dg2.close()
