"""Polynomial orders other than N = 4 (every kernel is templated on Nq = N + 1): the HIP path
against the oracle for N = 1..7 (advection-diffusion), N = 2, 3, 5, 6 (dry atmosphere, with
hyperdiffusion and with SmagorinskyLilly -- N = 6 is the order of BASELINE configs[3]), and the
reference's filter / integral tests at their own orders (N = 3).  ``-m gpu``."""
import numpy as np
import pytest

from helpers import held_suarez_setup, pseudo1d_setup, rel_linf, rising_bubble_setup
from test_filters_oracle import _filter_test_state
from test_integrals_oracle import _approx, integral_test_aux, integral_test_grid

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


@pytest.mark.parametrize("N", [1, 2, 3, 5, 6, 7])
def test_advdiff_orders_match_oracle(cm, oracle, torch, N):
    law, grid, dt = pseudo1d_setup(Ne=3, N=N)
    odg = oracle.OracleDGModel(law, grid)
    dg = cm.dgmodel.DGModel(law, grid)
    Q0 = law.init_state_prognostic(grid, odg.state_auxiliary, 0.0)
    Q0 = Q0 + 1e-3 * np.random.default_rng(N).standard_normal(Q0.shape)
    T0 = np.random.default_rng(9).standard_normal(Q0.shape)
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
    assert dg.min_node_distance() == cm.mesh.grids.min_node_distance(grid)
    dg.close()


@pytest.mark.parametrize("N", [2, 3, 5, 6])
def test_held_suarez_orders_match_oracle(cm, oracle, torch, N):
    law, grid, d, dd = held_suarez_setup(n_horz=2, n_vert=2, N=N)
    odg = oracle.OracleDGModel(law, grid, direction=d, diffusion_direction=dd)
    dg = cm.dgmodel.DGModel(law, grid, direction=d, diffusion_direction=dd)
    Q0 = law.init_state_prognostic(grid, odg.state_auxiliary, 0.0)
    rng = np.random.default_rng(N)
    Q0[:, 1:4] += Q0[:, 0:1] * 2.0 * rng.standard_normal(Q0[:, 1:4].shape)
    To = np.zeros_like(Q0)
    odg(To, Q0.copy(), 0.0, 1.0, 0.0)
    Tg = _gpu(torch, np.zeros_like(Q0))
    Qg = _gpu(torch, Q0)
    dg(Tg, Qg, 0.0, 1.0, 0.0)
    Tn = Tg.cpu().numpy()
    nr = grid.nreal
    for s in range(5):
        assert rel_linf(Tn[:nr, s], To[:nr, s]) < TOL, s
    for kind in (0, 1):
        o = oracle.courant(kind, odg, Q0, 1.0, 0.0, 0)
        assert abs(dg.courant(kind, Qg, 1.0, 0.0, 0) - o) <= 1e-12 * abs(o)
    dg.close()


def test_smagorinsky_order_six_matches_oracle(cm, oracle, torch):
    """N = 6 with the SmagorinskyLilly closure: one element's working set is 78 KB of LDS in
    k_tendency (more than the 64 KB a work-group gets on older parts)."""
    law, grid = rising_bubble_setup(nx=2, ny=2, nz=3, N=6)
    odg = oracle.OracleDGModel(law, grid)
    dg = cm.dgmodel.DGModel(law, grid)
    Q0 = law.init_state_prognostic(grid, odg.state_auxiliary, 0.0)
    rng = np.random.default_rng(6)
    Q0[:, 1:4] += Q0[:, 0:1] * 3.0 * rng.standard_normal(Q0[:, 1:4].shape)
    To = np.zeros_like(Q0)
    odg(To, Q0.copy(), 0.0, 1.0, 0.0)
    Tg = _gpu(torch, np.zeros_like(Q0))
    dg(Tg, _gpu(torch, Q0), 0.0, 1.0, 0.0)
    Tn = Tg.cpu().numpy()
    for s in range(5):
        assert rel_linf(Tn[:, s], To[:, s]) < TOL, s
    gfg = dg.state_gradient_flux.cpu().numpy()
    for s in range(law.ngradflux):
        sc = max(np.abs(odg.state_gradient_flux[:, s]).max(), 1e-300)
        assert np.abs(gfg[:, s] - odg.state_gradient_flux[:, s]).max() / sc < TOL, s
    Qo, dQo = Q0.copy(), np.zeros_like(Q0)
    for i in range(2):
        oracle.lsrk54_step(odg, Qo, dQo, i * 0.02, 0.02)
    Q = _gpu(torch, Q0)
    dQ = torch.zeros_like(Q)
    dg.lsrk_run(Q, dQ, 0.0, 0.02, 2, oracle.RKA, oracle.RKB, oracle.RKC)
    dg.synchronize()
    assert rel_linf(Q.cpu().numpy(), Qo) < 1e-11
    dg.close()


@pytest.mark.parametrize("kind", ["CutoffFilter", "MassPreservingCutoffFilter"])
@pytest.mark.parametrize("direction", [0, 1, 2])
def test_reference_filter_test_at_its_own_order(cm, oracle, torch, direction, kind):
    """test/Numerics/Mesh/filter.jl:199-330, dim = 3: N = 3, one element, CutoffFilter(grid, 2)."""
    F = cm.mesh.filters
    law, grid, _ = pseudo1d_setup(Ne=1, N=3)
    dg = cm.dgmodel.DGModel(law, grid)
    filt = getattr(F, kind)(grid, 2)
    Q0 = _filter_test_state(grid, None)
    Q = _gpu(torch, Q0)
    F.apply(Q, (1, 3), dg, filt, direction=direction)
    P = _filter_test_state(grid, direction)
    Qg = Q.cpu().numpy()
    assert np.linalg.norm(Qg - P) <= np.sqrt(np.finfo(float).eps) * np.linalg.norm(P)
    assert np.abs(Qg - P).max() < 5e-13
    Qo = Q0.copy()
    oracle.apply_filter(Qo, F.FilterIndices(1, 3), grid, filt, direction=direction)
    assert np.array_equal(Qg, Qo)
    dg.close()


@pytest.mark.parametrize("N", [3, 6])
def test_tmar_and_exponential_filters_other_orders(cm, oracle, torch, N):
    F = cm.mesh.filters
    law, grid, _ = pseudo1d_setup(Ne=2, N=N)
    dg = cm.dgmodel.DGModel(law, grid)
    Q0 = np.random.default_rng(N).standard_normal((grid.nelem, 3, grid.Np))
    for filt, tg in ((F.ExponentialFilter(grid, 1, 8), F.FilterIndices(1, 3)),
                     (F.TMARFilter(), F.FilterIndices(2))):
        Qo = Q0.copy()
        oracle.apply_filter(Qo, tg, grid, filt)
        Q = _gpu(torch, Q0)
        F.apply(Q, tg, dg, filt)
        assert np.array_equal(Q.cpu().numpy(), Qo)
    dg.close()


def test_stack_integrals_at_order_3(cm, oracle, torch):
    BL = cm.balancelaws
    grid = integral_test_grid((3, 3), Ne=(3, 2, 5))
    law = BL.AdvectionDiffusion(3, BL.Pseudo1D(np.ones(3) / np.sqrt(3), 1.0, 1 / 100, -1 / 2, 1 / 10), ())
    dg = cm.dgmodel.DGModel(law, grid)
    aux0 = integral_test_aux(grid)
    x, y, z = aux0[:, 4], aux0[:, 5], aux0[:, 6]
    ext = np.stack([x + y, 2 * x + np.sin(x) * y - (z - 1) ** 2 * y ** 2], axis=1)
    aux0 = np.ascontiguousarray(np.concatenate([aux0, ext], axis=1))
    aux = _gpu(torch, aux0)
    dg.indefinite_stack_integral(None, aux, [(0, 11), (0, 12)], [0, 1])
    dg.reverse_indefinite_stack_integral(aux, [0, 1], [2, 3])
    dg.synchronize()
    a = aux.cpu().numpy()
    nr = grid.nreal
    for c, ex in ((0, 7), (1, 8), (2, 9), (3, 10)):
        assert _approx(a[:nr, c], a[:nr, ex])
    dg.close()
