"""Column (stack) integrals on the GPU (``cmdg_indefinite_stack_integral`` /
``cmdg_reverse_indefinite_stack_integral``) against the oracle and the analytic integrals of
the reference's test/Numerics/DGMethods/integral_test.jl.  ``-m gpu``."""
import numpy as np
import pytest

from cmdg_loader import cm
from test_integrals_oracle import _approx, integral_test_aux, integral_test_grid

pytestmark = pytest.mark.gpu
BL = cm.balancelaws


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "no HIP device: the product path has no CPU fallback"
    return t


def _gpu(torch, a):
    x = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    torch.cuda.synchronize()
    return x


def _law():
    n = np.ones(3) / np.sqrt(3)
    return BL.AdvectionDiffusion(3, BL.Pseudo1D(n, 1.0, 1 / 100, -1 / 2, 1 / 10), ())


def _with_integrands(aux):
    x, y, z = aux[:, 4], aux[:, 5], aux[:, 6]
    ext = np.zeros((aux.shape[0], 2, aux.shape[2]))
    ext[:, 0] = x + y
    ext[:, 1] = (2 * x + np.sin(x) * y - (z - 1) ** 2 * y ** 2) / 0.5
    return np.ascontiguousarray(np.concatenate([aux, ext], axis=1))


@pytest.mark.parametrize("Ne", [(5, 6, 7), (3, 2, 13)])
def test_stack_integrals_match_oracle_and_analytic(cm, oracle, torch, Ne):
    grid = integral_test_grid((4, 4), Ne=Ne)
    nr = grid.nreal
    dg = cm.dgmodel.DGModel(_law(), grid)
    aux0 = _with_integrands(integral_test_aux(grid))
    Q0 = np.random.default_rng(0).standard_normal((grid.nelem, 2, grid.Np))
    aux0[:, 12] = (aux0[:, 12] * 0.5) / 0.25 - 0 * Q0[:, 1]      # rescaled: scale 0.25 below
    # oracle: field law, integrand 1 from aux, integrand 2 = 0.25 * aux column 12
    og = oracle.OracleGrid(grid)
    auxo = aux0.copy()
    fl = oracle.integral_fields_law([(0, 11), (0, 12)], [1.0, 0.25], [0, 1], [0, 1], [2, 3], 2, 13)
    oracle.indefinite_stack_integral(fl, og, Q0, auxo)
    oracle.reverse_indefinite_stack_integral(fl, og, Q0, auxo)
    aux = _gpu(torch, aux0)
    Q = _gpu(torch, Q0)
    dg.indefinite_stack_integral(Q, aux, [(0, 11), (0, 12)], [0, 1], scale=[1.0, 0.25])
    dg.reverse_indefinite_stack_integral(aux, [0, 1], [2, 3])
    dg.synchronize()
    auxg = aux.cpu().numpy()
    assert np.array_equal(auxg, auxo), np.abs(auxg - auxo).max()     # same order of operations
    # the reference's assertions (integral_test.jl:166-178)
    for c, ex in ((0, 7), (1, 8), (2, 9), (3, 10)):
        assert _approx(auxg[:nr, c], auxg[:nr, ex])
    dg.close()


def test_state_integrands_and_many_outputs(cm, oracle, torch):
    """integrands read from the prognostic state; 6 outputs = two launches of <= 4."""
    grid = integral_test_grid((4, 4), Ne=(2, 3, 4))
    dg = cm.dgmodel.DGModel(_law(), grid)
    rng = np.random.default_rng(1)
    Q0 = rng.standard_normal((grid.nelem, 3, grid.Np))
    aux0 = rng.standard_normal((grid.nelem, 14, grid.Np))
    src = [(1, 0), (1, 2), (0, 13), (1, 1), (0, 12), (1, 0)]
    scale = [1.0, -2.0, 0.5, 3.0, 1.0, 1e-3]
    dst = [0, 1, 2, 3, 4, 5]
    rdst = [6, 7, 8, 9, 10, 11]
    og = oracle.OracleGrid(grid)
    auxo = aux0.copy()
    fl = oracle.integral_fields_law(src, scale, dst, dst, rdst, 3, 14)
    oracle.indefinite_stack_integral(fl, og, Q0, auxo)
    oracle.reverse_indefinite_stack_integral(fl, og, Q0, auxo)
    aux = _gpu(torch, aux0)
    dg.indefinite_stack_integral(_gpu(torch, Q0), aux, src, dst, scale=scale)
    dg.reverse_indefinite_stack_integral(aux, dst, rdst)
    dg.synchronize()
    assert np.array_equal(aux.cpu().numpy(), auxo)
    with pytest.raises(cm._lib.CmdgError):
        dg.indefinite_stack_integral(None, aux, [(1, 0)], [0])       # state integrand without Q
    with pytest.raises(cm._lib.CmdgError):
        dg.reverse_indefinite_stack_integral(aux, [0], [14])         # column out of range
    dg.close()
