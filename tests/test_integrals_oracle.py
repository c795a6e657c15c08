"""Column (stack) integrals: the oracle's restatement of kernel_indefinite_stack_integral! /
kernel_reverse_indefinite_stack_integral! against the reference's analytic test
(test/Numerics/DGMethods/integral_test.jl, dim = 3, polynomial orders (4, 4) and (4, 3)).
CPU only."""
import numpy as np
import pytest

from cmdg_loader import cm

M = cm.mesh


def integral_test_aux(grid):
    """nodal_init_state_auxiliary! of IntegralTestModel{3} (integral_test.jl:62-87): columns
    int.a int.b rev_int.a rev_int.b coord[3] a b rev_a rev_b."""
    aux = np.zeros((grid.nelem, 11, grid.Np))
    x, y, z = (grid.vgeo[:, 12 + d, :] for d in range(3))
    aux[:, 4], aux[:, 5], aux[:, 6] = x, y, z
    aux[:, 7] = x * z + y * z
    aux[:, 8] = 2 * x * z + np.sin(x) * y * z - (1 + (z - 1) ** 3) * y ** 2 / 3
    zt = 3.0
    aux[:, 9] = (x * zt + y * zt) - aux[:, 7]
    aux[:, 10] = (2 * x * zt + np.sin(x) * y * zt - (1 + (zt - 1) ** 3) * y ** 2 / 3) - aux[:, 8]
    return aux


def integral_test_grid(N, Ne=(5, 6, 7), rank=0, size=1):
    rng = [np.linspace(0.0, 3.0, n + 1) for n in Ne]
    topl = M.StackedBrickTopology(rng, periodicity=(True,) * 3, connectivity="full",
                                  rank=rank, size=size)
    return M.DiscontinuousSpectralElementGrid(topl, N)


def _approx(a, b):
    """Julia's ``isapprox`` on arrays: norm(a - b) <= sqrt(eps) * max(norm(a), norm(b))."""
    return np.linalg.norm(a - b) <= np.sqrt(np.finfo(float).eps) * max(np.linalg.norm(a),
                                                                        np.linalg.norm(b))


def test_indefinite_integral_interpolation_matrix():
    x, w = M.elements.lglpoints(4)
    I = M.grids.indefinite_integral_interpolation_matrix(x, w)
    assert np.all(I[0] == 0)
    assert np.allclose(I[-1], w, atol=1e-15)                 # full integral = quadrature
    for p in range(5):                                       # exact for the interpolant
        assert np.allclose(I @ x ** p, (x ** (p + 1) - (-1.0) ** (p + 1)) / (p + 1), atol=1e-14)


@pytest.mark.parametrize("N", [(4, 4), (4, 3)])
def test_stack_integrals_match_reference_test(oracle, N):
    grid = integral_test_grid(N)
    nr = grid.nreal
    og = oracle.OracleGrid(grid)
    aux = integral_test_aux(grid)
    law = oracle.integral_test_law()
    oracle.indefinite_stack_integral(law, og, None, aux)
    oracle.reverse_indefinite_stack_integral(law, og, None, aux)
    assert _approx(aux[:nr, 0], aux[:nr, 7])        # forward integral a  (integral_test.jl:169)
    assert _approx(aux[:nr, 1], aux[:nr, 8])        # forward integral b
    assert _approx(aux[:nr, 2], aux[:nr, 9])        # reverse integral a
    assert _approx(aux[:nr, 3], aux[:nr, 10])       # reverse integral b
    # a is linear in z: integrated exactly
    assert np.abs(aux[:nr, 0] - aux[:nr, 7]).max() < 1e-12


def test_field_law_equals_test_law(oracle):
    """the field-combination law (what the C ABI carries) with the integrand precomputed into
    auxiliary columns reproduces the test law."""
    grid = integral_test_grid((4, 4), Ne=(2, 3, 4))
    og = oracle.OracleGrid(grid)
    aux = integral_test_aux(grid)
    ref = aux.copy()
    law = oracle.integral_test_law()
    oracle.indefinite_stack_integral(law, og, None, ref)
    oracle.reverse_indefinite_stack_integral(law, og, None, ref)
    x, y, z = aux[:, 4], aux[:, 5], aux[:, 6]
    aux2 = np.concatenate([aux, np.zeros((grid.nelem, 2, grid.Np))], axis=1)
    aux2[:, 11] = x + y
    aux2[:, 12] = (2 * x + np.sin(x) * y - (z - 1) ** 2 * y ** 2) / 0.5
    Q = np.zeros((grid.nelem, 1, grid.Np))
    fl = oracle.integral_fields_law([(0, 11), (0, 12)], [1.0, 0.5], [0, 1], [0, 1], [2, 3], 1, 13)
    oracle.indefinite_stack_integral(fl, og, Q, aux2)
    oracle.reverse_indefinite_stack_integral(fl, og, Q, aux2)
    assert np.abs(aux2[:, :4] - ref[:, :4]).max() < 1e-13
