"""Column integrals on the cubed sphere against the reference's stored errors
(test/Numerics/DGMethods/integral_test_sphere.jl:228-272: euclidean distance between the exact
and the computed auxiliary state, N = 4 and N = 1).  The integrands of the test law are
tabulated into auxiliary columns and integrated through the field-combination law.  CPU tests
use the oracle; the ``gpu`` ones libcmdg."""
import numpy as np
import pytest

from cmdg_loader import cm

M = cm.mesh
EXPECTED = {4: [4.662884229467401e-7, 7.218989778540723e-9, 1.1258613174916711e-10],
            1: [1.5934735012225074e-02, 4.0030667455285352e-03, 1.0020652111566574e-03]}
RIN, ROUT = 0.5, 1.0


def sphere_integral_setup(N, level):
    nh, nv = 2 ** (level - 1) * 4, 2 ** (level - 1) * 2
    topl = M.StackedCubedSphereTopology(nh, np.linspace(RIN, ROUT, nv + 1))
    grid = M.DiscontinuousSpectralElementGrid(topl, N, meshwarp=M.equiangular_cubed_sphere_warp)
    x, y, z = (grid.vgeo[:, 12 + d, :] for d in range(3))
    r = np.sqrt(x * x + y * y + z * z)            # hypot(x, y, z)
    th, ph = np.arctan2(y, x), np.arcsin(z / r)
    v = 1 + np.cos(ph) ** 2 * np.sin(th) ** 2 + np.sin(ph) ** 2
    # columns: int.v int.r rev_int.v rev_int.r r v | integrand.v integrand.r
    exact = np.stack([np.exp(-v * r ** 2) - np.exp(-v * RIN ** 2), r - RIN,
                      np.exp(-v * ROUT ** 2) - np.exp(-v * r ** 2), ROUT - r, r, v,
                      -2 * r * v * np.exp(-v * r ** 2), np.ones_like(r)], axis=1)
    return grid, np.ascontiguousarray(exact)


def distance(grid, a, b):
    nr = grid.nreal
    Mw = grid.vgeo[:nr, 9, :][:, None, :]
    return float(np.sqrt(np.sum(Mw * (a[:nr, :6] - b[:nr, :6]) ** 2)))


def _isapprox(a, b):
    return np.linalg.norm(a - b) <= np.sqrt(np.finfo(float).eps) * max(np.linalg.norm(a), np.linalg.norm(b))


@pytest.mark.parametrize("N,level", [(4, 1), (4, 2), (1, 1), (1, 2)])
def test_sphere_integrals_oracle(oracle, N, level):
    grid, exact = sphere_integral_setup(N, level)
    og = oracle.OracleGrid(grid)
    aux = exact.copy()
    law = oracle.integral_fields_law([(0, 6), (0, 7)], [1.0, 1.0], [0, 1], [0, 1], [2, 3], 1, 8)
    Q = np.zeros((grid.nelem, 1, grid.Np))
    oracle.indefinite_stack_integral(law, og, Q, aux)
    oracle.reverse_indefinite_stack_integral(law, og, Q, aux)
    nr = grid.nreal
    assert _isapprox(aux[:nr, 1], exact[:nr, 1]) and _isapprox(aux[:nr, 3], exact[:nr, 3])
    err = distance(grid, exact, aux)
    exp = EXPECTED[N][level - 1]
    assert abs(err - exp) <= 1e-3 * exp + np.finfo(float).eps      # the reference's tolerance
    assert abs(err - exp) <= 2e-6 * exp                            # observed ~1e-7 or better


@pytest.mark.gpu
@pytest.mark.parametrize("N,level", [(4, 1), (4, 2), (4, 3), (1, 1), (1, 3)])
def test_sphere_integrals_gpu(N, level):
    import torch
    BL = cm.balancelaws
    grid, exact = sphere_integral_setup(N, level)
    law = BL.AdvectionDiffusion(3, BL.SolidBodyRotation(), (BL.NoFlowBC(),), diffusion=False)
    dg = cm.dgmodel.DGModel(law, grid)
    aux = torch.from_numpy(exact).cuda()
    torch.cuda.synchronize()
    dg.indefinite_stack_integral(None, aux, [(0, 6), (0, 7)], [0, 1])
    dg.reverse_indefinite_stack_integral(aux, [0, 1], [2, 3])
    dg.synchronize()
    err = distance(grid, exact, aux.cpu().numpy())
    exp = EXPECTED[N][level - 1]
    assert abs(err - exp) <= 1e-3 * exp + np.finfo(float).eps
    dg.close()
