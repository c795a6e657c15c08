"""``auxiliary_field_gradient!`` on the stacked cubed sphere against the reference's stored
errors (test/Numerics/DGMethods/grad_test_sphere.jl, polynomial order (4,4)): a = r^3 on the
shell [1/2, 1]; the element-local gradient in every / vertical direction differs from
3 r^2 r_hat by the tabulated mass-weighted distance, the horizontal one vanishes to 1.3e-13.
Host-side one-time operator (it produces grad Phi of the orientation).  CPU only."""
import json
import os

import numpy as np
import pytest

from cmdg_loader import cm

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_values.json")))["grad_test_sphere"]
M = cm.mesh


@pytest.mark.parametrize("level", [1, 2])
def test_grad_sphere_matches_reference(level):
    nh, nv = 4 * 2 ** (level - 1), 2 * 2 ** (level - 1)
    topl = M.StackedCubedSphereTopology(nh, np.linspace(0.5, 1.0, nv + 1))
    grid = M.DiscontinuousSpectralElementGrid(topl, 4, meshwarp=M.equiangular_cubed_sphere_warp)
    G = M.grids
    x = [grid.vgeo[:, c, :] for c in (G._x1, G._x2, G._x3)]
    r = np.sqrt(x[0] ** 2 + x[1] ** 2 + x[2] ** 2)     # hypot(x, y, z)
    a = r ** 3
    exact = np.stack([3 * r ** 2 * xd / r for xd in x], axis=1)
    Mw = grid.vgeo[:, G._M, :]
    for direction in (0, 2):
        g = G.auxiliary_field_gradient(grid, a, direction)
        err = np.sqrt((Mw[:, None, :] * (g - exact) ** 2)[:grid.nreal].sum())
        ref = GOLD["N44"][level - 1]
        assert abs(err - ref) <= GOLD["rtol"] * ref, (direction, err, ref)
    g = G.auxiliary_field_gradient(grid, a, 1)
    err = np.sqrt((Mw[:, None, :] * g ** 2)[:grid.nreal].sum())
    assert err < GOLD["horizontal_atol"]


@pytest.mark.parametrize("direction", [0, 1, 2])
def test_grad_brick_is_exact_for_the_test_polynomial(direction):
    """test/Numerics/DGMethods/grad_test.jl:33-47, 121-150 (dim = 3, polynomial order (4,4),
    5^3 elements on [0,3]^3): a = x^2 + y^3 + z^2 y^2 - x y z is in the N = 4 space, so the
    element-local gradient equals the analytic one in the test's `isapprox` sense
    (norm(g - exact) <= sqrt(eps) * norm(exact)); the components a direction leaves out are zero."""
    rng = [np.linspace(0.0, 3.0, 6)] * 3
    topl = M.StackedBrickTopology(rng, periodicity=(False,) * 3, connectivity="full")
    grid = M.DiscontinuousSpectralElementGrid(topl, 4)
    G = M.grids
    x, y, z = (grid.vgeo[:, c, :] for c in (G._x1, G._x2, G._x3))
    a = x ** 2 + y ** 3 + z ** 2 * y ** 2 - x * y * z
    gx, gy, gz = 2 * x - y * z, 3 * y ** 2 + 2 * z ** 2 * y - x * z, 2 * z * y ** 2 - x * y
    zero = np.zeros_like(x)
    exact = {0: (gx, gy, gz), 1: (gx, gy, zero), 2: (zero, zero, gz)}[direction]
    exact = np.stack(exact, axis=1)
    g = G.auxiliary_field_gradient(grid, a, direction)
    assert np.linalg.norm(g - exact) <= np.sqrt(np.finfo(float).eps) * np.linalg.norm(exact)
    assert np.linalg.norm(g - exact) <= 1e-11 * np.linalg.norm(exact)
