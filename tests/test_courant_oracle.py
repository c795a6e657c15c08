"""Courant numbers and time-step selection: the oracle's restatement of ``courant`` /
``kernel_min_neighbor_distance!`` / ``kernel_local_courant!`` against the analytic values of
the reference's test/Numerics/DGMethods/courant.jl:137-200.  CPU only."""
import numpy as np
import pytest

from cmdg_loader import cm
from helpers import courant_test_setup, held_suarez_setup

M = cm.mesh
EVERY, HORZ, VERT = 0, 1, 2


def _expected(law, grid, setup, dt):
    ps = law.ps
    dx_h = M.grids.min_node_distance(grid, HORZ)
    dx_v = M.grids.min_node_distance(grid, VERT)
    speed = np.linalg.norm([150.0, 150.0, 0.0])
    rho = setup.p_inf / (ps.R_d * setup.T_inf)
    cs = np.sqrt(ps.cp_d / ps.cv_d * ps.R_d * setup.T_inf)
    return dict(c_h=dt * (speed + cs) / dx_h, c_v=dt * cs / dx_v,
                d_h=dt * (2.0 / rho) / dx_h ** 2, d_v=dt * (2.0 / rho) / dx_v ** 2)


def test_min_neighbor_distance_matches_host(oracle):
    law, grid, _ = courant_test_setup(Neh=3, Nev=2)
    og = oracle.OracleGrid(grid)
    for d in (EVERY, HORZ, VERT):
        pw = oracle.min_neighbor_distance(og, d)
        assert pw.shape == (grid.nreal, grid.Np)
        assert pw.min() == pytest.approx(M.grids.min_node_distance(grid, d), rel=1e-14)
    # LGL spacing of a 1/3-wide element at N = 4: end interval (1 - sqrt(3/7)) / 2 of the width
    assert oracle.min_neighbor_distance(og, HORZ).min() == pytest.approx(
        (1 - np.sqrt(3 / 7)) / 2 / 3, rel=1e-12)


def test_courant_numbers_match_reference_test(oracle):
    law, grid, setup = courant_test_setup()
    dg = oracle.OracleDGModel(law, grid)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    dt = 1 / 200
    exp = _expected(law, grid, setup, dt)
    C = oracle
    rtol = 1e-4      # courant.jl:158
    assert C.courant(C.NONDIFFUSIVE_COURANT, dg, Q, dt, 0.0, HORZ) == pytest.approx(exp["c_h"], rel=rtol)
    assert C.courant(C.NONDIFFUSIVE_COURANT, dg, Q, dt, 0.0, VERT) == pytest.approx(exp["c_v"], rel=rtol)
    rt = np.sqrt(np.finfo(float).eps)
    assert C.courant(C.DIFFUSIVE_COURANT, dg, Q, dt, 0.0, HORZ) == pytest.approx(exp["d_h"], rel=rt)
    assert C.courant(C.DIFFUSIVE_COURANT, dg, Q, dt, 0.0, VERT) == pytest.approx(exp["d_v"], rel=rt)
    # advective: vertical velocity is zero; horizontal is the translation speed at x = 1
    assert C.courant(C.ADVECTIVE_COURANT, dg, Q, dt, 0.0, VERT) < 1e-8   # k = DG grad(Phi) / g
    adv = C.courant(C.ADVECTIVE_COURANT, dg, Q, dt, 0.0, HORZ)
    assert adv == pytest.approx(dt * np.hypot(150.0, 150.0) / M.grids.min_node_distance(grid, HORZ),
                                rel=1e-12)
    # calculate_dt (DGMethods.jl:79-83): the Courant number of the returned dt is the target
    dt2 = C.calculate_dt(dg, Q, 0.4, 0.0, EVERY)
    assert C.courant(C.NONDIFFUSIVE_COURANT, dg, Q, dt2, 0.0, EVERY) == pytest.approx(0.4, rel=1e-13)


def test_courant_on_the_sphere_uses_the_radial_direction(oracle):
    law, grid, d, dd = held_suarez_setup(n_horz=2, n_vert=2)
    dg = oracle.OracleDGModel(law, grid, direction=d, diffusion_direction=dd)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    # purely radial flow: horizontal advective Courant number vanishes
    x = dg.state_auxiliary[:, 0:3, :]
    r = np.sqrt((x ** 2).sum(axis=1, keepdims=True))
    Q[:, 1:4, :] = Q[:, 0:1, :] * 3.0 * x / r
    C = oracle
    ch = C.courant(C.ADVECTIVE_COURANT, dg, Q, 1.0, 0.0, HORZ)
    cv = C.courant(C.ADVECTIVE_COURANT, dg, Q, 1.0, 0.0, VERT)
    dxv = M.grids.min_node_distance(grid, VERT)
    assert ch < 1e-3 * cv          # grad(Phi) is a DG gradient: radial to discretisation error
    assert cv == pytest.approx(3.0 / dxv, rel=2e-2)     # coarse sphere: |grad Phi| / g = 1 +- 0.5 %


def test_min_node_distance_on_the_reference_stretched_brick():
    """test/Numerics/Mesh/min_node_distance.jl:18-79 (dim = 3, N = 4, Float64): 10 x 10 x 4
    stacked brick whose upper half is stretched by two in each direction, so the smallest
    node spacings are those of the unstretched elements: dxi / (2 Neh) and dxi / (2 Nev)."""
    Neh, Nev, N = 10, 4, 4
    rng = [np.linspace(0.0, 1.0, Neh + 1), np.linspace(0.0, 1.0, Neh + 1),
           np.linspace(1.0, 2.0, Nev + 1)]
    topl = M.StackedBrickTopology(rng, periodicity=(False,) * 3)

    def warp(a, b, c):
        a = np.where(a >= 0.5, 0.5 + 2 * (a - 0.5), a)
        b = np.where(b >= 0.5, 0.5 + 2 * (b - 0.5), b)
        c = np.where(c >= 1.5, 1.5 + 2 * (c - 1.5), c)
        return a, b, c
    grid = M.DiscontinuousSpectralElementGrid(topl, N, meshwarp=warp)
    xi = M.elements.lglpoints(N)[0]
    dxi = xi[1] - xi[0]
    hmnd, vmnd = dxi / (2 * Neh), dxi / (2 * Nev)
    rel = np.sqrt(np.finfo(float).eps)          # the test's `isapprox`
    assert M.grids.min_node_distance(grid, 0) == pytest.approx(hmnd, rel=rel)
    assert M.grids.min_node_distance(grid, VERT) == pytest.approx(vmnd, rel=rel)
    assert M.grids.min_node_distance(grid, HORZ) == pytest.approx(hmnd, rel=rel)
    assert M.grids.min_node_distance(grid, HORZ) == pytest.approx(hmnd, rel=1e-13)
