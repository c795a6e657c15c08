"""Dry rising bubble (BASELINE config 2: experiments/TestCase/risingbubble.jl) in the oracle:
SmagorinskyLilly closure, hydrostatic reference state, LSRK144.  The reference pins this case
only through ``norm(Q_end) / norm(Q_0) ~ 1`` (atol 1.5e-3, risingbubble.jl:226-233); the
closure formulas are checked against a direct evaluation of TurbulenceClosures.jl:476-497.
CPU only, reduced mesh."""
import numpy as np
import pytest

from cmdg_loader import cm
from helpers import rising_bubble_setup

RKA, RKB, RKC = cm.odesolvers.LSRK144_COEFFICIENTS


def test_lsrk144_coefficients_are_consistent():
    # c_i = sum of the effective b's before stage i: order conditions of a 2N scheme
    A, B, Cc = RKA, RKB, RKC
    n = len(A)
    assert n == 14 and A[0] == 0.0 and Cc[0] == 0.0
    # integrate y' = 1: y(dt) = sum_i b_i * (1 + a_i * (1 + a_{i-1} * ...)) must be 1
    y, dy = 0.0, 0.0
    for s in range(n):
        dy = 1.0 + A[s] * dy
        y += B[s] * dy
        if s + 1 < n:
            assert y == pytest.approx(Cc[s + 1], abs=1e-12)
    assert y == pytest.approx(1.0, abs=1e-12)


def test_smagorinsky_viscosity_formula(oracle):
    law, grid = rising_bubble_setup(nx=2, nz=2)
    dg = oracle.OracleDGModel(law, grid)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    rng = np.random.default_rng(0)
    gf = dg.state_gradient_flux
    gf[:, 3:9, :] = 1e-2 * rng.standard_normal(gf[:, 3:9, :].shape)
    gf[:, 9, :] = 1e-4 * rng.standard_normal(gf[:, 9, :].shape)
    ps = law.ps
    S = gf[:, 3:9, :]
    norm2 = S[:, 0] ** 2 + 2 * S[:, 1] ** 2 + 2 * S[:, 2] ** 2 + S[:, 3] ** 2 + 2 * S[:, 4] ** 2 + S[:, 5] ** 2
    normS = np.sqrt(2 * norm2)
    Ri = gf[:, 9] / (normS ** 2 + np.spacing(normS))
    fb2 = np.sqrt(np.clip(1 - Ri * ps.inv_Pr_turb, 0, 1))
    Delta = dg.state_auxiliary[:, law.off_turb]
    nu0 = normS * (ps.C_smag * Delta) ** 2 + 1e-5
    k = dg.state_auxiliary[:, law.off_phi + 1: law.off_phi + 4] / ps.grav       # ~ (0, 0, 1)
    dk = nu0 * k[:, 0] + nu0 * k[:, 1] + nu0 * k[:, 2]
    nu = np.stack([(nu0 - k[:, d] * dk) + k[:, d] * dk * fb2 for d in range(3)], axis=1)
    dxv = oracle.min_neighbor_distance(dg.og, 2)
    dxh = oracle.min_neighbor_distance(dg.og, 1)
    nr = grid.nreal
    kk = np.moveaxis(k, 1, 0)
    nuv = (nu[:, 0] * kk[0] + nu[:, 1] * kk[1] + nu[:, 2] * kk[2])
    exp_v = (0.5 * nuv[:nr] / dxv ** 2).max()
    nuh = np.sqrt(sum((nu[:, d] - nuv * kk[d]) ** 2 for d in range(3)))
    exp_h = (0.5 * nuh[:nr] / dxh ** 2).max()
    assert oracle.courant(oracle.DIFFUSIVE_COURANT, dg, Q, 0.5, 0.0, 2) == pytest.approx(exp_v, rel=1e-13)
    assert oracle.courant(oracle.DIFFUSIVE_COURANT, dg, Q, 0.5, 0.0, 1) == pytest.approx(exp_h, rel=1e-13)
    assert (fb2 < 1).any() and (fb2 == 1).any()          # both branches of the clamp exercised


def test_rising_bubble_norm_ratio_and_buoyancy(oracle):
    law, grid = rising_bubble_setup(nx=6, nz=6)
    dg = oracle.OracleDGModel(law, grid)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    Q0 = Q.copy()
    # at rest the only forces are buoyancy and the bubble's pressure perturbation
    T = np.zeros_like(Q)
    dg(T, Q, 0.0)
    assert np.abs(T[:, 0]).max() == 0.0 and np.abs(T[:, 2]).max() < 1e-10
    rho, rho_ref = Q[:, 0], dg.state_auxiliary[:, law.off_ref]
    assert rho.min() > 0 and (rho < rho_ref - 1e-4).any()            # the bubble is lighter
    assert T[:, 3].max() > 0.05                                      # and accelerates upward
    dt = oracle.calculate_dt(dg, Q, 1.7)          # SolverConfiguration: Courant_number = 1.7
    assert 0.3 < dt < 0.6
    dQ = np.zeros_like(Q)
    t = 0.0
    for _ in range(60):
        oracle.lsrk_step(dg, Q, dQ, t, dt, RKA, RKB, RKC)
        t += dt
    n0 = np.sqrt(oracle.weighted_norm2_local(grid, Q0))
    n1 = np.sqrt(oracle.weighted_norm2_local(grid, Q))
    assert abs(n1 / n0 - 1) < 1.5e-3                                  # risingbubble.jl:233
    w = Q[:, 3] / Q[:, 0]
    assert 0.3 < w.max() < 5.0 and np.isfinite(Q).all()
    # Smagorinsky produced a non-trivial eddy viscosity where the bubble shears the flow
    assert oracle.courant(oracle.DIFFUSIVE_COURANT, dg, Q, dt) > 1e-6
