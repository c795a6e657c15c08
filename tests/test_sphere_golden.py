"""The cubed-sphere mesh, metrics and the horizontal diffusion / hyperdiffusion operators
against the reference's stored errors: advection_sphere.jl (solid-body rotation, LSRK144) and
diffusion_hyperdiffusion_sphere.jl (l = 2 harmonic, LSRK54).  These pin the grid the
Held-Suarez workload runs on.  CPU only; the finer levels run in tests/test_gpu_sphere.py."""
import json
import os

import numpy as np
import pytest

from cmdg_loader import cm
from helpers import advection_sphere_setup, diffusion_sphere_setup

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_values.json")))
RKA, RKB, RKC = cm.odesolvers.LSRK144_COEFFICIENTS


@pytest.mark.parametrize("level", [1, 2, 3])
@pytest.mark.parametrize("problem", ["SolidBodyRotation", "ReversingDeformationalFlow"])
def test_sphere_advection_matches_reference_error(oracle, problem, level):
    """ReversingDeformationalFlow: the velocity is refreshed by the nodal
    update_auxiliary_state! at every stage time (advection_sphere.jl:76-101)."""
    law, grid, dt = advection_sphere_setup(level, problem=problem)
    tend = law.problem.finaltime
    dg = oracle.OracleDGModel(law, grid, nf_first=0)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    Qe = Q.copy()
    dQ = np.zeros_like(Q)
    t = 0.0
    while t < tend:                      # solve! with adjustfinalstep
        step = tend - t if t + dt > tend else dt
        oracle.lsrk_step(dg, Q, dQ, t, step, RKA, RKB, RKC)
        t += step
    err = np.sqrt(oracle.weighted_norm2_local(grid, Q, Qe))
    g = GOLD["advection_sphere"]
    exp = g[problem + "_LSRK144"][level - 1]
    assert abs(err - exp) <= g["rtol"] * exp
    assert abs(err - exp) <= 1e-10 * exp          # observed 4e-14 .. 5e-12
    M = grid.vgeo[: grid.nreal, 9, :]
    m0, m1 = np.sum(M * Qe[: grid.nreal, 0]), np.sum(M * Q[: grid.nreal, 0])
    assert abs(m1 - m0) / m0 <= 5e-14             # advection_sphere.jl:431


@pytest.mark.parametrize("level", [1, 2])
@pytest.mark.parametrize("method", ["SSPRK33", "SSPRK34"])
@pytest.mark.parametrize("problem", ["SolidBodyRotation", "ReversingDeformationalFlow"])
def test_sphere_advection_ssprk_matches_reference_error(oracle, problem, method, level):
    """The strong-stability-preserving steppers (advection_sphere.jl:313-317, 330-372)."""
    g = GOLD["advection_sphere"]
    law, grid, dt = advection_sphere_setup(level, problem=problem, cfl=g["max_cfl"][method])
    rka, rkb, rkc = cm.odesolvers.SSPRK_COEFFICIENTS[
        {"SSPRK33": "SSPRK33ShuOsher", "SSPRK34": "SSPRK34SpiteriRuuth"}[method]]
    tend = law.problem.finaltime
    dg = oracle.OracleDGModel(law, grid, nf_first=0)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    Qe = Q.copy()
    R, Qs = np.zeros_like(Q), np.zeros_like(Q)
    t = 0.0
    while t < tend:
        step = tend - t if t + dt > tend else dt
        oracle.ssprk_step(dg, Q, R, Qs, t, step, rka, rkb, rkc)
        t += step
    err = np.sqrt(oracle.weighted_norm2_local(grid, Q, Qe))
    exp = g["%s_%s" % (problem, method)][level - 1]
    assert abs(err - exp) <= g["rtol"] * exp, (err, exp)
    assert abs(err - exp) <= 1e-10 * exp


@pytest.mark.parametrize("hyper", [False, True])
def test_sphere_diffusion_matches_reference_error(oracle, hyper):
    law, grid, dt = diffusion_sphere_setup(1, hyper)
    dg = oracle.OracleDGModel(law, grid, nf_first=1, direction=0, diffusion_direction=1)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    t, _ = oracle.solve(dg, Q, dt, 1.0)
    Qe = law.init_state_prognostic(grid, dg.state_auxiliary, 1.0)
    err = np.sqrt(oracle.weighted_norm2_local(grid, Q, Qe))
    g = GOLD["diffusion_hyperdiffusion_sphere"]
    exp = g["HyperDiffusion" if hyper else "Diffusion"][0]
    assert abs(err - exp) <= g["rtol"] * exp
    assert abs(err - exp) <= 1e-10 * exp          # observed 2e-14 / 2e-13
