"""``VerticalIntegralModel`` of the split-explicit ocean (src/Ocean/SplitExplicit/
VerticalIntegralModel.jl): the upward column integral of the 3-D velocity, and the analytic
barotropic state on the 2-D grid, against the reference's StateCheck tables at six times
(test/Ocean/SplitExplicit/test_vertical_integral_model.jl,
test/Ocean/refvals/test_vertical_integral_model_refvals.jl).  CPU: oracle; ``gpu``: libcmdg."""
import json
import os

import numpy as np
import pytest
from scipy.linalg import expm

from cmdg_loader import cm
from helpers import ocean_spindown_setup
from test_ocean_oracle import statecheck

M = cm.mesh
REF = json.load(open(os.path.join(os.path.dirname(__file__), "golden",
                                  "ocean_vertical_integral_refvals.json")))["tables"]
TIMES = {"initial": 0, "day": 86400, "month": 30 * 86400, "year": 365 * 86400,
         "decade": 10 * 365 * 86400, "century": 100 * 365 * 86400}


def _check(stats, row, digits=(12, 12, 0, 12), rtol=5e-12):
    for got, ref, p in zip(stats, row[2:], digits):
        if p == 0:
            continue
        tol = rtol * max(abs(ref), 1e-300)
        assert abs(got - ref) <= tol, (row[0], row[1], got, ref)


def _barotropic_2d(law, t):
    """ShallowWaterModel on the 2-D grid, ocean_init_state!(::SWModel, ::SimpleBox, ...)
    (simple_box_problem.jl:128-146): U = A1 sin(kx x), eta = A2 cos(kx x) with
    A = exp(M t) [1, 1]."""
    p = law.problem
    rng = [np.linspace(0.0, p.Lx, 6), np.linspace(0.0, p.Ly, 6)]
    grid = M.DiscontinuousSpectralElementGrid(M.BrickTopology(rng, periodicity=(True, True)), 4)
    x = grid.vgeo[:, 12, :]
    kx = 2 * np.pi / p.Lx
    gH = law.grav * p.H
    A = expm(np.array([[-law.nu_h * kx ** 2, gH * kx], [-kx, 0.0]]) * t) @ np.array([1.0, 1.0])
    return A[0] * np.sin(kx * x), A[1] * np.cos(kx * x)


@pytest.mark.parametrize("name", list(TIMES))
def test_vertical_integral_of_velocity_oracle(oracle, name):
    law, grid = ocean_spindown_setup()
    t = float(TIMES[name])
    Q = law.init_state_prognostic(grid, None, t)
    aux = np.zeros((grid.nelem, 2, grid.Np))
    aux[:, 0], aux[:, 1] = Q[:, 0], Q[:, 1]               # f!: A.int_x = u
    og = oracle.OracleGrid(grid)
    ilaw = oracle.integral_fields_law([(0, 0), (0, 1)], [1.0, 1.0], [0, 1], [0, 1], [0, 1], 4, 2)
    oracle.indefinite_stack_integral(ilaw, og, Q, aux)
    rows = {r[1]: r for r in REF[name]}
    # the analytic state is exp(M t) [1, 1] with |M t| ~ 1e3 after a month: scipy's and Julia's
    # matrix exponentials agree to ~1e-11 there, which bounds what the later tables can show
    rtol = 5e-12 if t <= 86400 else (1e-9 if t < 50 * 365 * 86400 else 1e-7)
    _check(statecheck(aux[: grid.nreal, 0]), rows["∫x[1]"], rtol=rtol)
    assert not aux[:, 1].any()                            # int_x[2] identically zero
    U, eta = _barotropic_2d(law, t)
    _check(statecheck(eta), rows["η"], rtol=rtol)
    _check(statecheck(U), rows["U[1]"], rtol=rtol)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["initial", "day", "century"])
def test_vertical_integral_of_velocity_gpu(name):
    import torch
    law, grid = ocean_spindown_setup()
    dg = cm.dgmodel.DGModel(law, grid)
    t = float(TIMES[name])
    Q = dg.init_ode_state(t)
    aux = Q[:, 0:2, :].contiguous()
    torch.cuda.synchronize()
    dg.indefinite_stack_integral(Q, aux, [(0, 0), (0, 1)], [0, 1])
    dg.synchronize()
    rows = {r[1]: r for r in REF[name]}
    _check(statecheck(aux.cpu().numpy()[: grid.nreal, 0]), rows["∫x[1]"],
           rtol=5e-12 if t <= 86400 else 1e-7)
    dg.close()
