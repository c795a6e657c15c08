"""Cubed-sphere golden values on the GPU: advection_sphere.jl levels 1-4 (LSRK144 and the two
SSPRK steppers) and
diffusion_hyperdiffusion_sphere.jl levels 1-3 (N = 3) through libcmdg.  ``-m gpu``."""
import json
import os

import numpy as np
import pytest

from helpers import advection_sphere_setup, diffusion_sphere_setup

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_values.json")))


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "no HIP device: the product path has no CPU fallback"
    return t


@pytest.mark.parametrize("level", [1, 2, 3, 4])
@pytest.mark.parametrize("problem", ["SolidBodyRotation", "ReversingDeformationalFlow"])
def test_sphere_advection_on_the_gpu(cm, torch, problem, level):
    law, grid, dt = advection_sphere_setup(level, problem=problem)
    dg = cm.dgmodel.DGModel(law, grid)
    Q = dg.init_ode_state(0.0)
    Qe = Q.clone()
    solver = cm.odesolvers.LSRK144NiegemannDiehlBusch(dg, Q, dt=dt)
    cm.odesolvers.solve(Q, solver, timeend=law.problem.finaltime)
    err = dg.euclidean_distance(Q, Qe)
    g = GOLD["advection_sphere"]
    exp = g[problem + "_LSRK144"][level - 1]
    assert abs(err - exp) <= g["rtol"] * exp, (err, exp)
    assert abs(err - exp) <= 1e-9 * exp
    dg.close()


@pytest.mark.parametrize("level", [1, 2, 3, 4])
@pytest.mark.parametrize("method", ["SSPRK33", "SSPRK34"])
@pytest.mark.parametrize("problem", ["SolidBodyRotation", "ReversingDeformationalFlow"])
def test_sphere_advection_ssprk_on_the_gpu(cm, torch, problem, method, level):
    g = GOLD["advection_sphere"]
    law, grid, dt = advection_sphere_setup(level, problem=problem, cfl=g["max_cfl"][method])
    dg = cm.dgmodel.DGModel(law, grid)
    Q = dg.init_ode_state(0.0)
    Qe = Q.clone()
    make = {"SSPRK33": cm.odesolvers.SSPRK33ShuOsher, "SSPRK34": cm.odesolvers.SSPRK34SpiteriRuuth}
    solver = make[method](dg, Q, dt=dt)
    cm.odesolvers.solve(Q, solver, timeend=law.problem.finaltime)
    err = dg.euclidean_distance(Q, Qe)
    exp = g["%s_%s" % (problem, method)][level - 1]
    assert abs(err - exp) <= g["rtol"] * exp, (err, exp)
    assert abs(err - exp) <= 1e-9 * exp
    dg.close()


@pytest.mark.parametrize("level", [1, 2, 3])
@pytest.mark.parametrize("hyper", [False, True])
def test_sphere_diffusion_on_the_gpu(cm, torch, hyper, level):
    law, grid, dt = diffusion_sphere_setup(level, hyper)
    dg = cm.dgmodel.DGModel(law, grid, numerical_flux_first_order=1, direction=0,
                            diffusion_direction=1)
    Q = dg.init_ode_state(0.0)
    solver = cm.odesolvers.LSRK54CarpenterKennedy(dg, Q, dt=dt)
    # whole steps in one library call, then the shortened last step (solve! semantics)
    n = int(np.floor(1.0 / dt + 1e-12))
    solver.dostep(Q, nsteps=n)              # advances solver.t as updatetime! does
    if solver.t < 1.0:
        solver.dostep(Q, nsteps=1, dt=1.0 - solver.t)
    dg.synchronize()
    Qe = dg.init_ode_state(1.0)
    err = dg.euclidean_distance(Q, Qe)
    g = GOLD["diffusion_hyperdiffusion_sphere"]
    exp = g["HyperDiffusion" if hyper else "Diffusion"][level - 1]
    assert abs(err - exp) <= g["rtol"] * exp, (err, exp)
    dg.close()
