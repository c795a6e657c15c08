"""Order of accuracy of the low-storage Runge-Kutta steppers on the reference's scalar test
problem dq/dt = q cos t (test/Numerics/ODESolvers/ode_tests_convergence.jl:17-43 with the
expected orders of ode_tests_common.jl:11-21): the stage loop of ``dostep!`` + ``update!`` as the
oracle restates them, with LSRK54CarpenterKennedy and LSRK144NiegemannDiehlBusch coefficients;
on the GPU the same loop over ``cmdg_lsrk_update``."""
import types

import numpy as np
import pytest

from cmdg_loader import cm

FINAL, DTS, ATOL = 20.0, [2.0 ** -6, 2.0 ** -7], 0.17
Q0 = np.linspace(-1.0, 1.0, 303)


def tableaus(oracle):
    return {"LSRK54CarpenterKennedy": (oracle.RKA, oracle.RKB, oracle.RKC),
            "LSRK144NiegemannDiehlBusch": cm.odesolvers.LSRK144_COEFFICIENTS}


@pytest.mark.parametrize("method", ["LSRK54CarpenterKennedy", "LSRK144NiegemannDiehlBusch"])
def test_lsrk_order_oracle(oracle, method):
    rka, rkb, rkc = tableaus(oracle)[method]

    class Rhs:
        grid = types.SimpleNamespace(nreal=1)

        def __call__(self, dQ, Q, t, alpha, beta):       # rhs!(dQ, Q, p, t; increment = true)
            dQ[...] = alpha * (Q * np.cos(t)) + beta * dQ

    errors = []
    for dt in DTS:
        Q = Q0.reshape(1, 1, -1).copy()
        dQ = np.zeros_like(Q)
        n = int(round(FINAL / dt))
        for i in range(n):
            oracle.lsrk_step(Rhs(), Q, dQ, i * dt, dt, rka, rkb, rkc)
        errors.append(np.abs(Q.reshape(-1) - Q0 * np.exp(np.sin(FINAL))).max())
    rate = np.log2(errors[0] / errors[1])
    assert abs(rate - 4) <= ATOL, (errors, rate)


@pytest.mark.parametrize("method,order", [("SSPRK22Heuns", 2), ("SSPRK22Ralstons", 2),
                                          ("SSPRK33ShuOsher", 3), ("SSPRK34SpiteriRuuth", 3)])
def test_ssprk_order_oracle(oracle, method, order):
    rka, rkb, rkc = cm.odesolvers.SSPRK_COEFFICIENTS[method]

    class Rhs:
        grid = types.SimpleNamespace(nreal=1)

        def __call__(self, dQ, Q, t, alpha, beta):
            dQ[...] = alpha * (Q * np.cos(t)) + beta * dQ

    errors = []
    for dt in DTS:
        Q = Q0.reshape(1, 1, -1).copy()
        R, Qs = np.zeros_like(Q), np.zeros_like(Q)
        for i in range(int(round(FINAL / dt))):
            oracle.ssprk_step(Rhs(), Q, R, Qs, i * dt, dt, rka, rkb, rkc)
        errors.append(np.abs(Q.reshape(-1) - Q0 * np.exp(np.sin(FINAL))).max())
    rate = np.log2(errors[0] / errors[1])
    assert abs(rate - order) <= ATOL, (errors, rate)


@pytest.mark.parametrize("method,order", [("LS3NRK44Classic", 4), ("LS3NRK33Heuns", 3)])
def test_ls3n_order_oracle(oracle, method, order):
    rka, rkb, rkc = cm.odesolvers.LS3N_COEFFICIENTS[method]

    class Rhs:
        grid = types.SimpleNamespace(nreal=1)

        def __call__(self, dQ, Q, t, alpha, beta):
            dQ[...] = alpha * (Q * np.cos(t)) + beta * dQ

    errors = []
    for dt in DTS:
        Q = Q0.reshape(1, 1, -1).copy()
        dQ, dR = np.zeros_like(Q), np.zeros_like(Q)
        for i in range(int(round(FINAL / dt))):
            oracle.ls3n_step(Rhs(), Q, dQ, dR, i * dt, dt, rka, rkb, rkc)
        errors.append(np.abs(Q.reshape(-1) - Q0 * np.exp(np.sin(FINAL))).max())
    rate = np.log2(errors[0] / errors[1])
    assert abs(rate - order) <= ATOL, (errors, rate)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["LS3NRK44Classic", "LS3NRK33Heuns", "SSPRK33ShuOsher"])
def test_other_steppers_match_oracle_on_the_device(oracle, method):
    """cmdg_ls3n_step / cmdg_ssprk_step against the oracle's restatement on the advection-diffusion
    test problem."""
    import torch
    from helpers import pseudo1d_setup, rel_linf
    assert torch.cuda.is_available(), "no HIP device: the product path has no CPU fallback"
    law, grid, dt = pseudo1d_setup(Ne=3, N=4)
    odg = oracle.OracleDGModel(law, grid)
    dg = cm.dgmodel.DGModel(law, grid)
    Q0h = law.init_state_prognostic(grid, odg.state_auxiliary, 0.0)
    Qo = Q0h.copy()
    Q = torch.from_numpy(Q0h.copy()).cuda()
    torch.cuda.synchronize()
    if method.startswith("LS3N"):
        rka, rkb, rkc = cm.odesolvers.LS3N_COEFFICIENTS[method]
        dQ, dR = np.zeros_like(Qo), np.zeros_like(Qo)
        for i in range(3):
            oracle.ls3n_step(odg, Qo, dQ, dR, i * dt, dt, rka, rkb, rkc)
        solver = getattr(cm.odesolvers, method)(dg, Q, dt=dt)
    else:
        rka, rkb, rkc = cm.odesolvers.SSPRK_COEFFICIENTS[method]
        R, Qs = np.zeros_like(Qo), np.zeros_like(Qo)
        for i in range(3):
            oracle.ssprk_step(odg, Qo, R, Qs, i * dt, dt, rka, rkb, rkc)
        solver = getattr(cm.odesolvers, method)(dg, Q, dt=dt)
    solver.dostep(Q, nsteps=3)
    dg.synchronize()
    assert rel_linf(Q.cpu().numpy(), Qo) < 1e-12
    dg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["LSRK54CarpenterKennedy", "LSRK144NiegemannDiehlBusch"])
def test_lsrk_order_device_update_kernel(oracle, method):
    import ctypes as C
    import torch
    from helpers import pseudo1d_setup
    assert torch.cuda.is_available(), "no HIP device: the product path has no CPU fallback"
    rka, rkb, rkc = tableaus(oracle)[method]
    law, grid, _ = pseudo1d_setup(Ne=2, N=2)
    dg = cm.dgmodel.DGModel(law, grid)                  # any handle: update! is pointwise
    nq = grid.nreal * law.ns * grid.Np
    q0 = np.resize(Q0, nq)
    errors = []
    for dt in DTS:
        Q = torch.from_numpy(q0.copy()).cuda().reshape(grid.nelem, law.ns, grid.Np)
        dQ = torch.zeros_like(Q)
        n, ns = int(round(FINAL / dt)), len(rka)
        for i in range(n):
            for s in range(ns):
                dQ += Q * np.cos(i * dt + rkc[s] * dt)
                torch.cuda.current_stream().synchronize()
                cm._lib.check(dg.L.cmdg_lsrk_update(dg.handle, dQ.data_ptr(), Q.data_ptr(),
                                                    C.c_double(rka[(s + 1) % ns]),
                                                    C.c_double(rkb[s] * dt)), dg.handle)
                dg.synchronize()
        errors.append(np.abs(Q.cpu().numpy().reshape(-1) - q0 * np.exp(np.sin(FINAL))).max())
    rate = np.log2(errors[0] / errors[1])
    assert abs(rate - 4) <= ATOL, (errors, rate)
    dg.close()
