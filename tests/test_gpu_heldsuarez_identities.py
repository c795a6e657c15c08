"""The Held-Suarez pins of tests/test_heldsuarez_independent.py and
tests/test_hyperdiffusion_cross_law.py, through the C ABI on the device."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import held_suarez_setup  # noqa: E402
from hs_crosslaw import crosslaw_residual  # noqa: E402

pytestmark = pytest.mark.gpu


class _DeviceOp:
    """numpy-in / numpy-out call of a device DGModel."""

    def __init__(self, cm, torch, law, grid, d, dd, nf):
        self.torch = torch
        self.dg = cm.dgmodel.DGModel(law, grid, numerical_flux_first_order=nf, direction=d,
                                     diffusion_direction=dd)

    def __call__(self, T, Q, t, alpha, beta):
        torch = self.torch
        Qg = torch.from_numpy(np.ascontiguousarray(Q)).cuda()
        Tg = torch.from_numpy(np.ascontiguousarray(T)).cuda()
        torch.cuda.synchronize()
        self.dg(Tg, Qg, t, alpha, beta)
        T[...] = Tg.cpu().numpy()
        self.dg.close()


def test_atmos_hyperdiffusion_equals_scalar_law_sum_device(cm, torch):
    res, cond = crosslaw_residual(cm, lambda law, grid, d, dd, nf: _DeviceOp(cm, torch, law, grid, d, dd, nf))
    assert res[0] == 0.0
    for s in range(1, 5):
        assert cond[s] < 1e7, (s, cond)
        assert res[s] < 2e-15 * cond[s], (s, res, cond)


def _tendency(cm, torch, law, grid, Q0):
    op = _DeviceOp(cm, torch, law, grid, 0, 1, 0)
    T = np.zeros_like(Q0)
    op(T, Q0, 0.0, 1.0, 0.0)
    return T[:grid.nreal]


def test_coriolis_does_no_work_and_forcing_vanishes_in_equilibrium(cm, torch):
    """Source terms isolated as differences of device tendencies with the source bits of the
    parameter block switched: u . S_coriolis = 0 at every node; at rest with T = T_equil(p) the
    Held-Suarez forcing adds nothing at all."""
    import copy
    from test_heldsuarez_independent import julia_hs_coefficients, R_D, CV_D, T_0
    A = cm.atmos
    law, grid, _, _ = held_suarez_setup(2, 2)
    aux = law.init_state_auxiliary(grid)
    Q0 = law.init_state_prognostic(grid, aux, 0.0)
    rng = np.random.default_rng(4)
    Q0[:, 1:4] += Q0[:, 0:1] * 20.0 * rng.standard_normal(Q0[:, 1:4].shape)

    def with_sources(bits):
        l2 = copy.copy(law)
        l2.sources = bits
        return l2
    full = A.SRC_GRAVITY | A.SRC_CORIOLIS | A.SRC_HELD_SUAREZ
    T_all = _tendency(cm, torch, with_sources(full), grid, Q0)
    T_noc = _tendency(cm, torch, with_sources(A.SRC_GRAVITY | A.SRC_HELD_SUAREZ), grid, Q0)
    cor = (T_all - T_noc)[:, 1:4]
    u = (Q0[:, 1:4] / Q0[:, 0:1])[:grid.nreal]
    work = np.abs((u * cor).sum(axis=1))
    bound = np.sqrt((u ** 2).sum(axis=1)) * np.sqrt((cor ** 2).sum(axis=1))
    # the difference of two O(10) tendencies carries 1e-15 of rounding; Coriolis itself is 1e-3
    assert (work <= 1e-9 * bound + 1e-13).all(), float((work / (bound + 1e-300)).max())
    assert np.abs((T_all - T_noc)[:, [0, 4]]).max() <= 1e-12 * np.abs(T_all[:, 4]).max()

    # rest state in radiative equilibrium: rho = rho_ref, u = 0, T solves T = T_equil(rho R_d T)
    o, r = law.off_phi, law.off_ref
    Qe = np.zeros_like(Q0)
    for e in range(grid.nelem):
        for i in range(grid.Np):
            rho, T = aux[e, r, i], 250.0
            for _ in range(100):
                T = julia_hs_coefficients(rho * R_D * T, aux[e, 0:3, i])[2]
            Qe[e, 0, i] = rho
            Qe[e, 4, i] = rho * (CV_D * (T - T_0) + aux[e, o, i])
    T_hs = _tendency(cm, torch, with_sources(full), grid, Qe)
    T_nohs = _tendency(cm, torch, with_sources(A.SRC_GRAVITY | A.SRC_CORIOLIS), grid, Qe)
    k_T_max = 1 / (4 * 86400.0)
    scale = k_T_max * np.abs(Qe[:, 0]).max() * CV_D * 300.0          # the forcing's natural size
    assert np.abs(T_hs - T_nohs)[:, 0:4].max() == 0.0
    assert np.abs(T_hs - T_nohs)[:, 4].max() <= 1e-9 * scale
