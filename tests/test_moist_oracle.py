"""Moist LES law (EquilMoist) in the oracle.  What pins it:
  * with q_tot = 0 it equals the dry law bit for bit (tendencies and gradient flux), so the dry
    golden values carry over to its structure;
  * test/Numerics/DGMethods/compressible_Navier_Stokes/density_current_model.jl:247 -- EquilMoist
    at q_tot = 0 with the AnisoMinDiss closure: norm(Q_10) / norm(Q_0) = 9.9999970927037096e-01;
  * the saturated branch (Thermodynamics.jl 0.3.2, not in the reference tree) is PARITY UNPINNED:
    only its self-consistency is checked here.
CPU only (~1.5 min, the density current is the reference's full 10 000-element mesh)."""
import ctypes as C

import numpy as np
import pytest

from cmdg_loader import cm
from helpers import density_current_setup, rising_bubble_setup

A, MO = cm.atmos, cm.moist


class DryAsMoist:
    """a dry initial condition handed to the moist law with q_tot = 0"""

    def __init__(self, lawd):
        self.lawd = lawd

    def __call__(self, law, aux, coord, t):
        rho, ru, re = self.lawd.init_state(self.lawd, aux[:, :self.lawd.naux], coord, t)
        return rho, ru, re, 0 * rho


def moist_twin_of_bubble(closure=MO.CLOSURE_SMAGORINSKY, **kw):
    lawd, grid = rising_bubble_setup(nx=3, ny=2, nz=3)
    ps = MO.MoistParameters()
    lawm = MO.MoistAtmosModel(DryAsMoist(lawd), A.DryAdiabaticProfile(ps, 300.0, 0.0),
                              closure=closure, param_set=ps, **kw)
    return lawd, lawm, grid


def test_moist_law_equals_dry_law_at_zero_moisture(oracle):
    lawd, lawm, grid = moist_twin_of_bubble()
    od, om = oracle.OracleDGModel(lawd, grid), oracle.OracleDGModel(lawm, grid)
    rng = np.random.default_rng(0)
    Q0 = lawd.init_state_prognostic(grid, od.state_auxiliary, 0.0)
    Q0[:, 1:4] += Q0[:, 0:1] * 3 * rng.standard_normal(Q0[:, 1:4].shape)
    Q0[:, 4] *= 1 + 1e-3 * rng.standard_normal(Q0[:, 4].shape)
    Qm = np.concatenate([Q0, np.zeros_like(Q0[:, :1])], axis=1)
    for alpha, beta in ((1.0, 0.0), (0.5, 2.0)):
        T0 = rng.standard_normal(Q0.shape)
        Td, Tm = T0.copy(), np.concatenate([T0, np.zeros_like(T0[:, :1])], axis=1)
        od(Td, Q0.copy(), 0.0, alpha, beta)
        om(Tm, Qm.copy(), 0.0, alpha, beta)
        assert np.array_equal(Tm[:, :5], Td) and not Tm[:, 5].any()
        assert np.array_equal(om.state_gradient_flux[:, :10], od.state_gradient_flux)
        assert not om.state_gradient_flux[:, 10:].any()
    # the refreshed moisture block: temperature, theta_v as the dry model's air_T, theta_v
    assert np.array_equal(om.state_auxiliary[:, 15], od.state_auxiliary[:, lawd.off_moist + 1])
    assert np.array_equal(om.state_auxiliary[:, 16], od.state_auxiliary[:, lawd.off_moist])
    assert not om.state_auxiliary[:, 17:19].any()


def test_saturation_adjustment_is_self_consistent(oracle):
    """unsaturated air: the dry-air temperature formula with the mixture's heat capacity;
    saturated air: e_int(T*, partition(T*)) = e_int to the tolerance, condensate > 0, and
    q_vap = q_tot - q_liq - q_ice equals the saturation value."""
    _, lawm, grid = moist_twin_of_bubble(maxiter=30, tolerance=1e-10)
    om = oracle.OracleDGModel(lawm, grid)
    L, ps = oracle.lib(), lawm.ps
    ql, qi, eb = C.c_double(), C.c_double(), C.c_double()
    rng = np.random.default_rng(3)
    nsat = 0
    for _ in range(200):
        T_true = rng.uniform(225.0, 305.0)
        rho = rng.uniform(0.4, 1.2)
        qt = rng.uniform(0.0, 0.03)
        e_int = float(ps.internal_energy(T_true, qt))        # all vapour at T_true
        T = L.orc_moist_saturation_adjustment(om.ph.c, e_int, rho, qt, C.byref(ql), C.byref(qi),
                                              C.byref(eb))
        if ql.value + qi.value == 0:
            assert T == ps.T_0 + (e_int - qt * (ps.LH_v0 - ps.R_v * ps.T_0)) / float(ps.cv_m(qt))
        else:
            nsat += 1
            assert T > T_true                                 # condensation releases latent heat
            assert abs(eb.value - e_int) <= 1e-7 * ps.cv_d
            assert 0 <= ql.value and 0 <= qi.value and ql.value + qi.value < qt
            if T > ps.T_freeze:
                assert qi.value == 0
            if T < ps.T_icenuc:
                assert ql.value == 0
    assert 50 < nsat < 190


def test_saturated_state_conserves_mass_and_water(oracle):
    """a bubble with condensate: the tendencies of rho and rho q_tot integrate to zero over the
    closed (periodic / impenetrable, impermeable) box."""
    lawd, grid = rising_bubble_setup(nx=4, ny=2, nz=4)
    ps = MO.MoistParameters()
    lawm = MO.MoistAtmosModel(MO.MoistBubbleSetup(ps), A.DryAdiabaticProfile(ps, 300.0, 0.0),
                              param_set=ps)
    om = oracle.OracleDGModel(lawm, grid)
    Q = lawm.init_state_prognostic(grid, om.state_auxiliary, 0.0)
    T = np.zeros_like(Q)
    om(T, Q, 0.0, 1.0, 0.0)
    assert om.state_auxiliary[:, 17].max() > 1e-3            # q_liq: the bubble is cloudy
    M = grid.vgeo[:grid.nreal, 9, :]
    for s in (0, 5):
        assert abs((M * T[:grid.nreal, s]).sum()) <= 1e-10 * (M * np.abs(T[:grid.nreal, s])).sum()
    assert np.isfinite(T).all()


def test_density_current_matches_reference_norm_ratio(oracle):
    law, grid, dt, nsteps = density_current_setup()
    dg = oracle.OracleDGModel(law, grid)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    eng0 = np.sqrt(oracle.weighted_norm2_local(grid, Q))
    dQ = np.zeros_like(Q)
    for i in range(nsteps):
        oracle.lsrk54_step(dg, Q, dQ, i * dt, dt)
    ratio = np.sqrt(oracle.weighted_norm2_local(grid, Q)) / eng0
    ref = 9.9999970927037096e-01                              # density_current_model.jl:247
    assert abs(ratio - ref) <= 1.5e-8 * ref                   # the test's `≈`
    # the number that carries information is the change: 2.907e-7, reproduced to 1e-4 of itself
    assert abs((1 - ratio) - (1 - ref)) <= 2e-4 * (1 - ref)
    assert not Q[:, 5].any()
