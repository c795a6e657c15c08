"""Independent pins for the Held-Suarez-specific physics of the headline workload.

The reference stores no number for ``experiments/AtmosGCM/heldsuarez.jl`` and the oracle's
``physics_atmos.c`` is a twin of the device functor, so a misreading of the forcing, the
reference state, the Coriolis term or the DryBiharmonic scaling would be common to both.  This
file is a third restatement, numpy only, written from the Julia text alone --

  * ``held_suarez_forcing_coefficients`` / ``source(::Energy|::Momentum, ::HeldSuarezForcing)``
    ``experiments/AtmosGCM/heldsuarez.jl:106-172``
  * ``latitude`` ``src/Common/Orientations/Orientations.jl:178-179``, ``vertical_unit_vector``
    ``:73-80``, ``projection_tangential`` ``:82-99``
  * ``source(::Momentum, ::Coriolis)`` ``src/Atmos/Model/tendencies_momentum.jl:74-85``
  * ``DecayingTemperatureProfile`` ``src/Atmos/TemperatureProfiles/TemperatureProfiles.jl:133-155``
  * ``transform_post_gradient_laplacian!(::DryBiharmonic)``
    ``src/Common/TurbulenceClosures/TurbulenceClosures.jl:899-912``

-- evaluated at sample states and compared with the oracle's pointwise callbacks, plus
identities that need no restatement at all (they follow from Held & Suarez 1994 and from vector
algebra).  The device equals the oracle bit for bit on this workload
(``tests/test_gpu_parity.py::test_held_suarez_tendency_matches_oracle``, ``bench.py`` parity
block), which closes the chain; ``tests/test_gpu_heldsuarez_identities.py`` repeats the
identities through the C ABI.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import held_suarez_setup  # noqa: E402

# CLIMAParameters.jl 0.1.11 Planet values (published constants of the package)
R_D = 8.3144598 / 28.97e-3
KAPPA_D = 2 / 7
CP_D = R_D / KAPPA_D
CV_D = CP_D - R_D
T_0 = 273.16
GRAV = 9.81
OMEGA = 7.2921159e-5
MSLP = 1.01325e5
DAY = 86400.0
PLANET_RADIUS = 6.371e6


# ---------------------------------------------------------------------------------------
# the third restatement (Julia text -> numpy, scalar code on purpose)
def julia_hs_coefficients(p, coord):
    """heldsuarez.jl:116-155."""
    k_a = 1 / (40 * DAY)
    k_f = 1 / DAY
    k_s = 1 / (4 * DAY)
    dT_y, dth_z, T_equator, T_min, sig_b = 60.0, 10.0, 315.0, 200.0, 7 / 10
    phi = np.arcsin(coord[2] / np.sqrt(coord[0] ** 2 + coord[1] ** 2 + coord[2] ** 2))
    sig = p / MSLP
    exner_p = sig ** (R_D / CP_D)
    dsig = (sig - sig_b) / (1 - sig_b)
    height_factor = max(0.0, dsig)
    T_equil = (T_equator - dT_y * np.sin(phi) ** 2 - dth_z * np.log(sig) * np.cos(phi) ** 2) * exner_p
    T_equil = max(T_min, T_equil)
    k_T = k_a + (k_s - k_a) * height_factor * np.cos(phi) ** 4
    k_v = k_f * height_factor
    return k_v, k_T, T_equil


def julia_thermo(rho, rhou, rhoe, Phi):
    """PhaseDry of Thermodynamics.jl 0.3.2 through its published closed forms: internal energy
    from the total one, T = T_0 + e_int / cv_d, p = rho R_d T."""
    e_int = (rhoe - (rhou @ rhou) / (2 * rho) - rho * Phi) / rho
    T = T_0 + e_int / CV_D
    return T, rho * R_D * T


def julia_sources(rho, rhou, rhoe, coord, Phi, gradPhi, rho_ref):
    """Gravity (tendencies_momentum.jl:52-55 with the reference density subtracted, ref_state
    subtract_off), Coriolis (:74-85), HeldSuarezForcing (heldsuarez.jl:157-172)."""
    T, p = julia_thermo(rho, rhou, rhoe, Phi)
    k_v, k_T, T_equil = julia_hs_coefficients(p, coord)
    khat = gradPhi / GRAV
    gravity = -(rho - rho_ref) * gradPhi
    coriolis = -np.cross(np.array([0.0, 0.0, 2 * OMEGA]), rhou)
    friction = -k_v * (rhou - khat * (khat @ rhou))
    heating = -k_T * rho * CV_D * (T - T_equil)
    return gravity, coriolis, friction, heating


def julia_decaying_profile(z, T_virt_surf=290.0, T_min_ref=220.0, H_t=8e3):
    """TemperatureProfiles.jl:133-155."""
    H_sfc = R_D * T_virt_surf / GRAV
    zp = z / H_t
    th = np.tanh(zp)
    dTv = T_virt_surf - T_min_ref
    Tv = T_virt_surf - dTv * th
    dTvp = dTv / T_virt_surf
    p = -H_t * (zp + dTvp * (np.log(1 - dTvp * th) - np.log(1 + th) + zp))
    p /= H_sfc * (1 - dTvp ** 2)
    return Tv, MSLP * np.exp(p)


# ---------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def hs():
    from oracle import oracle as O
    O.build()
    law, grid, d, dd = held_suarez_setup(2, 2)
    ph = O.OraclePhysics(law, 0)
    aux = law.init_state_auxiliary(grid)
    return O, law, grid, ph, aux


def _oracle_source(ph, Q, aux, gf=None):
    fn = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                     C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_int)(
        ph.c.contents.source)
    S = np.zeros(5)
    gf = np.zeros(16) if gf is None else gf
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    Qc, ac = np.ascontiguousarray(Q), np.ascontiguousarray(aux)
    fn(ph.c.contents.p, dp(S), dp(Qc), dp(gf), dp(ac), 0.0, 0)
    return S


def _sample_states(law, grid, aux, n=200, seed=11):
    """(Q, aux) pairs at random nodes with winds, off-balance temperature and density."""
    rng = np.random.default_rng(seed)
    Q0 = law.init_state_prognostic(grid, aux, 0.0)
    out = []
    for _ in range(n):
        e, i = rng.integers(grid.nreal), rng.integers(grid.Np)
        Q = Q0[e, :, i].copy()
        a = aux[e, :, i].copy()
        Q[0] *= 1 + 0.05 * rng.standard_normal()
        Q[1:4] += Q[0] * 30.0 * rng.standard_normal(3)
        Q[4] *= 1 + 0.02 * rng.standard_normal()
        out.append((Q, a))
    return out


def test_constants_are_the_package_values(hs):
    """The parameter block handed to oracle and device carries CLIMAParameters' numbers."""
    _, law, *_ = hs
    ps = law.ps
    for mine, theirs in ((R_D, ps.R_d), (CP_D, ps.cp_d), (CV_D, ps.cv_d), (T_0, ps.T_0),
                         (GRAV, ps.grav), (OMEGA, ps.Omega), (MSLP, ps.MSLP), (DAY, ps.day),
                         (PLANET_RADIUS, ps.planet_radius)):
        assert abs(mine - theirs) <= 1e-15 * abs(mine), (mine, theirs)


def test_sources_equal_the_third_restatement(hs):
    """Gravity + Coriolis + Held-Suarez friction and heating of the oracle, at 200 perturbed
    states, against the numpy restatement of the Julia text."""
    _, law, grid, ph, aux = hs
    worst = 0.0
    for Q, a in _sample_states(law, grid, aux):
        S = _oracle_source(ph, Q, a)
        o = law.off_phi
        g, c, f, h = julia_sources(Q[0], Q[1:4], Q[4], a[0:3], a[o], a[o + 1:o + 4],
                                   a[law.off_ref])
        mom = g + c + f
        assert S[0] == 0.0
        scale = np.abs(g).max() + np.abs(c).max() + np.abs(f).max()
        worst = max(worst, np.abs(S[1:4] - mom).max() / scale, abs(S[4] - h) / abs(h))
    # pow / log / asin of libm against numpy's: a few ulp, amplified by T - T_equil
    assert worst < 1e-11, worst


def test_forcing_identities_of_held_suarez_1994(hs):
    """What the definition implies, without reading any code: the equilibrium temperature is
    315 K at the equatorial surface, 255 K at the polar surface and never below 200 K; the
    relaxation rate is 1 / (4 days) at the equatorial surface and 1 / (40 days) above
    sigma_b = 0.7 and at the poles; friction acts below sigma_b only, with 1 / day at the surface."""
    eq, pole = np.array([PLANET_RADIUS, 0.0, 0.0]), np.array([0.0, 0.0, PLANET_RADIUS])
    k_v, k_T, T_eq = julia_hs_coefficients(MSLP, eq)
    assert abs(T_eq - 315.0) < 1e-12 and abs(k_T * 4 * DAY - 1) < 1e-14 and abs(k_v * DAY - 1) < 1e-14
    k_v, k_T, T_eq = julia_hs_coefficients(MSLP, pole)
    assert abs(T_eq - 255.0) < 1e-9 and abs(k_T * 40 * DAY - 1) < 1e-12
    for sig in (0.7, 0.5, 0.1, 0.01):
        k_v, k_T, T_eq = julia_hs_coefficients(sig * MSLP, eq)
        assert k_v == 0.0 and abs(k_T * 40 * DAY - 1) < 1e-14 and T_eq >= 200.0
    assert julia_hs_coefficients(0.01 * MSLP, pole)[2] == 200.0


def test_oracle_forcing_vanishes_at_rest_in_equilibrium(hs):
    """u = 0 and T = T_equil(p): no friction, no heating; with rho = rho_ref no gravity term
    either -- the whole source is zero.  T_equil depends on p = rho R_d T_equil, so the
    equilibrium temperature is found by fixed-point iteration first."""
    _, law, grid, ph, aux = hs
    o = law.off_phi
    rng = np.random.default_rng(5)
    for _ in range(50):
        e, i = rng.integers(grid.nreal), rng.integers(grid.Np)
        a = aux[e, :, i].copy()
        rho = a[law.off_ref]
        T = 250.0
        for _ in range(200):
            T = julia_hs_coefficients(rho * R_D * T, a[0:3])[2]
        Q = np.array([rho, 0.0, 0.0, 0.0, rho * (CV_D * (T - T_0) + a[o])])
        S = _oracle_source(ph, Q, a)
        assert np.all(S[0:4] == 0.0)
        k_T = julia_hs_coefficients(rho * R_D * T, a[0:3])[1]
        assert abs(S[4]) <= 1e-10 * k_T * rho * CV_D * T       # T - T_equil: rounding only


def test_coriolis_does_no_work_and_friction_is_tangential(hs):
    """u . (Omega x u) = 0, and the Rayleigh friction has no vertical component: evaluated
    from oracle source differences (sources are selected by bits of the parameter block)."""
    O, law, grid, _, aux = hs
    import copy
    A = sys.modules[type(law).__module__]
    laws = {}
    for name, bits in (("all", A.SRC_GRAVITY | A.SRC_CORIOLIS | A.SRC_HELD_SUAREZ),
                       ("no_coriolis", A.SRC_GRAVITY | A.SRC_HELD_SUAREZ),
                       ("no_hs", A.SRC_GRAVITY | A.SRC_CORIOLIS)):
        l2 = copy.copy(law)
        l2.sources = bits
        laws[name] = O.OraclePhysics(l2, 0)
    o = law.off_phi
    for Q, a in _sample_states(law, grid, aux, n=100, seed=3):
        S_all = _oracle_source(laws["all"], Q, a)
        cor = S_all - _oracle_source(laws["no_coriolis"], Q, a)
        hsf = S_all - _oracle_source(laws["no_hs"], Q, a)
        u = Q[1:4] / Q[0]
        assert abs(u @ cor[1:4]) <= 1e-12 * np.linalg.norm(u) * np.linalg.norm(cor[1:4]) + 1e-300
        assert cor[3] == 0.0 or abs(cor[3]) < 1e-14 * np.abs(cor[1:4]).max()   # Omega is along z
        # k = grad Phi / g is the DG gradient of Phi, not normalised (Orientations.jl:73-80):
        # the friction is tangential up to 1 - |k|^2, the discretisation error of that gradient
        khat = a[o + 1:o + 4] / GRAV
        rhou = Q[1:4]
        tang = rhou - khat * (khat @ rhou)
        assert np.abs(np.cross(hsf[1:4], tang)).max() <= 1e-9 * np.linalg.norm(hsf[1:4]) * np.linalg.norm(tang)
        k_v = np.linalg.norm(hsf[1:4]) / np.linalg.norm(tang)
        assert abs(khat @ hsf[1:4]) <= 1.001 * abs(1 - khat @ khat) * k_v * abs(khat @ rhou) + 1e-20


def test_reference_state_is_hydrostatic_and_matches_the_profile(hs):
    """DecayingTemperatureProfile: dp/dz = -g p / (R_d T_v) (checked by a centred difference of
    the restated formula), T_v decays from 290 K to 220 K, p(0) = MSLP; and the auxiliary
    reference columns the device reads equal the restatement at the node altitudes."""
    _, law, grid, _, aux = hs
    for z in (0.0, 500.0, 5e3, 15e3, 29e3):
        Tv, p = julia_decaying_profile(z)
        h = 1.0
        dpdz = (julia_decaying_profile(z + h)[1] - julia_decaying_profile(z - h)[1]) / (2 * h)
        assert abs(dpdz + GRAV * p / (R_D * Tv)) <= 2e-8 * abs(dpdz)
    assert julia_decaying_profile(0.0) == (290.0, MSLP)
    assert abs(julia_decaying_profile(1e6)[0] - 220.0) < 1e-9
    o, r = law.off_phi, law.off_ref
    z = aux[:, o, :] / GRAV
    Tv, p = julia_decaying_profile(z)
    assert np.abs(aux[:, r + 1, :] - p).max() <= 1e-13 * p.max()
    assert np.abs(aux[:, r, :] - p / (R_D * Tv)).max() <= 1e-13


def test_drybiharmonic_coefficient(hs):
    """nu_4 = (Delta / 2)^4 / 2 / tau with tau = 8 h, applied to every entry of the gradient of
    the Laplacian (TurbulenceClosures.jl:899-912)."""
    _, law, grid, ph, aux = hs
    fn = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                     C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double)(
        ph.c.contents.post_gradient_laplacian)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rng = np.random.default_rng(2)
    for _ in range(20):
        e, i = rng.integers(grid.nreal), rng.integers(grid.Np)
        a = np.ascontiguousarray(aux[e, :, i])
        gl = rng.standard_normal(12)
        hyp, Q = np.zeros(12), np.ones(5)
        fn(ph.c.contents.p, dp(hyp), dp(gl), dp(Q), dp(a), 0.0)
        nu4 = (a[law.off_delta] / 2) ** 4 / 2 / (8 * 3600.0)
        assert np.abs(hyp - nu4 * gl).max() <= 4e-16 * np.abs(nu4 * gl).max()


def test_horizontal_length_scale_is_a_node_spacing(hs):
    """aux.hyperdiffusion.Delta = lengthscale_horizontal(geom) (Geometry.jl:129-151) is the local
    average horizontal node distance: on the 6 x 2 x 2 cubed sphere with N = 4 about a quarter of
    an element width, 2 pi r / (4 * 2) / 4 = 1.25e6 m, within the equiangular grid's distortion.
    (A reading of the stored x-xi columns instead of the inverse of the xi-x ones gives 1e-5 m.)"""
    _, law, grid, _, aux = hs
    d = aux[:grid.nreal, law.off_delta, :]
    nominal = 2 * np.pi * PLANET_RADIUS / (4 * 2) / 4
    assert 0.6 * nominal < d.min() and d.max() < 1.6 * nominal, (d.min(), d.max(), nominal)


def test_hyperdiffusion_gradient_argument_is_the_horizontal_velocity(hs):
    """u_h = (I - k k') u and h_tot = e_tot + R_d T (TurbulenceClosures.jl:875-889,
    AtmosModel.jl:625-690): gradient arguments 5..8 of the law."""
    _, law, grid, ph, aux = hs
    fn = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                     C.POINTER(C.c_double), C.c_double)(ph.c.contents.gradient_argument)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    o = law.off_phi
    for Q, a in _sample_states(law, grid, aux, n=50, seed=9):
        G = np.zeros(8)
        fn(ph.c.contents.p, dp(G), dp(np.ascontiguousarray(Q)), dp(np.ascontiguousarray(a)), 0.0)
        u = Q[1:4] / Q[0]
        khat = a[o + 1:o + 4] / GRAV
        u_h = u - khat * (khat @ u)
        T, _ = julia_thermo(Q[0], Q[1:4], Q[4], a[o])
        h_tot = Q[4] / Q[0] + R_D * T
        assert np.abs(G[0:3] - u).max() <= 1e-15 * np.abs(u).max()
        assert np.abs(G[4:7] - u_h).max() <= 1e-13 * np.abs(u).max()
        assert abs(G[3] - h_tot) <= 1e-14 * abs(h_tot) and G[7] == G[3]
        # horizontal up to 1 - |k|^2 (k is the unnormalised DG gradient of Phi over g)
        assert abs(khat @ G[4:7]) <= 1.001 * abs(1 - khat @ khat) * abs(khat @ u) + 1e-12
