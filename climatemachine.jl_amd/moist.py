"""Host-side description of the moist LES configuration of ``AtmosModel``: ``TotalEnergyModel``
with ``EquilMoist``, ``FlatOrientation``, ``HydrostaticState`` (``subtract_off``), closure constant
viscosity / ``SmagorinskyLilly`` / ``AnisoMinDiss``, source ``Gravity``, default ``AtmosBC``
(device functor ``csrc/physics_moist.h``).

Reference: ``src/Atmos/Model/AtmosModel.jl:397-520`` (layouts), ``moisture.jl:70-115``
(EquilMoist), ``thermo_states.jl``, ``tendencies_{mass,momentum,energy,moisture}.jl``,
``atmos_tendencies.jl``; ``src/Common/TurbulenceClosures/TurbulenceClosures.jl:411-497, 600-690``;
``test/Numerics/DGMethods/compressible_Navier_Stokes/density_current_model.jl``.

Thermodynamics.jl 0.3.2 and CLIMAParameters.jl 0.1.11 are not in the reference tree; the moist
formulas and constants restate their published definitions.  **Parity unpinned** for saturated
states (no file of the reference holds a number that exercises them); with ``q_tot = 0`` the
law reduces operation by operation to the dry one.

State ``rho, rho u(3), rho e, rho q_tot``; auxiliary ``coord(3), Phi, grad Phi(3), ref_state(7),
Delta, moisture(temperature, theta_v, q_liq, q_ice)``; gradient ``u(3), h_tot, theta_v, q_tot``;
gradient flux ``grad h_tot(3), S(6) | grad u(9), N^2, grad q_tot(3)``.
"""
import numpy as np

from .atmos import DryAtmosModel, PlanetParameters, ORIENT_FLAT, SRC_GRAVITY, BC_ATMOS_DEFAULT
from .balancelaws import PHYSICS_MOIST_ATMOS
from .mesh import grids as G

__all__ = ["MoistParameters", "MoistAtmosModel", "DensityCurrentSetup", "MoistBubbleSetup",
           "BomexSetup", "bomex_model", "SRC_BOMEX_TENDENCIES", "SRC_BOMEX_SPONGE",
           "SRC_BOMEX_GEOSTROPHIC", "BC_BOMEX_SURFACE",
           "CLOSURE_CONSTANT", "CLOSURE_SMAGORINSKY", "CLOSURE_ANISO_MIN_DISS"]

CLOSURE_CONSTANT, CLOSURE_SMAGORINSKY, CLOSURE_ANISO_MIN_DISS = 0, 1, 2
# source bits next to SRC_GRAVITY = 1 and the boundary kind of the BOMEX surface.  With
# u_slope = 0 BomexSponge is the library's RayleighSponge (relaxation to (u_geo, v_geo, 0)) and
# BomexGeostrophic its GeostrophicForcing (tendencies_momentum.jl:74-130): same formulas.
SRC_BOMEX_TENDENCIES, SRC_BOMEX_SPONGE, SRC_BOMEX_GEOSTROPHIC = 2, 4, 8
BC_BOMEX_SURFACE = 2


class MoistParameters(PlanetParameters):
    """CLIMAParameters.jl 0.1.11 ``Planet`` values of the moist thermodynamics."""
    molmass_water = 18.01528e-3
    cp_v, cp_l, cp_i = 1859.0, 4181.0, 2100.0
    LH_v0, LH_s0 = 2.5008e6, 2.8344e6
    T_triple, T_freeze, T_icenuc = 273.16, 273.15, 233.0
    press_triple = 611.657
    T_min = 150.0

    @property
    def R_v(self):
        return self.gas_constant / self.molmass_water

    # -- mixture properties, vectorised (host-side initial conditions and tests)
    def cv_m(self, qt, ql=0.0, qi=0.0):
        cv_v = self.cp_v - self.R_v
        return self.cv_d + (cv_v - self.cv_d) * qt + (self.cp_l - cv_v) * ql + (self.cp_i - cv_v) * qi

    def internal_energy(self, T, qt, ql=0.0, qi=0.0):
        e_v0 = self.LH_v0 - self.R_v * self.T_0
        e_i0 = self.LH_s0 - self.LH_v0
        return self.cv_m(qt, ql, qi) * (T - self.T_0) + (qt - ql) * e_v0 - qi * (e_v0 + e_i0)


class BomexSetup:
    """``init_bomex!`` of experiments/AtmosLES/bomex_model.jl:252-345: piecewise-linear
    ``theta_liq`` and ``q_tot`` profiles, zonal wind -8.75 m/s sheared above 700 m, pressure
    ``P_sfc exp(-z / H)``; the thermodynamic state comes from ``PhaseEquil_pthetaq`` (restated:
    ``theta_liq_ice(T, p, q_equil(T, p)) = theta`` solved for ``T`` by bisection)."""

    def __init__(self, ps):
        self.ps = ps

    def thermo_from_p_theta_q(self, p, theta, qt):
        ps = self.ps

        def state(T):
            # equilibrium partition at (T, p): rho depends on the partition, two sweeps suffice
            ql = np.zeros_like(T)
            qi = np.zeros_like(T)
            for _ in range(3):
                Rm = ps.R_d * (1 + (ps.R_v / ps.R_d - 1) * qt - ps.R_v / ps.R_d * (ql + qi))
                rho = p / (Rm * T)
                lam = np.clip((T - ps.T_icenuc) / (ps.T_freeze - ps.T_icenuc), 0.0, 1.0)
                LH0 = lam * ps.LH_v0 + (1 - lam) * ps.LH_s0
                dcp = lam * (ps.cp_v - ps.cp_l) + (1 - lam) * (ps.cp_v - ps.cp_i)
                pvs = ps.press_triple * (T / ps.T_triple) ** (dcp / ps.R_v) * np.exp(
                    (LH0 - dcp * ps.T_0) / ps.R_v * (1 / ps.T_triple - 1 / T))
                qc = np.maximum(qt - pvs / (rho * ps.R_v * T), 0.0)
                ql, qi = lam * qc, (1 - lam) * qc
            cpm = ps.cp_d + (ps.cp_v - ps.cp_d) * qt + (ps.cp_l - ps.cp_v) * ql + (ps.cp_i - ps.cp_v) * qi
            Lv = ps.LH_v0 + (ps.cp_v - ps.cp_l) * (T - ps.T_0)
            Ls = ps.LH_s0 + (ps.cp_v - ps.cp_i) * (T - ps.T_0)
            th = T / (p / ps.MSLP) ** (Rm / cpm) * (1 - (Lv * ql + Ls * qi) / (cpm * T))
            return th, rho, ql, qi

        lo, hi = 150.0 + 0 * theta, 400.0 + 0 * theta
        for _ in range(80):
            mid = (lo + hi) / 2
            th = state(mid)[0]
            lo = np.where(th < theta, mid, lo)
            hi = np.where(th < theta, hi, mid)
        T = (lo + hi) / 2
        _, rho, ql, qi = state(T)
        return T, rho, ql, qi

    def __call__(self, law, aux, coord, t):
        ps = self.ps
        z = coord[2]
        P_sfc, qg, T_sfc = 1.015e5, 22.45e-3, 300.4
        Rm_sfc = ps.R_d * (1 + (ps.R_v / ps.R_d - 1) * qg)
        zl1, zl2, zl3, zl4 = 520.0, 1480.0, 2000.0, 3000.0
        th = np.where(z <= zl1, 298.7,
             np.where(z <= zl2, 298.7 + (z - zl1) * (302.4 - 298.7) / (zl2 - zl1),
             np.where(z <= zl3, 302.4 + (z - zl2) * (308.2 - 302.4) / (zl3 - zl2),
                      308.2 + (z - zl3) * (311.85 - 308.2) / (zl4 - zl3))))
        qt = np.where(z <= zl1, 17.0 + (z / zl1) * (16.3 - 17.0),
             np.where(z <= zl2, 16.3 + (z - zl1) * (10.7 - 16.3) / (zl2 - zl1),
             np.where(z <= zl3, 10.7 + (z - zl2) * (4.2 - 10.7) / (zl3 - zl2),
                      4.2 + (z - zl3) * (3.0 - 4.2) / (zl4 - zl3)))) / 1000
        u = np.where(z <= 700.0, -8.75, -8.75 + (z - 700.0) * (-4.61 + 8.75) / (zl4 - 700.0))
        P = P_sfc * np.exp(-z / (Rm_sfc * T_sfc / ps.grav))
        T, rho, ql, qi = self.thermo_from_p_theta_q(P, th, qt)
        e_int = ps.internal_energy(T, qt, ql, qi)
        rhoe = rho * (0.5 * u * u + ps.grav * z + e_int)
        zero = 0.0 * rho
        return rho, [rho * u, zero, zero], rhoe, rho * qt


def bomex_model(zmax=3000.0, param_set=None, closure=CLOSURE_SMAGORINSKY):
    """``bomex_model(FT, config_type, zmax, "prescribed")`` (bomex_model.jl:347-470): EquilMoist
    (maxiter 5, tolerance 0.1), SmagorinskyLilly(0.23), the default reference state
    HydrostaticState(DecayingTemperatureProfile(param_set)), sources Gravity + BomexTendencies +
    BomexSponge + BomexGeostrophic, surface Impenetrable(DragLaw(u_star = 0.28)) with prescribed
    energy (LHF + SHF) and moisture (LHF / L_v(T_sfc)) fluxes, default AtmosBC at the top."""
    from .atmos import DecayingTemperatureProfile
    ps = param_set or MoistParameters()
    ref = DecayingTemperatureProfile(ps, 290.0, 220.0, ps.R_d * 290.0 / ps.grav)
    law = MoistAtmosModel(BomexSetup(ps), ref, closure=closure, coefficient=0.23,
                          sources=SRC_GRAVITY | SRC_BOMEX_TENDENCIES | SRC_BOMEX_SPONGE
                          | SRC_BOMEX_GEOSTROPHIC,
                          boundary_conditions=(BC_BOMEX_SURFACE, BC_ATMOS_DEFAULT), param_set=ps,
                          maxiter=5, tolerance=0.1)
    LHF, SHF, T_sfc = 147.2, 9.5, 300.4
    Lv = ps.LH_v0 + (ps.cp_v - ps.cp_l) * (T_sfc - ps.T_0)
    law.bomex = dict(u_star=0.28, e_flux=LHF + SHF, q_flux=LHF / Lv, f_coriolis=0.376e-4,
                     u_geostrophic=-10.0, u_slope=1.8e-3, v_geostrophic=0.0, z_sponge=2400.0,
                     alpha_max=0.75, gamma=2.0, z_max=float(zmax), dqt_peak=-1.2e-8,
                     zl_moisture=300.0, zh_moisture=500.0, dtheta_peak=-2 / ps.day, zl_sub=1500.0,
                     zh_sub=2100.0, w_sub=-0.65e-2)
    return law


class DensityCurrentSetup:
    """``Initialise_Density_Current!`` of density_current_model.jl:54-99 (Straka et al. 1993):
    cold bubble ``theta_c = -15 K`` of radii (4 km, 2 km) at (0, 3 km) in a neutral atmosphere
    ``theta = 300 K``; dry (``q_tot = 0``), at rest."""

    def __init__(self, ps):
        self.ps = ps

    def __call__(self, law, aux, coord, t):
        ps = self.ps
        x1, x3 = coord[0], coord[2]
        r = np.sqrt((x1 - 0.0) ** 2 / 4000.0 ** 2 + (x3 - 3000.0) ** 2 / 2000.0 ** 2)
        dth = np.where(r <= 1, -15.0 * (1 + np.cos(np.pi * r)) / 2, 0.0)
        theta = 300.0 + dth
        pi_exner = 1.0 - ps.grav / (ps.cp_d * theta) * x3
        rho = ps.MSLP / (ps.R_d * theta) * pi_exner ** (ps.cv_d / ps.R_d)
        # PhaseEquil_rho_theta_q with q_tot = 0: T = theta (rho R_d theta / MSLP)^(R_d / cv_d)
        T = theta * (rho * ps.R_d * theta / ps.MSLP) ** (ps.R_d / ps.cv_d)
        e_int = ps.cv_d * (T - ps.T_0)
        zero = 0.0 * rho
        rhoe = rho * (e_int + zero + aux[:, law.off_phi, :])
        return rho, [zero, zero, zero], rhoe, zero


class MoistBubbleSetup:
    """A warm, moist bubble in the dry-adiabatic atmosphere of the rising-bubble experiment
    (experiments/TestCase/risingbubble.jl:22-91 with a total-water perturbation added): the
    test state for saturated thermodynamics (condensate forms where ``q_tot`` exceeds
    saturation).  Not a reference configuration."""

    def __init__(self, ps, xc=1000.0, zc=1000.0, rc=800.0, dtheta=2.0, q0=0.0, dq=0.02):
        self.ps, self.xc, self.zc, self.rc, self.dtheta, self.q0, self.dq = ps, xc, zc, rc, dtheta, q0, dq

    def __call__(self, law, aux, coord, t):
        ps = self.ps
        x, z = coord[0], coord[2]
        r = np.sqrt((x - self.xc) ** 2 + (z - self.zc) ** 2)
        w = np.where(r <= self.rc, 1.0 - r / self.rc, 0.0)
        theta = 300.0 + self.dtheta * w
        qt = self.q0 + self.dq * w
        pi_exner = 1.0 - ps.grav / (ps.cp_d * theta) * z
        rho = ps.MSLP / (ps.R_d * theta) * pi_exner ** (ps.cv_d / ps.R_d)
        T = theta * pi_exner
        e_int = ps.internal_energy(T, qt)          # all vapour: the adjustment condenses the excess
        u = 5.0 * np.sin(np.pi * z / 2000.0)
        rhoe = rho * (e_int + u * u / 2 + aux[:, law.off_phi, :])
        zero = 0.0 * rho
        return rho, [rho * u, zero, zero], rhoe, rho * qt


class IsentropicVortexMoistSetup:
    """The isentropic vortex of test/Numerics/DGMethods/Euler/isentropicvortex.jl with
    ``moisture = EquilMoist()`` (:335-339): the dry state plus ``rho q_tot = 0``."""

    def __init__(self, ps):
        from .atmos import IsentropicVortexSetup
        self.dry = IsentropicVortexSetup(ps)
        for k in ("domain_halflength", "translation_speed", "T_inf"):
            setattr(self, k, getattr(self.dry, k))

    def __call__(self, law, aux, coord, t):
        rho, rhou, rhoe = self.dry(law, aux, coord, t)
        return rho, rhou, rhoe, 0.0 * rho


class MoistAtmosModel:
    physics_id = PHYSICS_MOIST_ATMOS
    off_phi, off_ref, off_turb, off_moist = 3, 7, 14, 15
    ns, naux, ngrad, ngradlap, nhyper = 6, 19, 6, 0, 0

    def __init__(self, init_state, ref_state, closure=CLOSURE_SMAGORINSKY, coefficient=None,
                 kinematic=True, subtract_off=True, sources=SRC_GRAVITY,
                 boundary_conditions=(BC_ATMOS_DEFAULT, BC_ATMOS_DEFAULT), param_set=None,
                 maxiter=8, tolerance=1e-1, no_orientation=False):
        self.ps = param_set or MoistParameters()
        # NoOrientation + NoReferenceState (the moist isentropic vortex): the potential, its
        # gradient, the reference state and Delta keep their slots of the auxiliary state and
        # stay zero -- Phi = 0, nothing to subtract, no filter width
        self.no_orientation = bool(no_orientation)
        self.init_state, self.ref_state = init_state, ref_state
        self.closure = int(closure)
        if coefficient is None:
            coefficient = {CLOSURE_CONSTANT: 0.0, CLOSURE_SMAGORINSKY: self.ps.C_smag,
                           CLOSURE_ANISO_MIN_DISS: 1.0}[self.closure]
        self.coefficient, self.kinematic = float(coefficient), bool(kinematic)
        self.subtract_off, self.sources = bool(subtract_off), int(sources)
        self.boundary_conditions = tuple(boundary_conditions)
        # PhaseEquil(param_set, e_int, rho, q_tot, maxiter, tolerance): Thermodynamics.jl defaults
        self.maxiter, self.tolerance = int(maxiter), float(tolerance)
        self.ngradflux = 3 + (10 if self.closure == CLOSURE_ANISO_MIN_DISS else 7) + 3
        # the dry model with the same options builds the shared part of the auxiliary state
        self._dry = None if self.no_orientation else DryAtmosModel(
            None, orientation=ORIENT_FLAT, ref_state=ref_state, smagorinsky=self.ps.C_smag,
            param_set=self.ps)

    def state_names(self):
        return ["ρ", "ρu[1]", "ρu[2]", "ρu[3]", "energy.ρe", "moisture.ρq_tot"]

    def descriptor(self):
        ps = self.ps
        ip = np.zeros(16, dtype=np.int32)
        ip[0], ip[1], ip[2], ip[3] = self.closure, int(self.subtract_off), int(self.kinematic), self.maxiter
        ip[5], ip[6] = self.sources, len(self.boundary_conditions)
        for i, bc in enumerate(self.boundary_conditions):
            ip[7 + i] = bc
        dp = np.zeros(64)
        dp[0] = self.coefficient
        dp[2:10] = [ps.R_d, ps.cp_d, ps.cv_d, ps.T_0, ps.grav, ps.MSLP, ps.inv_Pr_turb, self.tolerance]
        dp[16:27] = [ps.R_v, ps.cp_v, ps.cp_l, ps.cp_i, ps.LH_v0, ps.LH_s0, ps.T_triple, ps.T_freeze,
                     ps.T_icenuc, ps.press_triple, ps.T_min]
        b = getattr(self, "bomex", None)
        if b:
            dp[32:50] = [b[k] for k in (
                "u_star", "e_flux", "q_flux", "f_coriolis", "u_geostrophic", "u_slope",
                "v_geostrophic", "z_sponge", "alpha_max", "gamma", "z_max", "dqt_peak",
                "zl_moisture", "zh_moisture", "dtheta_peak", "zl_sub", "zh_sub", "w_sub")]
        return ip, dp

    def init_state_auxiliary(self, grid):
        """coord, orientation, reference state (relative humidity 0: the moist entries stay 0)
        and Delta as for the dry model (atmos.py); the moisture block starts at 0 and is
        refreshed by the nodal update before it is read."""
        aux = np.zeros((grid.nelem, self.naux, grid.Np))
        if self.no_orientation:
            for d in range(3):
                aux[:, d, :] = grid.vgeo[:, 12 + d, :]
            return aux
        d = self._dry.init_state_auxiliary(grid)               # (.., 3 + 4 + 7 + 1 + 2, ..)
        aux[:, :15, :] = d[:, :15, :]
        return aux

    def init_state_prognostic(self, grid, aux, t):
        Q = np.zeros((grid.nelem, self.ns, grid.Np))
        coord = [aux[:, d, :] for d in range(3)]
        rho, rhou, rhoe, rhoq = self.init_state(self, aux, coord, t)
        Q[:, 0, :] = rho
        for d in range(3):
            Q[:, 1 + d, :] = rhou[d]
        Q[:, 4, :], Q[:, 5, :] = rhoe, rhoq
        return Q
