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
           "CLOSURE_CONSTANT", "CLOSURE_SMAGORINSKY", "CLOSURE_ANISO_MIN_DISS"]

CLOSURE_CONSTANT, CLOSURE_SMAGORINSKY, CLOSURE_ANISO_MIN_DISS = 0, 1, 2


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


class MoistAtmosModel:
    physics_id = PHYSICS_MOIST_ATMOS
    off_phi, off_ref, off_turb, off_moist = 3, 7, 14, 15
    ns, naux, ngrad, ngradlap, nhyper = 6, 19, 6, 0, 0

    def __init__(self, init_state, ref_state, closure=CLOSURE_SMAGORINSKY, coefficient=None,
                 kinematic=True, subtract_off=True, sources=SRC_GRAVITY,
                 boundary_conditions=(BC_ATMOS_DEFAULT, BC_ATMOS_DEFAULT), param_set=None,
                 maxiter=8, tolerance=1e-1):
        self.ps = param_set or MoistParameters()
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
        self._dry = DryAtmosModel(None, orientation=ORIENT_FLAT, ref_state=ref_state,
                                  smagorinsky=self.ps.C_smag, param_set=self.ps)

    def state_names(self):
        return ["ρ", "ρu[1]", "ρu[2]", "ρu[3]", "energy.ρe", "moisture.ρq_tot"]

    def descriptor(self):
        ps = self.ps
        ip = np.zeros(16, dtype=np.int32)
        ip[0], ip[1], ip[2], ip[3] = self.closure, int(self.subtract_off), int(self.kinematic), self.maxiter
        ip[5], ip[6] = self.sources, len(self.boundary_conditions)
        for i, bc in enumerate(self.boundary_conditions):
            ip[7 + i] = bc
        dp = np.zeros(32)
        dp[0] = self.coefficient
        dp[2:10] = [ps.R_d, ps.cp_d, ps.cv_d, ps.T_0, ps.grav, ps.MSLP, ps.inv_Pr_turb, self.tolerance]
        dp[16:27] = [ps.R_v, ps.cp_v, ps.cp_l, ps.cp_i, ps.LH_v0, ps.LH_s0, ps.T_triple, ps.T_freeze,
                     ps.T_icenuc, ps.press_triple, ps.T_min]
        return ip, dp

    def init_state_auxiliary(self, grid):
        """coord, orientation, reference state (relative humidity 0: the moist entries stay 0)
        and Delta as for the dry model (atmos.py); the moisture block starts at 0 and is
        refreshed by the nodal update before it is read."""
        d = self._dry.init_state_auxiliary(grid)               # (.., 3 + 4 + 7 + 1 + 2, ..)
        aux = np.zeros((grid.nelem, self.naux, grid.Np))
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
