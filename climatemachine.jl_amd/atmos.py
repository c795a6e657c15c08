"""Host-side description of the dry ``AtmosModel`` configurations carried by libcmdg
(device functor: ``csrc/physics_atmos.h``).

Reference: ``src/Atmos/Model/AtmosModel.jl`` (vars_state :397-511, gradient argument
:625-690, wavespeed :808-828), ``tendencies_{mass,momentum,energy}.jl``,
``atmos_tendencies.jl``, ``energy.jl``, ``moisture.jl:47-62`` (DryModel auxiliary
``theta_v, air_T`` refreshed every RHS), ``ref_state.jl``, ``boundaryconditions.jl`` +
``bc_momentum.jl`` + ``bc_energy.jl`` (default ``AtmosBC``: Impenetrable(FreeSlip),
Insulating), ``src/Common/Orientations/Orientations.jl``,
``src/Common/TurbulenceClosures/TurbulenceClosures.jl:300-420`` (constant viscosity) and
``:846-912`` (DryBiharmonic), ``experiments/AtmosGCM/heldsuarez.jl`` (initial condition
:47-104, forcing :106-172, configuration :174-218),
``test/Numerics/DGMethods/Euler/isentropicvortex*.jl``.

Thermodynamics.jl 0.3.2 and CLIMAParameters.jl 0.1.11 are registered packages that are
not vendored in the reference tree; the dry closed forms and planet constants below
restate their published definitions and are pinned at this boundary by the isentropic
vortex golden value (``isentropicvortex.jl:105``).

State layout: prognostic ``rho, rho*u(3), rho*e``; auxiliary ``coord(3) [, Phi, grad
Phi(3)] [, ref rho, p, T, rho*e, rho*q_tot, rho*q_liq, rho*q_ice] [, Delta] , theta_v,
air_T``; gradient ``u(3), h_tot [, u_h(3), h_tot]``; gradient flux ``grad h_tot(3),
S(6)``; gradient-laplacian ``u_h(3), h_tot``; hyperdiffusive ``nu grad^3 u_h (9), nu
grad^3 h_tot (3)``.
"""
import numpy as np

from .balancelaws import PHYSICS_DRY_ATMOS
from .mesh import grids as G

__all__ = ["PlanetParameters", "DryAtmosModel", "IsentropicVortexSetup", "HeldSuarezSetup",
           "DecayingTemperatureProfile", "DryAdiabaticProfile", "RisingBubbleSetup",
           "CourantTestSetup", "MMSSetup"]


class PlanetParameters:
    """CLIMAParameters.jl 0.1.11 ``Planet`` / ``SubgridScale`` values used by the dry model."""
    gas_constant = 8.3144598
    molmass_dryair = 28.97e-3
    kappa_d = 2 / 7
    T_0 = 273.16
    grav = 9.81
    planet_radius = 6.371e6
    Omega = 7.2921159e-5
    MSLP = 1.01325e5
    day = 86400.0
    inv_Pr_turb = 3.0
    C_smag = 0.21

    @property
    def R_d(self):
        return self.gas_constant / self.molmass_dryair

    @property
    def cp_d(self):
        return self.R_d / self.kappa_d

    @property
    def cv_d(self):
        return self.cp_d - self.R_d


class DecayingTemperatureProfile:
    """``DecayingTemperatureProfile(T_virt_surf, T_min_ref, H_t)`` of
    Thermodynamics.TemperatureProfiles: returns ``(T_virt, p)`` at altitude ``z``."""

    def __init__(self, ps, T_virt_surf=290.0, T_min_ref=220.0, H_t=8e3):
        self.ps, self.Ts, self.Tm, self.Ht = ps, T_virt_surf, T_min_ref, H_t

    def __call__(self, z):
        ps = self.ps
        H_sfc = ps.R_d * self.Ts / ps.grav
        zp = z / self.Ht
        th = np.tanh(zp)
        dTv = self.Ts - self.Tm
        Tv = self.Ts - dTv * th
        dTvp = dTv / self.Ts
        p = -self.Ht * (zp + dTvp * (np.log(1 - dTvp * th) - np.log(1 + th) + zp))
        p = p / (H_sfc * (1 - dTvp ** 2))
        p = ps.MSLP * np.exp(p)
        return Tv, p


class DryAdiabaticProfile:
    """``DryAdiabaticProfile(T_surface, T_min_ref)`` of Thermodynamics.TemperatureProfiles
    (Thermodynamics.jl 0.3.2, not vendored): ``T = max(T_surface - Gamma z, T_min)`` with
    ``Gamma = g / cp_d``, ``p = MSLP (T / T_surface)^(g / (R_d Gamma))`` and an isothermal
    decay above the height where ``T_min`` is reached."""

    def __init__(self, ps, T_surface=300.0, T_min_ref=0.0):
        self.ps, self.T_surface, self.T_min_ref = ps, T_surface, T_min_ref

    def __call__(self, z):
        ps = self.ps
        Gamma = ps.grav / ps.cp_d
        T = np.maximum(self.T_surface - Gamma * z, self.T_min_ref)
        p = ps.MSLP * (T / self.T_surface) ** (ps.grav / (ps.R_d * Gamma))
        if self.T_min_ref > 0:
            z_top = (self.T_surface - self.T_min_ref) / Gamma
            H_min = ps.R_d * self.T_min_ref / ps.grav
            p = np.where(T == self.T_min_ref, p * np.exp(-(z - z_top) / H_min), p)
        return T, p


class RisingBubbleSetup:
    """``init_risingbubble!`` of experiments/TestCase/risingbubble.jl:22-91 (dry)."""

    def __init__(self, ps, theta_ref=300.0, xc=5000.0, zc=2000.0, rc=2000.0, theta_amplitude=2.0):
        self.ps, self.theta_ref = ps, theta_ref
        self.xc, self.zc, self.rc, self.amp = xc, zc, rc, theta_amplitude

    def __call__(self, law, aux, coord, t):
        ps = self.ps
        x, z = coord[0], coord[2]
        r = np.sqrt((x - self.xc) ** 2 + (z - self.zc) ** 2)
        dtheta = np.where(r <= self.rc, self.amp * (1.0 - r / self.rc), 0.0)
        theta = self.theta_ref + dtheta
        pi_exner = 1.0 - ps.grav / (ps.cp_d * theta) * z
        rho = ps.MSLP / (ps.R_d * theta) * pi_exner ** (ps.cv_d / ps.R_d)
        T = theta * pi_exner
        e_int = ps.cv_d * (T - ps.T_0)
        e_pot = aux[:, law.off_phi, :]
        rhoe = rho * (0.0 + e_pot + e_int)
        zero = 0.0 * rho
        return rho, [zero, zero, zero], rhoe


class IsentropicVortexSetup:
    """``IsentropicVortexSetup`` (isentropicvortex_setup.jl:3-60)."""

    def __init__(self, ps):
        self.ps = ps
        self.p_inf, self.T_inf = 1e5, 300.0
        self.rho_inf = self.p_inf / (ps.R_d * self.T_inf)
        self.translation_speed, self.translation_angle = 150.0, np.pi / 4
        self.vortex_speed, self.vortex_radius = 50.0, 1 / 200
        self.domain_halflength = 1 / 20

    def __call__(self, law, aux, coord, t):
        ps = self.ps
        a = self.translation_angle
        uinf = np.array([self.translation_speed * np.cos(a), self.translation_speed * np.sin(a), 0.0])
        L, R = self.domain_halflength, self.vortex_radius
        x = [coord[d] - uinf[d] * t for d in range(3)]
        x = [xd - np.floor((xd + L) / (2 * L)) * 2 * L for xd in x]
        r = np.sqrt(x[0] ** 2 + x[1] ** 2)
        du_x = -self.vortex_speed * x[1] / R * np.exp(-(r / R) ** 2 / 2)
        du_y = self.vortex_speed * x[0] / R * np.exp(-(r / R) ** 2 / 2)
        u = [uinf[0] + du_x, uinf[1] + du_y, uinf[2] + 0 * du_x]
        T = self.T_inf * (1 - ps.kappa_d * self.vortex_speed ** 2 / 2 * self.rho_inf / self.p_inf
                          * np.exp(-(r / R) ** 2))
        p = self.p_inf * (T / self.T_inf) ** (1.0 / ps.kappa_d)
        rho = p / (ps.R_d * T)
        e_kin = (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) / 2
        rhoe = rho * (e_kin + 0.0 + ps.cv_d * (T - ps.T_0))
        return rho, [rho * u[0], rho * u[1], rho * u[2]], rhoe


class MMSSetup:
    """``mms3_init_state!`` of test/Numerics/DGMethods/compressible_Navier_Stokes/
    mms_bc_atmos.jl:88-97: the manufactured solution (generating script mms_solution.jl:10-25)
    ``rho = c g + 3, rho u = rho (c g, c g, c h), rho e = c g + 100`` with ``c = cos(pi t)``,
    ``g = sin(pi x) cos(pi y) cos(pi z)``, ``h = sin(pi x) cos(pi y) sin(pi z)``."""

    def __init__(self, ps):
        self.ps = ps

    def __call__(self, law, aux, coord, t):
        x, y, z = coord
        c = np.cos(np.pi * t)
        g = np.sin(np.pi * x) * np.cos(np.pi * y) * np.cos(np.pi * z)
        h = np.sin(np.pi * x) * np.cos(np.pi * y) * np.sin(np.pi * z)
        rho = g * c + 3
        return rho, [rho * g * c, rho * g * c, rho * h * c], g * c + 100


class CourantTestSetup:
    """``initialcondition!`` of test/Numerics/DGMethods/courant.jl:31-60: isothermal air at
    ``T_inf``, ``p_inf`` moving with ``u = 150 x (1, 1, 0)``; the potential energy passed to
    ``total_energy`` is zero there."""

    def __init__(self, ps, p_inf=1e5, T_inf=300.0, translation_speed=150.0):
        self.ps, self.p_inf, self.T_inf, self.speed = ps, p_inf, T_inf, translation_speed

    def __call__(self, law, aux, coord, t):
        ps = self.ps
        u = [self.speed * coord[0], self.speed * coord[0], 0.0 * coord[0]]
        T = self.T_inf
        p = self.p_inf * (T / self.T_inf) ** (1.0 / ps.kappa_d)
        rho = p / (ps.R_d * T) + 0.0 * coord[0]
        e_kin = (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) / 2
        rhoe = rho * (e_kin + 0.0 + ps.cv_d * (T - ps.T_0))
        return rho, [rho * u[0], rho * u[1], rho * u[2]], rhoe


class HeldSuarezSetup:
    """``init_heldsuarez!`` (experiments/AtmosGCM/heldsuarez.jl:47-104)."""

    def __init__(self, ps):
        self.ps = ps

    def __call__(self, law, aux, coord, t):
        ps = self.ps
        a = ps.planet_radius
        z_t, lam_c, phi_c, d_0, V_p = 15e3, np.pi / 9, 2 * np.pi / 9, a / 6, 10.0
        nrm = np.sqrt(coord[0] ** 2 + coord[1] ** 2 + coord[2] ** 2)
        phi = np.arcsin(coord[2] / nrm)
        lam = np.arctan2(coord[1], coord[0])
        z = aux[:, law.off_phi, :] / ps.grav
        F_z = 1 - 3 * (z / z_t) ** 2 + 2 * (z / z_t) ** 3
        F_z = np.where(z > z_t, 0.0, F_z)
        arg = np.sin(phi) * np.sin(phi_c) + np.cos(phi) * np.cos(phi_c) * np.cos(lam - lam_c)
        d = a * np.arccos(np.clip(arg, -1.0, 1.0))
        c3 = np.cos(np.pi * d / 2 / d_0) ** 3
        s1 = np.sin(np.pi * d / 2 / d_0)
        ok = (0 < d) & (d < d_0) & (d != a * np.pi)
        with np.errstate(divide="ignore", invalid="ignore"):
            up = (-16 * V_p / 3 / np.sqrt(3) * F_z * c3 * s1
                  * (-np.sin(phi_c) * np.cos(phi) + np.cos(phi_c) * np.sin(phi) * np.cos(lam - lam_c))
                  / np.sin(d / a))
            vp = (16 * V_p / 3 / np.sqrt(3) * F_z * c3 * s1 * np.cos(phi_c) * np.sin(lam - lam_c)
                  / np.sin(d / a))
        up = np.where(ok, up, 0.0)
        vp = np.where(ok, vp, 0.0)
        wp = 0.0 * up
        sl, cl = np.sin(phi), np.cos(phi)
        so, co = np.sin(lam), np.cos(lam)
        u = [-so * up - sl * co * vp + cl * co * wp,
             co * up - sl * so * vp + cl * so * wp,
             cl * vp + sl * wp]
        e_kin = 0.5 * (u[0] * u[0] + u[1] * u[1] + u[2] * u[2])
        rho = aux[:, law.off_ref + 0, :]
        rhoe = aux[:, law.off_ref + 3, :] + rho * e_kin
        return rho, [rho * u[0], rho * u[1], rho * u[2]], rhoe


ORIENT_NONE, ORIENT_FLAT, ORIENT_SPHERICAL = 0, 1, 2
SRC_GRAVITY, SRC_CORIOLIS, SRC_HELD_SUAREZ, SRC_MMS = 1, 2, 4, 8
# AtmosBC() default (Impenetrable(FreeSlip), Insulating); InitStateBC (bc_initstate.jl) with the
# manufactured solution of MMSSetup
BC_NONE, BC_ATMOS_DEFAULT, BC_INIT_STATE_MMS = 0, 1, 2


class DryAtmosModel:
    """Dry compressible ``AtmosModel``: ``TotalEnergyModel``, ``DryModel``, constant
    viscosity or ``SmagorinskyLilly(C_smag)``, optional hydrostatic reference state,
    ``DryBiharmonic`` hyperdiffusion, sources ``Gravity / Coriolis / HeldSuarezForcing``."""
    physics_id = PHYSICS_DRY_ATMOS

    def __init__(self, init_state, orientation=ORIENT_SPHERICAL, ref_state=None,
                 subtract_off=True, viscosity=0.0, dynamic_viscosity=False,
                 hyperdiffusion_timescale=None, sources=0, boundary_conditions=(),
                 param_set=None, smagorinsky=None, discrete_hydrostatic_balance=False,
                 with_divergence=False, zero_enthalpy=False):
        # ConstantViscosity(..., WithDivergence()) (TurbulenceClosures.jl:290-370) and the
        # total_specific_enthalpy == 0 override of mms_bc_atmos.jl:50-51
        self.with_divergence, self.zero_enthalpy = bool(with_divergence), bool(zero_enthalpy)
        # ref_state.jl:150-175: rho_ref = -k . grad(p_ref) / (k . grad Phi) with the DG gradient of
        # the reference pressure.  Needs the operator (device or oracle); off = analytic density
        self.discrete_hydrostatic_balance = bool(discrete_hydrostatic_balance)
        self.C_smag = smagorinsky
        if smagorinsky is not None:
            assert hyperdiffusion_timescale is None and orientation != ORIENT_NONE
        self.ps = param_set or PlanetParameters()
        self.init_state = init_state
        self.orientation = orientation
        self.ref_state = ref_state
        self.subtract_off = bool(subtract_off)
        self.viscosity, self.dynamic_viscosity = float(viscosity), bool(dynamic_viscosity)
        self.tau_hyper = hyperdiffusion_timescale
        self.sources = int(sources)
        self.boundary_conditions = tuple(boundary_conditions)
        has_or = orientation != ORIENT_NONE
        has_ref = ref_state is not None
        has_hyp = hyperdiffusion_timescale is not None
        o = 3
        self.off_phi = o
        o += 4 if has_or else 0
        self.off_ref = o
        o += 7 if has_ref else 0
        has_smag = self.C_smag is not None
        self.off_turb = o                      # turbulence.Delta (AtmosModel.jl:494-511 order)
        o += 1 if has_smag else 0
        self.off_delta = o
        o += 1 if has_hyp else 0
        self.off_moist = o
        o += 2
        self.ns, self.naux = 5, o
        self.ngrad = 4 + (1 if has_smag else 0) + (4 if has_hyp else 0)
        self.ngradflux = 9 + (1 if has_smag else 0)
        self.ngradlap = 4 if has_hyp else 0
        self.nhyper = 12 if has_hyp else 0

    def descriptor(self):
        ps = self.ps
        ip = np.zeros(16, dtype=np.int32)
        ip[0] = self.orientation
        ip[1] = 1 if self.ref_state is not None else 0
        ip[2] = int(self.subtract_off)
        ip[3] = 0 if self.dynamic_viscosity else 1
        ip[4] = 1 if self.tau_hyper is not None else 0
        ip[5] = self.sources
        ip[6] = len(self.boundary_conditions)
        for i, bc in enumerate(self.boundary_conditions):
            ip[7 + i] = bc
        ip[14] = 1 if self.C_smag is not None else 0
        ip[15] = int(self.with_divergence) | (int(self.zero_enthalpy) << 1)
        dp = np.zeros(32)
        dp[0] = self.viscosity
        dp[1] = self.tau_hyper if self.tau_hyper is not None else 0.0
        dp[2:13] = [ps.R_d, ps.cp_d, ps.cv_d, ps.T_0, ps.grav, ps.Omega, ps.MSLP, ps.day,
                    ps.planet_radius, ps.inv_Pr_turb, ps.kappa_d]
        dp[13] = self.C_smag if self.C_smag is not None else 0.0
        return ip, dp

    def state_names(self):
        return ["ρ", "ρu[1]", "ρu[2]", "ρu[3]", "energy.ρe"]

    # -- init_state_auxiliary! (AtmosModel.jl:880-940) -------------------------------------
    def init_state_auxiliary(self, grid):
        ps = self.ps
        vg = grid.vgeo
        aux = np.zeros((grid.nelem, self.naux, grid.Np))
        x = [vg[:, c, :] for c in (G._x1, G._x2, G._x3)]
        for d in range(3):
            aux[:, d, :] = x[d]
        if self.orientation != ORIENT_NONE:
            if self.orientation == ORIENT_SPHERICAL:     # Orientations.jl:140-150
                phi = ps.grav * (np.sqrt(x[0] ** 2 + x[1] ** 2 + x[2] ** 2) - ps.planet_radius)
            else:                                         # FlatOrientation
                phi = ps.grav * x[2]
            aux[:, self.off_phi, :] = phi
            # auxiliary_field_gradient!: element-local strong gradient
            # (dgsem_auxiliary_field_gradient!, DGModel_kernels.jl:3097-3232)
            aux[:, self.off_phi + 1:self.off_phi + 4, :] = G.auxiliary_field_gradient(grid, phi)
        if self.ref_state is not None:
            # ref_state_init_p_rho! and ref_state_finalize_init! (ref_state.jl:70-140).  The
            # reference additionally re-derives rho from a DG gradient of p (discrete
            # hydrostatic balance, :150-175); the synthetic state here keeps the analytic
            # rho = p / (R_d T_v) -- documented in DESIGN.md.
            z = aux[:, self.off_phi, :] / ps.grav
            Tv, p = self.ref_state(z)
            rho = p / (Tv * ps.R_d)
            T = p / (ps.R_d * rho)
            o = self.off_ref
            aux[:, o + 0, :] = rho
            aux[:, o + 1, :] = p
            aux[:, o + 2, :] = T
            aux[:, o + 3, :] = rho * (0.0 + aux[:, self.off_phi, :] + ps.cv_d * (T - ps.T_0))
        if self.C_smag is not None:
            # init_aux_turbulence!: Delta = lengthscale(geom) = 2 / (cbrt(det(invJ)) max(N))
            # (TurbulenceClosures.jl:431-438, Geometry.jl:121-122)
            m = [[vg[:, c, :] for c in row] for row in (
                (G._xi1x1, G._xi1x2, G._xi1x3), (G._xi2x1, G._xi2x2, G._xi2x3),
                (G._xi3x1, G._xi3x2, G._xi3x3))]
            det = (m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1])
                   - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0])
                   + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]))
            aux[:, self.off_turb, :] = 2 / (np.cbrt(det) * max(max(1, n) for n in grid.N))
        if self.tau_hyper is not None:
            # lengthscale_horizontal (Geometry.jl:129-151): with invJ[i, j] = d xi_i / d x_j taken
            # from the xi-x columns of vgeo (LocalGeometry, Geometry.jl:78-85),
            #   Delta_1 = |invJ \ e_1| 2 / N_1,  Delta_2 = |invJ \ e_2| 2 / N_2.
            # (Not the stored x-xi columns: in this snapshot Metrics.jl:617-667 fills them with
            # adj * adj' / det, which is not the inverse, and nothing on this path reads them.
            # tests/test_hyperdiffusion_cross_law.py would see the difference.)
            N = grid.N
            invJ = np.stack([np.stack([vg[:, c] for c in row], axis=-1) for row in (
                (G._xi1x1, G._xi1x2, G._xi1x3), (G._xi2x1, G._xi2x2, G._xi2x3),
                (G._xi3x1, G._xi3x2, G._xi3x3))], axis=-2)          # (nelem, Np, 3, 3)
            J = np.linalg.inv(invJ)
            c1 = np.sqrt((J[..., :, 0] ** 2).sum(axis=-1))
            c2 = np.sqrt((J[..., :, 1] ** 2).sum(axis=-1))
            aux[:, self.off_delta, :] = (c1 * 2 / N[0] + c2 * 2 / N[1]) / 2
        return aux

    def rebalance_reference_state(self, grid, aux, gradp):
        """``ref_state_init_density_from_pressure!`` + ``ref_state_finalize_init!``
        (ref_state.jl:83-135) on the real elements: the density that balances the DG gradient
        of the reference pressure, then T and rho e consistent with (rho, p)."""
        ps, nr = self.ps, grid.nreal
        gP = aux[:nr, self.off_phi + 1:self.off_phi + 4, :]
        k = gP / ps.grav
        num = -(k[:, 0] * gradp[:nr, 0] + k[:, 1] * gradp[:nr, 1] + k[:, 2] * gradp[:nr, 2])
        den = k[:, 0] * gP[:, 0] + k[:, 1] * gP[:, 1] + k[:, 2] * gP[:, 2]
        rho = num / den
        o = self.off_ref
        p = aux[:nr, o + 1, :]
        T = p / (ps.R_d * rho)
        aux[:nr, o + 0, :] = rho
        aux[:nr, o + 2, :] = T
        aux[:nr, o + 3, :] = rho * (0.0 + aux[:nr, self.off_phi, :] + ps.cv_d * (T - ps.T_0))

    def init_state_prognostic(self, grid, aux, t):
        Q = np.zeros((grid.nelem, self.ns, grid.Np))
        coord = [aux[:, d, :] for d in range(3)]
        rho, rhou, rhoe = self.init_state(self, aux, coord, t)
        Q[:, 0, :] = rho
        for d in range(3):
            Q[:, 1 + d, :] = rhou[d]
        Q[:, 4, :] = rhoe
        return Q
