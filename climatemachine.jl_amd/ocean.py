"""Host-side mirror of the hydrostatic Boussinesq ocean model (uncoupled).

Reference: ``HydrostaticBoussinesqModel`` ``src/Ocean/HydrostaticBoussinesq/
hydrostatic_boussinesq_model.jl:25-85`` (parameters), state / auxiliary layouts ``:107-144``,
``OceanBC`` ``src/Ocean/OceanBoundaryConditions`` with ``bc_velocity.jl`` / ``bc_temperature.jl``,
the ``SimpleBox`` problem and its analytic spin-down solution
``src/Ocean/OceanProblems/simple_box_problem.jl:95-270``, ``OceanBoxGCMConfiguration``
``src/Driver/driver_configs.jl:470-540``.

State ``u[2], eta, theta`` (4); auxiliary ``y, w, pkin, wz0, u_d[2], dG_u[2]`` (8); gradient
``grad u[2], grad u_d[2], grad theta`` (5); gradient flux ``div_h u, nu grad u (3x2), kappa grad
theta`` (10).  The law's ``update_auxiliary_state!`` (filters) and
``update_auxiliary_state_gradient!`` (column integrals) are installed on a ``DGModel`` with
:func:`install_hydrostatic_boussinesq_hooks`.
"""
import numpy as np
from scipy.linalg import expm

from .balancelaws import PHYSICS_HYDROSTATIC_BOUSSINESQ
from .mesh import filters as F

__all__ = ["HydrostaticBoussinesqModel", "SimpleBox", "OceanGyre", "HomogeneousBox", "OceanBC",
           "IMPENETRABLE_NOSLIP", "IMPENETRABLE_FREESLIP", "PENETRABLE_FREESLIP",
           "IMPENETRABLE_KINEMATIC_STRESS", "PENETRABLE_KINEMATIC_STRESS", "INSULATING",
           "TEMPERATURE_FLUX", "install_hydrostatic_boussinesq_hooks"]

IMPENETRABLE_NOSLIP, IMPENETRABLE_FREESLIP, PENETRABLE_FREESLIP = 1, 2, 3
IMPENETRABLE_KINEMATIC_STRESS, PENETRABLE_KINEMATIC_STRESS = 4, 5
INSULATING, TEMPERATURE_FLUX = 0, 1
FIXED, ROTATING, BETA_PLANE = 0, 1, 2


def OceanBC(velocity, temperature=INSULATING):
    """``OceanBC(velocity, temperature)`` as the integer code of the parameter block."""
    return int(velocity) + 8 * int(temperature)


class SimpleBox:
    """``SimpleBox{FT}(Lx, Ly, H; rotation, BC)``: default BCs are a free-slip insulating
    bottom (tag 1) and a penetrable free-slip insulating surface (tag 2)."""

    def __init__(self, Lx, Ly, H, rotation=FIXED,
                 BC=(OceanBC(IMPENETRABLE_FREESLIP), OceanBC(PENETRABLE_FREESLIP))):
        self.Lx, self.Ly, self.H = float(Lx), float(Ly), float(H)
        self.rotation = rotation
        self.boundary_conditions = tuple(BC)

    def init_state(self, m, x, y, z, t):
        """``ocean_init_state!`` (simple_box_problem.jl:170-190) with ``barotropic_state!`` /
        ``baroclinic_deviation`` of the ``Fixed`` box (:192-222)."""
        assert self.rotation == FIXED, "analytic solution restated for the non-rotating box"
        kx, kz = 2 * np.pi / self.Lx, 2 * np.pi / self.H
        gH = m.grav * self.H
        Mx = np.array([[-m.nu_h * kx ** 2, gH * kx], [-kx, 0.0]])
        A = expm(Mx * t) @ np.array([1.0, 1.0])
        Ub = A[0] * np.sin(kx * x)
        eta = A[1] * np.cos(kx * x)
        lam = m.nu_h * kx ** 2 + m.nu_z * kz ** 2
        u0 = np.exp(-lam * t) * np.cos(kz * z) * np.sin(kx * x)
        return u0 + Ub / self.H, 0.0 * x, eta, 0.0 * x


class OceanGyre:
    """``OceanGyre{FT}(Lx, Ly, H; tau_o, lambda_r, theta_E, BC)`` (ocean_gyre.jl:14-37): wind
    stress, beta-plane Coriolis force and temperature relaxation; BCs: no-slip insulating walls
    (tag 1) and bottom (tag 2), penetrable surface with kinematic stress and temperature flux
    (tag 3)."""
    rotation = BETA_PLANE

    def __init__(self, Lx, Ly, H, tau_o=1e-1, lambda_r=4 / 86400, theta_E=10.0,
                 BC=(OceanBC(IMPENETRABLE_NOSLIP), OceanBC(IMPENETRABLE_NOSLIP),
                     OceanBC(PENETRABLE_KINEMATIC_STRESS, TEMPERATURE_FLUX))):
        self.Lx, self.Ly, self.H = float(Lx), float(Ly), float(H)
        self.tau_o, self.lambda_r, self.theta_E = tau_o, lambda_r, theta_E
        self.boundary_conditions = tuple(BC)

    def init_state(self, m, x, y, z, t):
        """``ocean_init_state!`` (ocean_gyre.jl:51-66)."""
        th = (5 + 4 * np.cos(y * np.pi / self.Ly)) * (1 + z / self.H)
        zero = -0.0 * np.ones_like(x)
        return zero, zero.copy(), zero.copy(), th


class HomogeneousBox(OceanGyre):
    """``HomogeneousBox{FT}(Lx, Ly, H; tau_o, BC)`` (homogeneous_box.jl:15-54): wind stress on
    a box of constant temperature."""

    def __init__(self, Lx, Ly, H, tau_o=1e-1,
                 BC=(OceanBC(IMPENETRABLE_NOSLIP), OceanBC(IMPENETRABLE_NOSLIP),
                     OceanBC(PENETRABLE_KINEMATIC_STRESS, INSULATING))):
        OceanGyre.__init__(self, Lx, Ly, H, tau_o=tau_o, lambda_r=0.0, theta_E=0.0, BC=BC)

    def init_state(self, m, x, y, z, t):
        zero = 0.0 * x
        return zero, zero.copy(), zero.copy(), zero + 20.0


class HydrostaticBoussinesqModel:
    physics_id = PHYSICS_HYDROSTATIC_BOUSSINESQ
    ns, naux, ngrad, ngradflux, ngradlap, nhyper = 4, 8, 5, 10, 0, 0

    def __init__(self, problem, momentum_advection=False, tracer_advection=True, rho_o=1000.0,
                 c_h=0.0, c_z=0.0, alpha_T=2e-4, nu_h=5e3, nu_z=5e-3, kappa_h=1e3, kappa_z=1e-4,
                 kappa_c=1e-1, f_o=1e-4, beta=1e-11, grav=9.81):
        self.problem = problem
        self.momentum_advection, self.tracer_advection = bool(momentum_advection), bool(tracer_advection)
        self.rho_o, self.c_h, self.c_z, self.alpha_T = rho_o, c_h, c_z, alpha_T
        self.nu_h, self.nu_z, self.kappa_h, self.kappa_z, self.kappa_c = nu_h, nu_z, kappa_h, kappa_z, kappa_c
        self.f_o, self.beta, self.grav = f_o, beta, grav

    def state_names(self):
        return ["u[1]", "u[2]", "η", "θ"]

    def descriptor(self):
        ip = np.zeros(16, dtype=np.int32)
        ip[0], ip[1] = int(self.momentum_advection), int(self.tracer_advection)
        ip[2] = self.problem.rotation
        bcs = self.problem.boundary_conditions
        ip[6] = len(bcs)
        for i, bc in enumerate(bcs):
            ip[7 + i] = bc
        dp = np.zeros(32)
        dp[0:11] = [self.grav, self.c_h, self.c_z, self.alpha_T, self.nu_h, self.nu_z,
                    self.kappa_h, self.kappa_z, self.kappa_c, self.f_o, self.beta]
        pr = self.problem
        dp[11:16] = [getattr(pr, "tau_o", 0.0), self.rho_o, pr.Ly, getattr(pr, "lambda_r", 0.0),
                     getattr(pr, "theta_E", 0.0)]
        return ip, dp

    def init_state_auxiliary(self, grid):
        """``ocean_init_aux!`` (simple_box_problem.jl:12-22): y, everything else -0."""
        aux = np.full((grid.nelem, self.naux, grid.Np), -0.0)
        aux[:, 0, :] = grid.vgeo[:, 13, :]
        return aux

    def init_state_prognostic(self, grid, aux, t):
        x, y, z = (grid.vgeo[:, 12 + d, :] for d in range(3))
        u, v, eta, th = self.problem.init_state(self, x, y, z, t)
        Q = np.zeros((grid.nelem, self.ns, grid.Np))
        Q[:, 0], Q[:, 1], Q[:, 2], Q[:, 3] = u, v, eta, th
        return Q


def install_hydrostatic_boussinesq_hooks(dg, vert_filter=None, exp_filter=None):
    """``modeldata = (vert_filter, exp_filter)`` of the ocean driver + the two law methods
    (hydrostatic_boussinesq_model.jl:654-726) as hooks of the device operator:
    before the gradient pass, filter u (vertical cutoff) and theta (vertical exponential);
    after it, ``w = -div_h u``, column integrals for ``w`` and ``pkin``, surface ``w`` to ``wz0``.
    Defaults follow test/Ocean/HydrostaticBoussinesq/test_3D_spindown.jl:96-100."""
    g, law = dg.grid, dg.balance_law
    vert_filter = vert_filter or F.CutoffFilter(g, g.N[-1] - 1)
    exp_filter = exp_filter or F.ExponentialFilter(g, 1, 8)
    fu = F.make_device_filter(dg, vert_filter, F.FilterIndices(1, 2), direction=F.VerticalDirection)
    ft = F.make_device_filter(dg, exp_filter, F.FilterIndices(4), direction=F.VerticalDirection)
    dg.set_rhs_hooks(
        pre_filters=[fu, ft],
        gradflux_to_aux=[(0, 1, -1.0)],                     # A.w = -D.div_h u
        integral=dict(src=[(0, 1), (1, 3)], scale=[1.0, -law.alpha_T], dst=[1, 2]),
        reverse_integral=dict(rsrc=[2], rdst=[2]),
        surface_to_column=[(1, 3)],                          # w at the top node -> wz0
    )
    return fu, ft
