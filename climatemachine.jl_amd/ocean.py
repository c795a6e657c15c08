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

from .balancelaws import PHYSICS_HYDROSTATIC_BOUSSINESQ, PHYSICS_SHALLOW_WATER
from .mesh import filters as F

__all__ = ["HydrostaticBoussinesqModel", "ShallowWaterModel", "extruded_barotropic_grid",
           "SimpleBox", "OceanGyre", "HomogeneousBox", "OceanBC",
           "IMPENETRABLE_NOSLIP", "IMPENETRABLE_FREESLIP", "PENETRABLE_FREESLIP",
           "IMPENETRABLE_KINEMATIC_STRESS", "PENETRABLE_KINEMATIC_STRESS", "INSULATING",
           "TEMPERATURE_FLUX", "install_hydrostatic_boussinesq_hooks"]

# local Courant numbers of the ocean model (kind argument of cmdg_courant)
OCEAN_ADVECTIVE_COURANT, OCEAN_NONDIFFUSIVE_COURANT = 0, 1
OCEAN_DIFFUSIVE_COURANT, OCEAN_VISCOUS_COURANT = 2, 3
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
        kx, kz = 2 * np.pi / self.Lx, 2 * np.pi / self.H
        gH = m.grav * self.H
        lam = m.nu_h * kx ** 2 + m.nu_z * kz ** 2
        if self.rotation == FIXED:
            Mx = np.array([[-m.nu_h * kx ** 2, gH * kx], [-kx, 0.0]])
            A = expm(Mx * t) @ np.array([1.0, 1.0])
            Ub, Vb, eta = A[0] * np.sin(kx * x), 0.0 * x, A[1] * np.cos(kx * x)
            u0, v0 = np.exp(-lam * t) * np.cos(kz * z) * np.sin(kx * x), 0.0 * x
        else:
            # barotropic_state!(::Rotating) / baroclinic_deviation(::Rotating) (:224-276)
            assert self.rotation == ROTATING, "analytic solution restated for f = 0 and f = f_o"
            f = m.f_o
            Mx = np.array([[-m.nu_h * kx ** 2, f, gH * kx], [-f, -m.nu_h * kx ** 2, 0.0],
                           [-kx, 0.0, 0.0]])
            A = expm(Mx * t) @ np.array([1.0, 1.0, 1.0])
            Ub, Vb, eta = A[0] * np.sin(kx * x), A[1] * np.sin(kx * x), A[2] * np.cos(kx * x)
            B = expm(np.array([[-lam, f], [-f, -lam]]) * t) @ np.array([1.0, 1.0])
            u0 = B[0] * np.cos(kz * z) * np.sin(kx * x)
            v0 = B[1] * np.cos(kz * z) * np.sin(kx * x)
        return u0 + Ub / self.H, v0 + Vb / self.H, eta, 0.0 * x


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
                 kappa_c=1e-1, f_o=1e-4, beta=1e-11, grav=9.81, coupled=False):
        self.problem = problem
        # Coupled(): the baroclinic half of the split-explicit pair
        # (src/Ocean/SplitExplicit/HydrostaticBoussinesqCoupling.jl)
        self.coupled = bool(coupled)
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
        ip[3] = int(self.coupled)
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

    def calculate_dt(self, courant, Courant_number, t=0.0):
        """``calculate_dt(dg, model::HBModel, Q, Courant_number, t, ::EveryDirection)``
        (src/Ocean/HydrostaticBoussinesq/Courant.jl:113-159): the smallest of the advective
        (vertical), gravity-wave (horizontal), viscous and diffusive (vertical) limits.
        ``courant(kind, direction)`` evaluates ``courant(local_courant, dg, m, Q, 1, t,
        direction)`` -- ``dg.courant`` of the device operator or the oracle's."""
        V, H = 2, 1
        cfls = [courant(OCEAN_ADVECTIVE_COURANT, V), courant(OCEAN_NONDIFFUSIVE_COURANT, H),
                courant(OCEAN_VISCOUS_COURANT, V), courant(OCEAN_DIFFUSIVE_COURANT, V)]
        with np.errstate(divide="ignore"):
            return min(Courant_number / np.float64(c) for c in cfls)

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


class ShallowWaterModel:
    """``ShallowWaterModel{FT}(param_set, problem, turbulence, advection; coupling, c, f_o, beta)``
    (src/Ocean/ShallowWater/ShallowWaterModel.jl:60-86): state ``eta, U[2]``; auxiliary
    ``y, G_U[2], Delta_u[2]``; gradient ``U[2]``; gradient flux ``nu grad U`` (3 x 2).
    The barotropic half of the split-explicit pair.  It is a 2-D law in the reference; here
    it runs on the 3-D kernels over :func:`extruded_barotropic_grid` (fields constant along the
    extrusion; the third flux component is identically zero)."""
    physics_id = PHYSICS_SHALLOW_WATER
    ns, naux, ngrad, ngradflux, ngradlap, nhyper = 3, 5, 2, 6, 0, 0

    def __init__(self, problem, nu, advection=False, coupled=False, c=0.0, f_o=1e-4, beta=1e-11,
                 grav=9.81):
        self.problem, self.nu, self.advection, self.coupled = problem, nu, bool(advection), bool(coupled)
        self.c, self.f_o, self.beta, self.grav = c, f_o, beta, grav

    def state_names(self):
        return ["η", "U[1]", "U[2]"]

    def descriptor(self):
        ip = np.zeros(16, dtype=np.int32)
        ip[0], ip[1], ip[2], ip[3] = int(self.advection), 0, self.problem.rotation, int(self.coupled)
        dp = np.zeros(32)
        dp[0:6] = [self.grav, self.problem.H, self.c, self.nu, self.f_o, self.beta]
        return ip, dp

    def init_state_auxiliary(self, grid):
        """``ocean_init_aux!(::SWModel, ...)`` (simple_box_problem.jl:33-38)."""
        aux = np.full((grid.nelem, self.naux, grid.Np), -0.0)
        aux[:, 0, :] = grid.vgeo[:, 13, :]
        return aux

    def init_state_prognostic(self, grid, aux, t):
        """``ocean_init_state!(::SWModel, ::SimpleBox, ...)`` (simple_box_problem.jl:128-146):
        the barotropic mode ``U = A1 sin(kx x)``, ``eta = A2 cos(kx x)``, ``A = exp(M t) [1, 1]``."""
        p = self.problem
        x = grid.vgeo[:, 12, :]
        kx = 2 * np.pi / p.Lx
        gH = self.grav * p.H
        Q = np.zeros((grid.nelem, 3, grid.Np))
        if p.rotation == FIXED:
            A = expm(np.array([[-self.nu * kx ** 2, gH * kx], [-kx, 0.0]]) * t) @ np.array([1.0, 1.0])
            Q[:, 0] = A[1] * np.cos(kx * x)
            Q[:, 1] = A[0] * np.sin(kx * x)
            Q[:, 2] = -0.0
        else:      # barotropic_state!(::Rotating) (simple_box_problem.jl:224-243)
            assert p.rotation == ROTATING
            f = self.f_o
            A = expm(np.array([[-self.nu * kx ** 2, f, gH * kx], [-f, -self.nu * kx ** 2, 0.0],
                               [-kx, 0.0, 0.0]]) * t) @ np.array([1.0, 1.0, 1.0])
            Q[:, 0] = A[2] * np.cos(kx * x)
            Q[:, 1] = A[0] * np.sin(kx * x)
            Q[:, 2] = A[1] * np.sin(kx * x)
        return Q


def extruded_barotropic_grid(xrange, yrange, N, periodicity=(True, True), boundary=((0, 0), (0, 0)),
                             connectivity="full", rank=0, size=1, N_extrusion=None):
    """One periodic layer of unit height over the horizontal brick mesh ``xrange x yrange``:
    the grid the 2-D barotropic model runs on with the 3-D kernels.  Built by the same stacked
    topology as the 3-D ocean grid, so element ``eh`` of it sits under the stack
    ``eh * nvert .. (eh + 1) * nvert - 1`` of the 3-D grid.  ``N_extrusion`` is the polynomial
    order along the extrusion (default ``N``); nothing varies in that direction, so
    ``N_extrusion = 1`` (two nodes) carries the same 2-D arithmetic at 2/5 of the work."""
    from . import mesh as M
    rng = [np.asarray(xrange), np.asarray(yrange), np.array([0.0, 1.0])]
    topl = M.StackedBrickTopology(rng, periodicity=(periodicity[0], periodicity[1], True),
                                  boundary=(boundary[0], boundary[1], (0, 0)),
                                  connectivity=connectivity, rank=rank, size=size)
    Nx = N if N_extrusion is None else int(N_extrusion)
    return M.DiscontinuousSpectralElementGrid(topl, (N, N, Nx))


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
    # Coupled(): u_d = u - (vertical mean of u) after the filters
    # (HydrostaticBoussinesqCoupling.jl:43-85); state u = columns 0, 1, aux u_d = columns 4, 5
    flow = (0, 4, law.problem.H) if getattr(law, "coupled", False) else None
    dg.set_rhs_hooks(
        pre_filters=[fu, ft],
        gradflux_to_aux=[(0, 1, -1.0)],                     # A.w = -D.div_h u
        integral=dict(src=[(0, 1), (1, 3)], scale=[1.0, -law.alpha_T], dst=[1, 2]),
        reverse_integral=dict(rsrc=[2], rdst=[2]),
        surface_to_column=[(1, 3)],                          # w at the top node -> wz0
        flow_deviation=flow,
    )
    return fu, ft


class SplitExplicitSolver:
    """``SplitExplicitSolver(slow_solver, fast_solver)`` of
    src/Numerics/ODESolvers/SplitExplicitMethod.jl:30-68 over two device operators: ``dg_slow``
    the 3-D HydrostaticBoussinesqModel (with its hooks installed), ``dg_fast`` the
    ShallowWaterModel on ``extruded_barotropic_grid``.  ``dostep`` advances both states in
    place by ``nsteps`` slow steps; the coupling runs when the slow law is ``Coupled``."""

    def __init__(self, dg_slow, dg_fast, Q_slow, Q_fast, dt_slow, dt_fast, t0=0.0,
                 coefficients=None, fast_priority=None):
        from . import _lib
        from .odesolvers import LSRK54CarpenterKennedy
        import ctypes as C
        import os
        self.dg_slow, self.dg_fast = dg_slow, dg_fast
        # The slow model's stream is the critical path of a slow step (it is never idle,
        # profiles/r04_ocean_timeline.txt) and its kernels run 30-55 % longer while a burst of
        # barotropic launches shares the CUs with them; the bursts themselves are hidden behind
        # the stage's second slow evaluation.  The barotropic streams therefore yield: lowest
        # stream priority (measured: 11.63 ms per slow step against 11.84 at the default and 11.81 at
        # the highest; the two models on disjoint compute units: 13.7 - 22.9 ms,
        # profiles/r04_ab_ocean_cu_partition.txt).  CMDG_OCEAN_FAST_PRIORITY overrides.
        if fast_priority is None:
            fast_priority = int(os.environ.get("CMDG_OCEAN_FAST_PRIORITY", "-1"))
        dg_fast.set_option(_lib.OPT_STREAM_PRIORITY, int(fast_priority))

        self.dt, self.dt_fast, self.t, self.steps = float(dt_slow), float(dt_fast), float(t0), 0
        if coefficients is None:
            ref = LSRK54CarpenterKennedy(dg_fast, Q_fast)
            coefficients = (ref.RKA, ref.RKB, ref.RKC)
            self.dQ_fast = ref.dQ
        else:
            self.dQ_fast = dg_fast.create_state(Q_fast.shape[1])
        self.RKA, self.RKB, self.RKC = (np.asarray(c, dtype=np.float64) for c in coefficients)
        self.dQ_slow = dg_slow.create_state(Q_slow.shape[1])
        self.dQ2fast = dg_slow.create_state(Q_slow.shape[1])
        law, g = dg_slow.balance_law, dg_slow.grid
        self.coupled = bool(getattr(law, "coupled", False))
        d = _lib.CmdgOceanCouplingDesc()
        d.nvertelem, d.H = int(g.topology.stacksize), float(law.problem.H)
        self._Imat = np.ascontiguousarray(np.asarray(g.Imat[-1], dtype=np.float64).T)
        d.Imat = self._Imat.ctypes.data
        d.slow_u_col, d.slow_eta_col, d.slow_dGu_col = 0, 2, 6
        d.fast_eta_col, d.fast_U_col, d.fast_GU_col, d.fast_du_col = 0, 1, 1, 3
        self.desc, self._C, self._lib = d, C, _lib

    def dostep(self, Q_slow, Q_fast, nsteps=1):
        C, L = self._C, self.dg_slow.L
        p = lambda a: C.c_void_p(a.ctypes.data)
        self.dg_slow._torch_ready()
        for _ in range(int(nsteps)):
            self._lib.check(L.cmdg_split_explicit_step(
                self.dg_slow.handle, self.dg_fast.handle, C.byref(self.desc), int(self.coupled),
                Q_slow.data_ptr(), self.dQ_slow.data_ptr(), self.dQ2fast.data_ptr(),
                Q_fast.data_ptr(), self.dQ_fast.data_ptr(), self.t, self.dt, self.dt_fast,
                len(self.RKA), p(self.RKA), p(self.RKB), p(self.RKC)), self.dg_slow.handle)
            self.steps += 1
            self.t += self.dt      # (the reference accumulates time the same way)
        self.dg_slow.synchronize()
        self.dg_fast.synchronize()

    @staticmethod
    def group_dostep(solvers, Q_slows, Q_fasts, nsteps=1):
        """The same step for the per-rank solvers of one process whose slow models and fast
        models are connected by ``dgmodel.connect_local`` (single-GPU rehearsal of the
        partitioned run): ``cmdg_group_split_explicit_step``."""
        from .dgmodel import _harr, _parr
        s0 = solvers[0]
        C, L, n = s0._C, s0.dg_slow.L, len(solvers)
        p = lambda a: C.c_void_p(a.ctypes.data)
        cast = lambda a: C.cast(a, C.c_void_p)
        slow, fast = _harr([s.dg_slow for s in solvers]), _harr([s.dg_fast for s in solvers])
        arr = [_parr(x) for x in (Q_slows, [s.dQ_slow for s in solvers],
                                  [s.dQ2fast for s in solvers], Q_fasts,
                                  [s.dQ_fast for s in solvers])]
        s0.dg_slow._torch_ready()
        for _ in range(int(nsteps)):
            s0._lib.check(L.cmdg_group_split_explicit_step(
                cast(slow), cast(fast), n, C.byref(s0.desc), int(s0.coupled), *[cast(a) for a in arr],
                s0.t, s0.dt, s0.dt_fast, len(s0.RKA), p(s0.RKA), p(s0.RKB), p(s0.RKC)),
                s0.dg_slow.handle)
            for s in solvers:
                s.steps += 1
                s.t += s.dt
        for s in solvers:
            s.dg_slow.synchronize()
            s.dg_fast.synchronize()

    # the exchange functions on their own (src/Ocean/SplitExplicit/Communication.jl)
    def initialize_states(self):
        self._lib.check(self.dg_slow.L.cmdg_ocean_initialize_states(
            self.dg_slow.handle, self.dg_fast.handle, self._C.byref(self.desc)), self.dg_slow.handle)

    def tendency_from_slow_to_fast(self, dQ_slow):
        self._lib.check(self.dg_slow.L.cmdg_ocean_tendency_from_slow_to_fast(
            self.dg_slow.handle, self.dg_fast.handle, self._C.byref(self.desc),
            dQ_slow.data_ptr()), self.dg_slow.handle)

    def reconcile_from_fast_to_slow(self, Q_slow, Q_fast):
        self._lib.check(self.dg_slow.L.cmdg_ocean_reconcile_from_fast_to_slow(
            self.dg_slow.handle, self.dg_fast.handle, self._C.byref(self.desc),
            Q_slow.data_ptr(), Q_fast.data_ptr()), self.dg_slow.handle)
