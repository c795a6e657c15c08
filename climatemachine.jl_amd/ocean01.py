"""Host-side mirror of the older split-explicit ocean of the reference,
``src/Ocean/SplitExplicit01`` -- the variant ``experiments/OceanSplitExplicit/simple_box.jl`` and
``test/Ocean/SplitExplicit/simple_box_2dt.jl`` run: ``OceanModel`` (``OceanModel.jl``),
``Continuity3dModel`` (``Continuity3dModel.jl``), ``BarotropicModel`` (``BarotropicModel.jl``),
``OceanDGModel`` (``OceanModel.jl:82-170``), the boundary-condition types of
``OceanBoundaryConditions.jl`` and ``SplitExplicitLSRK2nSolver``
(``SplitExplicitLSRK2nMethod.jl``, exchange functions ``Communication.jl``).

Layouts are the reference's: 3-D state ``u[2], eta, theta``; 3-D auxiliary
``w, pkin, wz0, u_d[2], dG_u[2], y``; 2-D state ``U[2], eta``; 2-D auxiliary
``G_U[2], U_c[2], eta_c, U_s[2], eta_s, Delta_u[2], eta_diag, Delta_eta, y``.  The 2-D model runs
on the one-layer extrusion of the 2-D grid (``ocean.extruded_barotropic_grid``).  Implicit
vertical diffusion (``numImplSteps > 0``, ``IVDCModel.jl``) is not carried.
"""
import ctypes as C

import numpy as np

from .mesh import filters as F

__all__ = ["SimpleBox01", "OceanModel01", "Continuity3dModel01", "BarotropicModel01",
           "OceanDGModel01", "SplitExplicitLSRK2nSolver01", "CoastlineFreeSlip", "CoastlineNoSlip",
           "OceanFloorFreeSlip", "OceanFloorNoSlip", "OceanSurfaceNoStressNoForcing",
           "OceanSurfaceStressNoForcing", "OceanSurfaceNoStressForcing", "OceanSurfaceStressForcing",
           "STATE_NAMES_3D", "AUX_NAMES_3D", "STATE_NAMES_2D", "AUX_NAMES_2D"]

PHYSICS_OCEAN_SE01, PHYSICS_CONTINUITY3D_SE01, PHYSICS_BAROTROPIC_SE01 = 7, 8, 9
CoastlineFreeSlip, CoastlineNoSlip, OceanFloorFreeSlip, OceanFloorNoSlip = 1, 2, 3, 4
OceanSurfaceNoStressNoForcing, OceanSurfaceStressNoForcing = 6, 7
OceanSurfaceNoStressForcing, OceanSurfaceStressForcing = 8, 9
STATE_NAMES_3D = ("u[1]", "u[2]", "η", "θ")
AUX_NAMES_3D = ("w", "pkin", "wz0", "u_d[1]", "u_d[2]", "ΔGu[1]", "ΔGu[2]", "y")
STATE_NAMES_2D = ("U[1]", "U[2]", "η")
AUX_NAMES_2D = ("Gᵁ[1]", "Gᵁ[2]", "U_c[1]", "U_c[2]", "η_c", "U_s[1]", "U_s[2]", "η_s", "Δu[1]",
                "Δu[2]", "η_diag", "Δη", "y")


class SimpleBox01:
    """``SimpleBox{T, BC}(Lx, Ly, H, tau_o, lambda_r, theta_E, boundary_conditions)`` of
    test/Ocean/SplitExplicit/simple_box_2dt.jl:45-69: wind stress ``tau_o``, surface relaxation
    towards ``theta_E (1 - y / Ly)`` at rate ``lambda_r``, a state at rest with a stratified
    temperature."""

    def __init__(self, Lx, Ly, H, tau_o=2e-1, lambda_r=20 / 86400, theta_E=10.0,
                 BC=(CoastlineNoSlip, OceanFloorNoSlip, OceanSurfaceStressForcing)):
        self.Lx, self.Ly, self.H = float(Lx), float(Ly), float(H)
        self.tau_o, self.lambda_r, self.theta_E = float(tau_o), float(lambda_r), float(theta_E)
        self.boundary_conditions = tuple(BC)

    def init_state(self, x, y, z):
        th = (5 + 4 * np.cos(y * np.pi / self.Ly)) * (1 + z / self.H)
        zero = -0.0 * np.ones_like(x)
        return zero, zero.copy(), zero.copy(), th


class OceanModel01:
    """``OceanModel{FT}(problem; grav, rho_o, c_h, c_z, add_fast_substeps, numImplSteps, ivdc_dt,
    alpha_T, nu_h, nu_z, kappa_h, kappa_z, kappa_c, f_o, beta)`` (OceanModel.jl:1-63; note
    ``kappa_c = 1e-4`` and ``grav = 10`` by default here)."""
    physics_id = PHYSICS_OCEAN_SE01
    ns, naux, ngrad, ngradflux, ngradlap, nhyper = 4, 8, 5, 9, 0, 0

    def __init__(self, problem, grav=10.0, rho_o=1000.0, c_h=0.0, c_z=0.0, add_fast_substeps=0,
                 numImplSteps=0, alpha_T=2e-4, nu_h=5e3, nu_z=5e-3, kappa_h=1e3, kappa_z=1e-4,
                 kappa_c=1e-4, f_o=1e-4, beta=1e-11):
        if numImplSteps:
            raise NotImplementedError("implicit vertical diffusion (IVDCModel) is not carried")
        self.problem = problem
        self.grav, self.rho_o, self.c_h, self.c_z = grav, rho_o, c_h, c_z
        self.add_fast_substeps, self.numImplSteps = int(add_fast_substeps), int(numImplSteps)
        self.alpha_T, self.nu_h, self.nu_z = alpha_T, nu_h, nu_z
        self.kappa_h, self.kappa_z, self.kappa_c, self.f_o, self.beta = kappa_h, kappa_z, kappa_c, f_o, beta

    def descriptor(self):
        ip = np.zeros(16, dtype=np.int32)
        ip[0] = int(self.numImplSteps > 0)
        bcs = self.problem.boundary_conditions
        ip[6] = len(bcs)
        for i, bc in enumerate(bcs):
            ip[7 + i] = bc
        pr = self.problem
        dp = np.zeros(32)
        dp[0:11] = [self.grav, self.c_h, self.c_z, self.alpha_T, self.nu_h, self.nu_z, self.kappa_h,
                    self.kappa_z, self.kappa_c, self.f_o, self.beta]
        dp[11:17] = [pr.tau_o, self.rho_o, pr.Ly, pr.lambda_r, pr.theta_E, pr.H]
        return ip, dp

    def init_state_auxiliary(self, grid):
        """``ocean_init_aux!(::OceanModel, ::SimpleBox, ...)`` (simple_box_2dt.jl:71-83)."""
        aux = np.full((grid.nelem, self.naux, grid.Np), -0.0)
        aux[:, 7, :] = grid.vgeo[:, 13, :]
        return aux

    def init_state_prognostic(self, grid, aux, t):
        x, y, z = (grid.vgeo[:, 12 + d, :] for d in range(3))
        Q = np.zeros((grid.nelem, self.ns, grid.Np))
        Q[:, 0], Q[:, 1], Q[:, 2], Q[:, 3] = self.problem.init_state(x, y, z)
        return Q


class Continuity3dModel01:
    """``Continuity3dModel(ocean)``: the prognostic variables of the ocean model, no auxiliary
    state; one evaluation leaves ``-grad_h . u`` in the tendency of the theta slot."""
    physics_id = PHYSICS_CONTINUITY3D_SE01
    ns, naux, ngrad, ngradflux, ngradlap, nhyper = 4, 0, 0, 0, 0, 0

    def __init__(self, ocean):
        self.ocean, self.problem = ocean, ocean.problem

    def descriptor(self):
        return self.ocean.descriptor()

    def init_state_auxiliary(self, grid):
        return np.zeros((grid.nelem, 1, grid.Np))      # a placeholder the law never reads

    def init_state_prognostic(self, grid, aux, t):
        return np.zeros((grid.nelem, self.ns, grid.Np))


class BarotropicModel01:
    """``BarotropicModel(baroclinic)`` (BarotropicModel.jl): ``U[2], eta`` on the 2-D grid."""
    physics_id = PHYSICS_BAROTROPIC_SE01
    ns, naux, ngrad, ngradflux, ngradlap, nhyper = 3, 13, 2, 6, 0, 0

    def __init__(self, baroclinic):
        self.baroclinic, self.problem = baroclinic, baroclinic.problem

    def descriptor(self):
        return self.baroclinic.descriptor()

    def init_state_auxiliary(self, grid):
        """``ocean_init_aux!(::BarotropicModel, ::SimpleBox, ...)`` (simple_box_2dt.jl:85-98)."""
        aux = np.full((grid.nelem, self.naux, grid.Np), -0.0)
        aux[:, 12, :] = grid.vgeo[:, 13, :]
        return aux

    def init_state_prognostic(self, grid, aux, t):
        return np.full((grid.nelem, self.ns, grid.Np), -0.0)


def default_filters(grid):
    """``vert_filter = CutoffFilter(grid, Nvert - 1)``, ``exp_filter = ExponentialFilter(grid, 1, 8)``
    (OceanModel.jl:90-93)."""
    return F.CutoffFilter(grid, grid.N[-1] - 1), F.ExponentialFilter(grid, 1, 8)


def hook_recipe(model):
    """The composition of ``update_auxiliary_state!(dg, ::OceanModel, Q, t, elems)``
    (OceanModel.jl:432-541) in terms of the operator's hook operations (0-based columns)."""
    return dict(
        pre_rhs=(3, 0),                                        # A.w = dQ.theta of conti3d_dg
        integral=dict(src=[(0, 0), (1, 3)], scale=[1.0, -(model.grav * model.alpha_T)], dst=[0, 1]),
        reverse_integral=dict(rsrc=[1], rdst=[1]),            # pkin: top value - value
        surface_to_column=[(0, 2)],                           # w at z = 0 -> wz0
        flow_deviation=(0, 3, model.problem.H),               # u_d = u - (1/H) int u
    )


class OceanDGModel01:
    """``OceanDGModel(bl::OceanModel, grid, ...)`` (OceanModel.jl:82-170) on the device: the
    ocean operator, the continuity operator its ``update_auxiliary_state!`` evaluates, the two
    filters, and the recorded composition.  ``self.dg`` is the operator; ``close()`` releases
    everything."""

    def __init__(self, model, grid, device="cuda:0"):
        from . import dgmodel
        self.model, self.grid = model, grid
        self.dg = dgmodel.DGModel(model, grid, device=device)
        self.conti3d_dg = dgmodel.DGModel(Continuity3dModel01(model), grid, device=device)
        vf, ef = default_filters(grid)
        self.fu = F.make_device_filter(self.dg, vf, F.FilterIndices(1, 2), direction=F.VerticalDirection)
        self.ft = F.make_device_filter(self.dg, ef, F.FilterIndices(4), direction=F.VerticalDirection)
        r = hook_recipe(model)
        self.dg.set_rhs_hooks(pre_filters=[self.fu, self.ft],
                              pre_rhs=(self.conti3d_dg,) + r["pre_rhs"],
                              ops_before_gradients=True, integral=r["integral"],
                              reverse_integral=r["reverse_integral"],
                              surface_to_column=r["surface_to_column"],
                              flow_deviation=r["flow_deviation"])

    def close(self):
        self.dg.set_rhs_hooks()
        for f in (self.fu, self.ft):
            f.close()
        self.conti3d_dg.close()
        self.dg.close()

    @staticmethod
    def connect_local(odgs):
        """The per-rank operators of one process (rank r = odgs[r]) through device copies: the
        ocean operators among themselves and the continuity operators they evaluate among
        themselves (``cmdg_group_rhs`` runs the latter in lock step inside the former's
        ``update_auxiliary_state!``)."""
        from . import dgmodel
        dgmodel.connect_local([o.dg for o in odgs])
        dgmodel.connect_local([o.conti3d_dg for o in odgs])


class SplitExplicitLSRK2nSolver01:
    """``SplitExplicitLSRK2nSolver(slow_solver, fast_solver)`` (SplitExplicitLSRK2nMethod.jl:40-78)
    over the device operators: ``dostep`` = ``cmdg_split_explicit01_step``."""

    def __init__(self, ocean_dg, dg_fast, Q_slow, Q_fast, dt_slow, dt_fast, t0=0.0,
                 fast_coefficients=None):
        """``fast_coefficients``: ``(RKA, RKB, RKC)`` of the fast solver when it is not the slow
        solver's scheme (the reference builds both from LSRK54CarpenterKennedy)."""
        from . import _lib
        from .odesolvers import LSRK54CarpenterKennedy
        self._lib = _lib
        self.ocean_dg, self.dg_slow, self.dg_fast = ocean_dg, ocean_dg.dg, dg_fast
        self.dt, self.dt_fast, self.t, self.steps = float(dt_slow), float(dt_fast), float(t0), 0
        ref = LSRK54CarpenterKennedy(dg_fast, Q_fast)
        self.dQ_fast = ref.dQ
        self.RKA, self.RKB, self.RKC = (np.asarray(c, dtype=np.float64) for c in (ref.RKA, ref.RKB, ref.RKC))
        self.dQ_slow = self.dg_slow.create_state(Q_slow.shape[1])
        self.dQ2fast = self.dg_slow.create_state(Q_slow.shape[1])
        model, g = ocean_dg.model, ocean_dg.grid
        d = _lib.CmdgOcean01Desc()
        d.nvertelem, d.H = int(g.topology.stacksize), float(model.problem.H)
        self._Imat = np.ascontiguousarray(np.asarray(g.Imat[-1], dtype=np.float64).T)
        d.Imat = self._Imat.ctypes.data
        d.add_fast_substeps = int(model.add_fast_substeps)
        if fast_coefficients is not None:
            self._fast = tuple(np.ascontiguousarray(c, dtype=np.float64) for c in fast_coefficients)
            d.nstages_fast = len(self._fast[0])
            d.rka_fast, d.rkb_fast, d.rkc_fast = (c.ctypes.data for c in self._fast)
        self.desc = d

    def dostep(self, Q_slow, Q_fast, nsteps=1):
        L = self.dg_slow.L
        p = lambda a: C.c_void_p(a.ctypes.data)
        self.dg_slow._torch_ready()
        for _ in range(int(nsteps)):
            self._lib.check(L.cmdg_split_explicit01_step(
                self.dg_slow.handle, self.dg_fast.handle, C.cast(C.byref(self.desc), C.c_void_p),
                Q_slow.data_ptr(), self.dQ_slow.data_ptr(), self.dQ2fast.data_ptr(),
                Q_fast.data_ptr(), self.dQ_fast.data_ptr(), self.t, self.dt, self.dt_fast,
                len(self.RKA), p(self.RKA), p(self.RKB), p(self.RKC)), self.dg_slow.handle)
            self.steps += 1
            self.t += self.dt
        self.dg_slow.synchronize()
        self.dg_fast.synchronize()
        self.ocean_dg.conti3d_dg.synchronize()

    @staticmethod
    def group_dostep(solvers, Q_slows, Q_fasts, nsteps=1):
        """The same step for the per-rank solvers of one process whose slow models
        (``OceanDGModel01.connect_local``) and fast models (``dgmodel.connect_local``) are
        connected locally: ``cmdg_group_split_explicit01_step``."""
        from .dgmodel import _harr, _parr
        s0 = solvers[0]
        L, n = s0.dg_slow.L, len(solvers)
        p = lambda a: C.c_void_p(a.ctypes.data)
        cast = lambda a: C.cast(a, C.c_void_p)
        slow, fast = _harr([s.dg_slow for s in solvers]), _harr([s.dg_fast for s in solvers])
        arr = [_parr(x) for x in (Q_slows, [s.dQ_slow for s in solvers], [s.dQ2fast for s in solvers],
                                  Q_fasts, [s.dQ_fast for s in solvers])]
        s0.dg_slow._torch_ready()
        for _ in range(int(nsteps)):
            s0._lib.check(L.cmdg_group_split_explicit01_step(
                cast(slow), cast(fast), n, C.cast(C.byref(s0.desc), C.c_void_p), *[cast(a) for a in arr],
                s0.t, s0.dt, s0.dt_fast, len(s0.RKA), p(s0.RKA), p(s0.RKB), p(s0.RKC)),
                s0.dg_slow.handle)
            for s in solvers:
                s.steps += 1
                s.t += s.dt
        for s in solvers:
            s.dg_slow.synchronize()
            s.dg_fast.synchronize()
            s.ocean_dg.conti3d_dg.synchronize()
