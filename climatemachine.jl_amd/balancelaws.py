"""Host-side descriptions of the balance laws the HIP library carries as C++
functors.

In the reference a ``BalanceLaw`` is Julia code inlined into every kernel
(``src/BalanceLaws/interface.jl:37-464``).  A HIP library cannot call Julia per
node, so each supported law exists twice: here (state counts, one-time host
initialisation of auxiliary/prognostic state, and the ``(physics_id, iparam,
dparam)`` parameter block handed across the C ABI) and as a device functor in
``csrc/physics_*.h``.

``AdvectionDiffusion`` mirrors the reference's test law
``test/Numerics/DGMethods/advection_diffusion/advection_diffusion_model.jl:92-617``.
"""
import numpy as np

from .mesh.grids import _x1, _x2, _x3

EveryDirection, HorizontalDirection, VerticalDirection = 0, 1, 2
RusanovNumericalFlux, CentralNumericalFluxFirstOrder = 0, 1
# methods of the dry AtmosModel (src/Atmos/Model/AtmosModel.jl:1006, :1154, :1515)
RoeNumericalFlux, HLLCNumericalFlux, LMARSNumericalFlux = 2, 3, 4
# RoeNumericalFluxMoist(; LM, HH, LV, LVPP) of the moist AtmosModel (AtmosModel.jl:1276-1513)
(RoeNumericalFluxMoist, RoeNumericalFluxMoistLM, RoeNumericalFluxMoistHH, RoeNumericalFluxMoistLV,
 RoeNumericalFluxMoistLVPP) = 5, 6, 7, 8, 9

PHYSICS_ADVECTION_DIFFUSION = 1
PHYSICS_DRY_ATMOS = 2
PHYSICS_HYDROSTATIC_BOUSSINESQ = 3
PHYSICS_PRESSURE_GRADIENT = 4
PHYSICS_SHALLOW_WATER = 5
PHYSICS_MOIST_ATMOS = 6

__all__ = [
    "EveryDirection", "HorizontalDirection", "VerticalDirection",
    "RusanovNumericalFlux", "CentralNumericalFluxFirstOrder", "RoeNumericalFlux",
    "HLLCNumericalFlux", "LMARSNumericalFlux", "RoeNumericalFluxMoist", "RoeNumericalFluxMoistLM",
    "RoeNumericalFluxMoistHH", "RoeNumericalFluxMoistLV", "RoeNumericalFluxMoistLVPP",
    "InhomogeneousBC", "HomogeneousBC", "AdvectionDiffusion", "Pseudo1D",
    "ConstantHyperDiffusion", "DirectionSplitBox",
]


class InhomogeneousBC:
    """``InhomogeneousBC{O}`` (advection_diffusion_model.jl:62-64)."""

    def __init__(self, order):
        self.order = order

    @property
    def bit(self):
        return 1 << self.order


class HomogeneousBC:
    """``HomogeneousBC{O}`` (advection_diffusion_model.jl:62-63)."""

    def __init__(self, order):
        self.order = order

    @property
    def bit(self):
        return 1 << (self.order + 4)


class PressureGradientModel:
    """``PressureGradientModel`` (src/Atmos/Model/ref_state.jl:196-233): tendency = DG gradient
    of the auxiliary field ``p``."""
    physics_id = PHYSICS_PRESSURE_GRADIENT
    ns, naux, ngrad, ngradflux, ngradlap, nhyper = 3, 1, 0, 0, 0, 0

    def __init__(self, p):
        self._p = np.ascontiguousarray(p[:, None, :], dtype=np.float64)

    def descriptor(self):
        return np.zeros(16, dtype=np.int32), np.zeros(32)

    def init_state_auxiliary(self, grid):
        return self._p

    def init_state_prognostic(self, grid, aux, t):
        return np.zeros((grid.nelem, 3, grid.Np))


class NoFlowBC:
    """``NoFlowBC`` of test/Numerics/DGMethods/advection_diffusion/advection_sphere.jl:118-132:
    the plus-side advection velocity is the negative of the minus side."""
    bit = 1 << 8


class SolidBodyRotation:
    """``SolidBodyRotation`` (advection_sphere.jl:28-55): one revolution per unit time about
    the z axis on a spherical shell; the state is a Gaussian in longitude / latitude."""
    problem_id = 5           # host-only problem: the kernels need no data from it

    def dparam(self):
        return np.zeros(32)

    @staticmethod
    def _lonlat(coord):
        r = np.sqrt(coord[0] ** 2 + coord[1] ** 2 + coord[2] ** 2)
        return np.arctan2(coord[1], coord[0]), np.arcsin(coord[2] / r), r

    def init_velocity_diffusion(self, law, aux, coord):
        lam, phi, r = self._lonlat(coord)
        ul = 2 * np.pi * np.cos(phi) * r
        up = 0.0
        aux[:, law.off_u + 0, :] = -ul * np.sin(lam) - up * np.cos(lam) * np.sin(phi)
        aux[:, law.off_u + 1, :] = +ul * np.cos(lam) - up * np.sin(lam) * np.sin(phi)
        aux[:, law.off_u + 2, :] = +up * np.cos(phi)

    def initial_condition(self, coord, t):
        lam, phi, _ = self._lonlat(coord)
        return np.exp(-((3 * lam) ** 2 + (3 * phi) ** 2))

    finaltime = 1.0
    u_scale = 2 * np.pi


class ReversingDeformationalFlow:
    """``ReversingDeformationalFlow`` (advection_sphere.jl:56-103, after Lauritzen et al. 2012):
    two Gaussian hills in a time-dependent deformational flow that reverses at t = 2.5 and
    returns them at t = 5.  The velocity lives in the auxiliary state and is refreshed by the
    nodal ``update_auxiliary_state!`` at every right-hand-side evaluation (on the device:
    ``AdvDiff::update_aux``, problem 7)."""
    problem_id = 7
    finaltime = 5.0
    u_scale = 2.9

    def dparam(self):
        return np.zeros(32)

    def init_velocity_diffusion(self, law, aux, coord):
        pass                                   # init_velocity_diffusion!(...) = nothing

    def initial_condition(self, coord, t):
        x, y, z = coord
        r = np.sqrt(x * x + y * y + z * z)
        rho = 0.0 * x
        for lam, phi in ((5 * np.pi / 6, 0.0), (7 * np.pi / 6, 0.0)):
            xi, yi, zi = r * np.cos(phi) * np.cos(lam), r * np.cos(phi) * np.sin(lam), r * np.sin(phi)
            rho = rho + 0.95 * np.exp(-5 * ((x - xi) ** 2 + (y - yi) ** 2 + (z - zi) ** 2))
        return rho


class DiffusionSphere:
    """``DiffusionSphere`` of test/Numerics/DGMethods/advection_diffusion/
    diffusion_hyperdiffusion_sphere.jl:25-55.  The reference runs it as a two-equation law
    (equation 1: ``D = mu I``, ``H = 0``; equation 2: ``D = 0``, ``H = mu I``); the equations do
    not couple, so each is run here as its own single-equation law (``hyper`` selects which)
    and carries the l = 2, m = 1 spherical harmonic decaying at its analytic rate."""
    problem_id = 6           # host-only problem

    def __init__(self, hyper, mu=1 / 10000):
        self.hyper, self.mu = bool(hyper), mu

    def dparam(self):
        return np.zeros(32)

    def init_velocity_diffusion(self, law, aux, coord):
        I = (self.mu * np.eye(3)).flatten(order="F")
        if law.diffusion:
            aux[:, law.off_D:law.off_D + 9, :] = I[None, :, None]
        if law.hyperdiffusion:
            aux[:, law.off_H:law.off_H + 9, :] = I[None, :, None]

    def initial_condition(self, coord, t):
        x, y, z = coord
        r = np.sqrt(x * x + y * y + z * z)
        th = np.arctan2(np.sqrt(x ** 2 + y ** 2), z)
        ph = np.arctan2(y, x)
        rho0 = np.cos(ph) * np.sin(th) * np.cos(th)
        c = 2 * (2 + 1) / r ** 2
        return rho0 * np.exp(-(c ** 2 if self.hyper else c) * self.mu * t)


class Pseudo1D:
    """``Pseudo1D{n, alpha, beta, mu, delta}`` (pseudo1D_advection_diffusion.jl:28-68)."""
    problem_id = 0

    def __init__(self, n, alpha, beta, mu, delta):
        self.n = np.asarray(n, dtype=np.float64)
        self.alpha, self.beta, self.mu, self.delta = alpha, beta, mu, delta

    def dparam(self):
        d = np.zeros(32)
        d[0:3] = self.n
        d[3:7] = [self.alpha, self.beta, self.mu, self.delta]
        return d

    def init_velocity_diffusion(self, law, aux, coord):
        n = self.n
        if law.advection:
            aux[:, law.off_u:law.off_u + 3, :] = (self.alpha * n)[None, :, None]
        if law.diffusion:
            D = (self.beta * n)[:, None] * n[None, :]          # D[i, j] = beta n_i n_j
            aux[:, law.off_D:law.off_D + 9, :] = D.flatten(order="F")[None, :, None]

    def initial_condition(self, coord, t):
        n = self.n
        xn = n[0] * coord[0] + n[1] * coord[1] + n[2] * coord[2]
        a = xn - self.mu - self.alpha * t
        return np.exp(-(a * a) / (4 * self.beta * (self.delta + t))) / np.sqrt(1 + t / self.delta)


class ConstantHyperDiffusion:
    """``ConstantHyperDiffusion{dim, dir}`` (periodic_3D_hyperdiffusion.jl:29-63)."""
    problem_id = 1

    def __init__(self, dim, direction, D):
        self.dim, self.direction = dim, direction
        self.D = np.asarray(D, dtype=np.float64).reshape(3, 3)

    def dparam(self):
        d = np.zeros(32)
        d[0:9] = self.D.flatten(order="F")
        d[9], d[10] = self.dim, self.direction
        return d

    def init_velocity_diffusion(self, law, aux, coord):
        aux[:, law.off_H:law.off_H + 9, :] = self.D.flatten(order="F")[None, :, None]

    def _c(self):
        k = np.array([1.0, 2.0, 3.0])
        dim = self.dim
        kD = (k[:, None] * k[None, :]) * self.D
        if self.direction in (EveryDirection, HorizontalDirection):
            dd = dim if self.direction == EveryDirection else dim - 1
            s2 = sum(k[i] * k[i] for i in range(dd))
            skd = 0.0
            for j in range(dd):
                for i in range(dd):
                    skd += kD[i, j]
            return s2 * skd
        return k[dim - 1] ** 2 * kD[dim - 1, dim - 1]

    def initial_condition(self, coord, t):
        k = [1.0, 2.0, 3.0]
        kx = sum(k[i] * coord[i] for i in range(self.dim))
        return np.sin(kx) * np.exp(-self._c() * t)


class HyperDiffusionBC:
    """``ConstantHyperDiffusion{FT}(mu, k)`` of test/Numerics/DGMethods/advection_diffusion/
    hyperdiffusion_bc.jl:25-112: ``H = mu I``; the solution ``cos(k1 x) cos(k2 y) cos(k3 z)
    exp(-|k|^4 mu t)`` supplies boundary data of orders 0-3 (value, gradient, Laplacian,
    gradient of the Laplacian)."""
    problem_id = 2

    def __init__(self, mu, k):
        self.mu, self.k = float(mu), np.asarray(k, dtype=np.float64)

    def dparam(self):
        d = np.zeros(32)
        d[0] = self.mu
        d[1:4] = self.k
        return d

    def init_velocity_diffusion(self, law, aux, coord):
        aux[:, law.off_H:law.off_H + 9, :] = (self.mu * np.eye(3)).flatten(order="F")[None, :, None]

    def initial_condition(self, coord, t):
        k = self.k
        k2 = float(np.sum(k ** 2))
        return (np.cos(k[0] * coord[0]) * np.cos(k[1] * coord[1]) * np.cos(k[2] * coord[2])
                * np.exp(-k2 ** 2 * self.mu * t))


class HeatEqn:
    """``HeatEqn{n, kappa, A}`` (pseudo1D_heat_eqn.jl:28-88): ``D = n n'``, solution
    ``xi + A cos(kappa xi) exp(-kappa^2 t)`` with ``xi = n . x``; used with ``flux_bc=True``
    (the test's own ``normal_boundary_flux_second_order!`` is the flux form of the Dirichlet /
    Neumann conditions)."""
    problem_id = 3

    def __init__(self, n, kappa=10 * np.pi / 2, A=1.0):
        self.n, self.kappa, self.A = np.asarray(n, dtype=np.float64), float(kappa), float(A)

    def dparam(self):
        d = np.zeros(32)
        d[0:3] = self.n
        d[3], d[4] = self.kappa, self.A
        return d

    def init_velocity_diffusion(self, law, aux, coord):
        D = self.n[:, None] * self.n[None, :]
        aux[:, law.off_D:law.off_D + 9, :] = D.flatten(order="F")[None, :, None]

    def initial_condition(self, coord, t):
        n = self.n
        xn = n[0] * coord[0] + n[1] * coord[1] + n[2] * coord[2]
        return xn + self.A * np.cos(self.kappa * xn) * np.exp(-self.kappa ** 2 * t)


class DirectionSplitBox:
    """``TestProblem{adv, diff, dir, Box}`` of the reference's tendency-splitting test
    (direction_splitting_advection_diffusion.jl:28-68): u = P sin(pi x), D = P / 200,
    rho0 = prod(sin(pi x)), with P the projection selected by ``dir`` (k = e_3)."""
    problem_id = 4

    def __init__(self, direction, advection=True, diffusion=True):
        self.direction, self.adv, self.diff = direction, advection, diffusion
        k = np.array([0.0, 0.0, 1.0])
        kk = np.outer(k, k)
        self.P = {EveryDirection: np.eye(3), VerticalDirection: kk,
                  HorizontalDirection: np.eye(3) - kk}[direction]

    def dparam(self):
        return np.zeros(32)

    def init_velocity_diffusion(self, law, aux, coord):
        x = np.stack(coord, axis=1)                      # (nelem, 3, Np)
        if law.advection:
            u = np.einsum("ij,ejn->ein", self.P, np.sin(np.pi * x)) if self.adv else 0 * x
            aux[:, law.off_u:law.off_u + 3, :] = u
        if law.diffusion:
            D = self.P / 200 if self.diff else np.zeros((3, 3))
            aux[:, law.off_D:law.off_D + 9, :] = D.flatten(order="F")[None, :, None]

    def initial_condition(self, coord, t):
        return np.sin(np.pi * coord[0]) * np.sin(np.pi * coord[1]) * np.sin(np.pi * coord[2])


class AdvectionDiffusion:
    """``AdvectionDiffusion{dim}(problem, bcs; num_equations=1, flux_bc, advection,
    diffusion, hyperdiffusion)`` (advection_diffusion_model.jl:92-123).

    State layout (vars_state, :127-183): prognostic ``rho``; auxiliary
    ``coord(3) [, u(3)] [, D(9)] [, H(9)]``; gradient ``rho``; gradient-flux
    ``sigma(3)``; gradient-laplacian ``rho``; hyperdiffusive ``eta(3)``."""
    physics_id = PHYSICS_ADVECTION_DIFFUSION

    def __init__(self, dim, problem, boundary_conditions=(), flux_bc=False,
                 advection=True, diffusion=True, hyperdiffusion=False):
        self.dim = dim
        self.problem = problem
        self.boundary_conditions = tuple(boundary_conditions)
        self.flux_bc = bool(flux_bc)
        self.advection, self.diffusion, self.hyperdiffusion = (
            bool(advection), bool(diffusion), bool(hyperdiffusion))
        o = 3
        self.off_u = o
        o += 3 if self.advection else 0
        self.off_D = o
        o += 9 if self.diffusion else 0
        self.off_H = o
        o += 9 if self.hyperdiffusion else 0
        self.ns = 1
        self.naux = o
        self.ngrad = 1 if (self.diffusion or self.hyperdiffusion) else 0
        self.ngradflux = 3 if self.diffusion else 0
        self.ngradlap = 1 if self.hyperdiffusion else 0
        self.nhyper = 3 if self.hyperdiffusion else 0

    def descriptor(self):
        ip = np.zeros(16, dtype=np.int32)
        ip[0] = 1
        ip[1], ip[2], ip[3] = self.advection, self.diffusion, self.hyperdiffusion
        ip[4] = self.flux_bc
        ip[5] = self.problem.problem_id
        ip[6] = len(self.boundary_conditions)
        for i, bc in enumerate(self.boundary_conditions):
            bcs = bc if isinstance(bc, (tuple, list)) else (bc,)
            m = 0
            for b in bcs:
                m |= b.bit
            ip[7 + i] = m
        return ip, self.problem.dparam()

    # -- one-time host initialisation (nodal_init_state_auxiliary!, :357-365) --
    def init_state_auxiliary(self, grid):
        aux = np.zeros((grid.nelem, self.naux, grid.Np))
        coord = [grid.vgeo[:, c, :] for c in (_x1, _x2, _x3)]
        for d in range(3):
            aux[:, d, :] = coord[d]
        self.problem.init_velocity_diffusion(self, aux, coord)
        return aux

    # -- init_state_prognostic! (:384-392) --
    def init_state_prognostic(self, grid, aux, t):
        Q = np.zeros((grid.nelem, self.ns, grid.Np))
        coord = [aux[:, d, :] for d in range(3)]
        Q[:, 0, :] = self.problem.initial_condition(coord, t)
        return Q
