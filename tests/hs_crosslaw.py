"""Shared body of the cross-law hyperdiffusion check (CPU oracle and device).

The dry atmosphere's DryBiharmonic terms have no stored number in the reference.  The
advection-diffusion test law's hyperdiffusion does (``periodic_3D_hyperdiffusion.jl``,
``hyperdiffusion_bc.jl``, ``diffusion_hyperdiffusion_sphere.jl``: reproduced by oracle and device),
and it evaluates ``d rho / dt = -div(H grad Lap rho)`` with an arbitrary nodal tensor ``H``
through the same three passes.  With ``H = rho nu_4 I`` and ``rho := h_tot`` or a Cartesian
component of ``u_h`` that is, term by term, what ``flux_second_order!`` of the atmosphere adds
(``src/Atmos/Model/tendencies_momentum.jl``, ``tendencies_energy.jl`` hyperdiffusion fluxes:
``rho nu grad^3 u_h`` and ``rho nu grad^3 h_tot + (nu grad^3 u_h)' rho u``), so

    T_atmos(with DryBiharmonic) - T_atmos(without)  ==  sum of scalar-law tendencies

to rounding.  This pins the gradient arguments (u_h, h_tot), the nu_4 scaling, the signs and the
rho / rho u factors of the atmosphere functor against a law whose numbers the reference stores."""
import numpy as np


class _NodalHyper:
    """A host-only 'problem' of the advection-diffusion law: nodal H, given scalar field."""
    problem_id = 6

    def __init__(self, H, s):
        self.H, self.s = H, s

    def dparam(self):
        return np.zeros(32)

    def init_velocity_diffusion(self, law, aux, coord):
        for d in range(3):
            aux[:, law.off_H + 4 * d, :] = self.H          # H = h(x) I, column-major 3 x 3

    def initial_condition(self, coord, t):
        return self.s


def crosslaw_residual(cm, make_dg, n_horz=3, n_vert=2):
    """Returns (max |difference - scalar-law sum| / max |difference|) per prognostic variable.
    ``make_dg(law, grid, direction, diffusion_direction, nf)`` builds an operator whose call is
    ``dg(tendency, Q, t, alpha, beta)`` on numpy arrays."""
    M, A, BL = cm.mesh, cm.atmos, cm.balancelaws
    ps = A.PlanetParameters()
    R = np.linspace(ps.planet_radius, ps.planet_radius + 30e3, n_vert + 1)
    topl = M.StackedCubedSphereTopology(n_horz, R, boundary=(1, 2))
    grid = M.DiscontinuousSpectralElementGrid(topl, 4, meshwarp=M.equiangular_cubed_sphere_warp)

    def atmos(tau):
        return A.DryAtmosModel(A.HeldSuarezSetup(ps), orientation=A.ORIENT_SPHERICAL,
                               ref_state=A.DecayingTemperatureProfile(ps, 290.0, 220.0, 8e3),
                               viscosity=0.0, dynamic_viscosity=False, hyperdiffusion_timescale=tau,
                               sources=0, boundary_conditions=(A.BC_ATMOS_DEFAULT, A.BC_ATMOS_DEFAULT),
                               param_set=ps)
    tau = 8 * 3600.0
    lawh, law0 = atmos(tau), atmos(None)
    auxh = lawh.init_state_auxiliary(grid)
    x = auxh[:, 0:3, :]
    r = np.sqrt((x ** 2).sum(axis=1))
    Q = lawh.init_state_prognostic(grid, auxh, 0.0)
    rho = Q[:, 0, :].copy()
    # a smooth wind with all three Cartesian components and a temperature anomaly
    lam, phi = np.arctan2(x[:, 1], x[:, 0]), np.arcsin(x[:, 2] / r)
    u = np.stack([30 * np.sin(2 * lam) * np.cos(phi), 20 * np.cos(3 * phi) * np.cos(lam),
                  10 * np.sin(lam + phi)], axis=1)
    Q[:, 1:4, :] = rho[:, None, :] * u
    Q[:, 4, :] += rho * 0.5 * (u ** 2).sum(axis=1) + rho * ps.cv_d * 2.0 * np.sin(3 * lam) * np.cos(phi) ** 2
    Th, T0 = np.zeros_like(Q), np.zeros_like(Q)
    make_dg(lawh, grid, 0, 1, 0)(Th, Q.copy(), 0.0, 1.0, 0.0)
    make_dg(law0, grid, 0, 1, 0)(T0, Q.copy(), 0.0, 1.0, 0.0)
    D = (Th - T0)[:grid.nreal]

    # the atmosphere's gradient arguments, from its definition
    o = lawh.off_phi
    khat = auxh[:, o + 1:o + 4, :] / ps.grav
    u_h = u - khat * (khat * u).sum(axis=1)[:, None, :]
    e_int = (Q[:, 4] - 0.5 * rho * (u ** 2).sum(axis=1) - rho * auxh[:, o]) / rho
    T = ps.T_0 + e_int / ps.cv_d
    h_tot = Q[:, 4] / rho + ps.R_d * T
    nu4 = (auxh[:, lawh.off_delta, :] / 2) ** 4 / 2 / tau

    def scalar_tendency(H, s):
        law = BL.AdvectionDiffusion(3, _NodalHyper(H, s), (BL.HomogeneousBC(3), BL.HomogeneousBC(3)),
                                    advection=False, diffusion=False, hyperdiffusion=True)
        dg = make_dg(law, grid, 0, 1, 1)          # central first-order flux: there is none anyway
        q = np.zeros((grid.nelem, 1, grid.Np))
        q[:, 0, :] = s
        t = np.zeros_like(q)
        dg(t, q, 0.0, 1.0, 0.0)
        return t[:grid.nreal, 0, :]

    expect = np.zeros_like(D)
    for c in range(3):
        expect[:, 1 + c, :] = scalar_tendency(rho * nu4, u_h[:, c, :])
        expect[:, 4, :] += scalar_tendency(nu4 * Q[:, 1 + c, :], u_h[:, c, :])
    expect[:, 4, :] += scalar_tendency(rho * nu4, h_tot)
    res = []
    for s in range(5):
        scale = max(np.abs(expect[:, s]).max(), 1e-300)
        res.append(float(np.abs(D[:, s] - expect[:, s]).max() / scale) if s else
                   float(np.abs(D[:, 0]).max()))
    # how large the hyperdiffusive part is next to the rounding of the full tendency it was
    # differenced from: the check below is only as sharp as this ratio allows
    cond = [float(np.abs(Th[:grid.nreal, s]).max() / max(np.abs(expect[:, s]).max(), 1e-300))
            for s in range(5)]
    return res, cond
