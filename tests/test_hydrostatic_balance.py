"""Discrete hydrostatic balance of the reference state (src/Atmos/Model/ref_state.jl:150-175:
rho_ref from the DG gradient of p_ref) -- test/Atmos/Model/discrete_hydrostatic_balance.jl:
a state initialised to the reference state is steady to 100 eps over 100 s (central, Roe and
HLLC fluxes, LSRK54 at Courant number 0.1), in the LES box and on the GCM sphere, for the isothermal and the
decaying temperature profile.  CPU tests use the oracle; the ``gpu`` ones libcmdg, including
the device-side evaluation of the PressureGradientModel."""
import numpy as np
import pytest

from cmdg_loader import cm

M, A = cm.mesh, cm.atmos
EPS = np.finfo(float).eps
H = 50e3


class _InitToRefState:
    """init_to_ref_state! (discrete_hydrostatic_balance.jl:24-29)."""

    def __call__(self, law, aux, coord, t):
        rho = aux[:, law.off_ref, :]
        z = 0.0 * rho
        return rho, [z, z.copy(), z.copy()], aux[:, law.off_ref + 3, :]


def balanced_setup(config, profile, balance=True, rank=0, size=1):
    ps = A.PlanetParameters()
    prof = (A.DecayingTemperatureProfile(ps, 290.0, 290.0, 8e3) if profile == "isothermal"
            else A.DecayingTemperatureProfile(ps, 290.0, 220.0, 8e3))
    if config == "LES":          # resolution = H / (3 N): three elements of order 4 per side
        rng = [np.linspace(0.0, H, 4)] * 3
        topl = M.StackedBrickTopology(rng, periodicity=(True, True, False),
                                      boundary=((0, 0), (0, 0), (1, 2)), rank=rank, size=size)
        grid = M.DiscontinuousSpectralElementGrid(topl, 4)
        orient = A.ORIENT_FLAT
    else:                        # AtmosGCMConfiguration, (nelem_horz, nelem_vert) = (3, 3)
        a = ps.planet_radius
        topl = M.StackedCubedSphereTopology(3, np.linspace(a, a + H, 4), boundary=(1, 2),
                                            rank=rank, size=size)
        grid = M.DiscontinuousSpectralElementGrid(topl, 4, meshwarp=M.equiangular_cubed_sphere_warp)
        orient = A.ORIENT_SPHERICAL
    law = A.DryAtmosModel(_InitToRefState(), orientation=orient, ref_state=prof, subtract_off=False,
                          viscosity=0.0, dynamic_viscosity=True, sources=A.SRC_GRAVITY,
                          boundary_conditions=(A.BC_ATMOS_DEFAULT, A.BC_ATMOS_DEFAULT),
                          param_set=ps, discrete_hydrostatic_balance=balance)
    return law, grid


@pytest.mark.parametrize("nf", [1, 2, 3])      # Central, Roe, HLLC (:97-101)
@pytest.mark.parametrize("profile", ["isothermal", "decaying"])
@pytest.mark.parametrize("config", ["LES", "GCM"])
def test_balanced_state_is_steady_oracle(oracle, config, profile, nf):
    law, grid = balanced_setup(config, profile)
    dg = oracle.OracleDGModel(law, grid, nf_first=nf, direction=0, diffusion_direction=1)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    Q0 = Q.copy()
    T = np.zeros_like(Q)
    dg(T, Q, 0.0)
    # the vertical momentum equation is balanced to rounding (the analytic density leaves 5e-2)
    assert np.abs(T[: grid.nreal, 1:4]).max() < 1e-10
    dt = oracle.calculate_dt(dg, Q, 0.1)
    oracle.solve(dg, Q, dt, 100.0)
    err = np.sqrt(oracle.weighted_norm2_local(grid, Q, Q0) / oracle.weighted_norm2_local(grid, Q0))
    assert err <= 100 * EPS, err / EPS                  # discrete_hydrostatic_balance.jl:139


def test_rebalanced_density_is_close_to_the_analytic_one(oracle):
    law, grid = balanced_setup("GCM", "decaying")
    law0, _ = balanced_setup("GCM", "decaying", balance=False)
    a1 = oracle.OracleDGModel(law, grid).state_auxiliary
    a0 = oracle.OracleDGModel(law0, grid).state_auxiliary
    o = law.off_ref
    rel = np.abs(a1[:, o] - a0[:, o]) / a0[:, o]
    assert 1e-8 < rel[: grid.nreal].max() < 0.2         # discretisation error (3 elements over 50 km)
    assert np.array_equal(a1[:, o + 1], a0[:, o + 1])   # the pressure is untouched


@pytest.mark.gpu
@pytest.mark.parametrize("nf", [1, 2, 3])      # Central, Roe, HLLC (:97-101)
@pytest.mark.parametrize("config,profile", [("LES", "decaying"), ("GCM", "isothermal"),
                                            ("GCM", "decaying")])
def test_balanced_state_is_steady_gpu(oracle, config, profile, nf):
    import torch
    law, grid = balanced_setup(config, profile)
    dg = cm.dgmodel.DGModel(law, grid, numerical_flux_first_order=nf, direction=0,
                            diffusion_direction=1)
    odg = oracle.OracleDGModel(law, grid, nf_first=nf, direction=0, diffusion_direction=1)
    # device-side PressureGradientModel == oracle's
    aux = dg.state_auxiliary.cpu().numpy()
    o = law.off_ref
    for c in (o, o + 2, o + 3):
        sc = np.abs(odg.state_auxiliary[:, c]).max()
        assert np.abs(aux[:, c] - odg.state_auxiliary[:, c]).max() / sc < 1e-13
    Q = dg.init_ode_state(0.0)
    Q0 = Q.clone()
    dt = dg.calculate_dt(Q, 0.1)
    solver = cm.odesolvers.LSRK54CarpenterKennedy(dg, Q, dt=dt)
    cm.odesolvers.solve(Q, solver, timeend=100.0)
    err = dg.euclidean_distance(Q, Q0) / dg.norm(Q0)
    assert err <= 100 * EPS, err / EPS
    assert bool(torch.isfinite(Q).all())
    dg.close()


@pytest.mark.gpu
def test_group_halo_carries_the_rebalanced_reference_state():
    """multi-rank: the re-balanced columns live on real elements; one ghost exchange of the
    auxiliary state (as the reference does after init) makes ghosts equal to their owners."""
    import torch
    law1, grid1 = balanced_setup("GCM", "decaying")
    dg1 = cm.dgmodel.DGModel(law1, grid1, numerical_flux_first_order=1)
    a1 = dg1.state_auxiliary.cpu().numpy()
    byglobal = {int(g): a1[i] for i, g in enumerate(grid1.topology.globalelems[: grid1.nreal])}
    size = 3
    dgs, grids, cols = [], [], []
    o = law1.off_ref
    for r in range(size):
        law, grid = balanced_setup("GCM", "decaying", rank=r, size=size)
        dgs.append(cm.dgmodel.DGModel(law, grid, numerical_flux_first_order=1))
        grids.append(grid)
        cols.append(dgs[-1].state_auxiliary[:, [o, o + 2, o + 3], :].contiguous())
    cm.dgmodel.connect_local(dgs)
    torch.cuda.synchronize()
    cm.dgmodel.group_halo(dgs, cols)
    for d, grid, c in zip(dgs, grids, cols):
        cn = c.cpu().numpy()
        gl = grid.topology.globalelems
        # real elements: identical to the single-rank model
        for i in range(grid.nreal):
            assert np.allclose(cn[i], byglobal[int(gl[i])][[o, o + 2, o + 3]], rtol=1e-13, atol=0)
        # ghost face nodes now carry the owner's re-balanced values
        recv = np.asarray(grid.vmaprecv) - 1
        e, n = recv // grid.Np, recv % grid.Np
        for k in range(0, len(recv), 7):
            want = byglobal[int(gl[e[k]])][[o, o + 2, o + 3], n[k]]
            assert np.allclose(cn[e[k], :, n[k]], want, rtol=1e-13, atol=0)
    for d in dgs + [dg1]:
        d.close()
