"""Hydrostatic Boussinesq ocean model on the GPU: right-hand side with the law's
update_auxiliary_state! / update_auxiliary_state_gradient! hooks (filters, column integrals)
against the oracle, multi-rank against single-rank, and the reference's regression
(test_3D_spindown.jl + StateCheck refvals) run end to end on the device.  ``-m gpu``."""
import numpy as np
import pytest

from helpers import ocean_gyre_setup, ocean_spindown_setup, ocean_windstress_setup, rel_linf
from test_ocean_oracle import GOLD, GYRE, WIND, check_against_refvals, check_gyre_refvals

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "no HIP device: the product path has no CPU fallback"
    return t


def _gpu(torch, a):
    x = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    torch.cuda.synchronize()
    return x


def _perturbed(law, grid, aux, seed=5):
    Q = law.init_state_prognostic(grid, aux, 3600.0)
    rng = np.random.default_rng(seed)
    Q[:, 0:2] += 0.05 * rng.standard_normal(Q[:, 0:2].shape)
    Q[:, 3] = 10.0 + rng.standard_normal(Q[:, 3].shape)
    return Q


@pytest.mark.parametrize("variant", ["spindown", "advection_rotating"])
def test_ocean_tendency_and_aux_match_oracle(cm, oracle, torch, variant):
    O, F = cm.ocean, cm.mesh.filters
    law, grid = ocean_spindown_setup(Nx=3, Ny=2, Nz=3)
    law0 = law            # the analytic state is restated for the non-rotating box only
    if variant == "advection_rotating":
        law = O.HydrostaticBoussinesqModel(
            O.SimpleBox(1e6, 1e6, 400.0, rotation=O.ROTATING,
                        BC=(O.OceanBC(O.IMPENETRABLE_NOSLIP), O.OceanBC(O.PENETRABLE_FREESLIP))),
            momentum_advection=True, tracer_advection=True, c_h=1.0, c_z=0.1)
    odg = oracle.OracleDGModel(law, grid)
    oracle.hydrostatic_boussinesq_hooks(odg, F.CutoffFilter(grid, 3), F.ExponentialFilter(grid, 1, 8))
    dg = cm.dgmodel.DGModel(law, grid)
    keep = O.install_hydrostatic_boussinesq_hooks(dg)
    Q0 = _perturbed(law0, grid, odg.state_auxiliary)
    T0 = np.random.default_rng(1).standard_normal(Q0.shape)
    Qo, To = Q0.copy(), T0.copy()
    odg(To, Qo, 0.0, 1.0, 1.0)
    Qg, Tg = _gpu(torch, Q0), _gpu(torch, T0)
    dg(Tg, Qg, 0.0, 1.0, 1.0)
    # the filters of update_auxiliary_state! act on Q itself
    assert np.array_equal(Qg.cpu().numpy(), Qo) and not np.array_equal(Qo, Q0)
    auxg = dg.state_auxiliary.cpu().numpy()
    for c, name in ((1, "w"), (2, "pkin"), (3, "wz0")):
        sc = max(np.abs(odg.state_auxiliary[:, c]).max(), 1e-300)
        assert np.abs(auxg[:, c] - odg.state_auxiliary[:, c]).max() / sc < TOL, name
    Tn = Tg.cpu().numpy()
    for s in range(4):
        assert rel_linf(Tn[:, s], To[:, s]) < TOL, s
    assert rel_linf(dg.state_gradient_flux.cpu().numpy(), odg.state_gradient_flux) < TOL
    dg.set_rhs_hooks()
    for f in keep:
        f.close()
    dg.close()


def test_ocean_local_multirank_matches_single_rank(cm, torch):
    O = cm.ocean
    RKA, RKB, RKC = cm.odesolvers.LSRK144_COEFFICIENTS
    law, grid = ocean_spindown_setup(Nx=4, Ny=3, Nz=3)
    dg1 = cm.dgmodel.DGModel(law, grid)
    k1 = O.install_hydrostatic_boussinesq_hooks(dg1)
    Q1h = _perturbed(law, grid, dg1.state_auxiliary.cpu().numpy())
    gl1 = grid.topology.globalelems
    byglobal = {int(g): Q1h[i] for i, g in enumerate(gl1[:grid.nreal])}
    Q1 = _gpu(torch, Q1h)
    dQ1 = torch.zeros_like(Q1)
    dg1.lsrk_run(Q1, dQ1, 0.0, 60.0, 2, RKA, RKB, RKC)
    dg1.synchronize()
    ref = {int(g): Q1[i].cpu().numpy() for i, g in enumerate(gl1[:grid.nreal])}
    size = 3
    dgs, Qs, grids, keeps = [], [], [], []
    for r in range(size):
        lawr, gridr = ocean_spindown_setup(Nx=4, Ny=3, Nz=3, rank=r, size=size)
        d = cm.dgmodel.DGModel(lawr, gridr)
        keeps.append(O.install_hydrostatic_boussinesq_hooks(d))
        q = np.full((gridr.nelem, 4, gridr.Np), np.nan)
        for i, g in enumerate(gridr.topology.globalelems[:gridr.nreal]):
            q[i] = byglobal[int(g)]
        dgs.append(d)
        grids.append(gridr)
        Qs.append(_gpu(torch, q))
    cm.dgmodel.connect_local(dgs)
    dQs = [torch.zeros_like(q) for q in Qs]
    torch.cuda.synchronize()
    cm.dgmodel.group_lsrk_run(dgs, Qs, dQs, 0.0, 60.0, 2, RKA, RKB, RKC)
    for d in dgs:
        d.synchronize()
    for gr, q in zip(grids, Qs):
        qn = q.cpu().numpy()
        for i, g in enumerate(gr.topology.globalelems[:gr.nreal]):
            for s in range(4):
                sc = max(np.abs(ref[int(g)][s]).max(), 1e-6)
                assert np.abs(qn[i, s] - ref[int(g)][s]).max() / sc < 1e-11, (s, i)
    for d in dgs + [dg1]:
        d.set_rhs_hooks()
        d.close()


def test_spindown_reference_regression_on_the_gpu(cm, torch):
    """test_3D_spindown.jl end to end: 720 LSRK144 steps, error against the analytic solution
    and the StateCheck table."""
    O = cm.ocean
    law, grid = ocean_spindown_setup()
    dg = cm.dgmodel.DGModel(law, grid)
    keep = O.install_hydrostatic_boussinesq_hooks(dg)
    Q = dg.init_ode_state(0.0)
    solver = cm.odesolvers.LSRK144NiegemannDiehlBusch(dg, Q, dt=120.0)
    solver.dostep(Q, nsteps=720)
    dg.synchronize()
    Qe = dg.init_ode_state(86400.0)
    err = dg.euclidean_distance(Q, Qe) / dg.norm(Qe)
    assert err < 0.005
    assert abs(err - GOLD["error_printed_by_reference"]) < 1e-10
    check_against_refvals(Q.cpu().numpy(), dg.state_auxiliary.cpu().numpy(), rtol=5e-12)
    dg.set_rhs_hooks()
    for f in keep:
        f.close()
    dg.close()


def test_ocean_gyre_short_reference_regression_on_the_gpu(cm, oracle, torch):
    """test_ocean_gyre_short.jl end to end on the device + tendency parity with the oracle for
    the wind-stress / temperature-flux / no-slip boundary conditions."""
    O, F = cm.ocean, cm.mesh.filters
    law, grid = ocean_gyre_setup()
    dg = cm.dgmodel.DGModel(law, grid)
    keep = O.install_hydrostatic_boussinesq_hooks(dg)
    odg = oracle.OracleDGModel(law, grid)
    oracle.hydrostatic_boussinesq_hooks(odg, F.CutoffFilter(grid, 3), F.ExponentialFilter(grid, 1, 8))
    Q0 = law.init_state_prognostic(grid, odg.state_auxiliary, 0.0)
    rng = np.random.default_rng(2)
    Q0[:, 0:2] = 0.05 * rng.standard_normal(Q0[:, 0:2].shape)
    Q0[:, 2] = 0.1 * rng.standard_normal(Q0[:, 2].shape)
    To = np.zeros_like(Q0)
    odg(To, Q0.copy(), 0.0, 1.0, 0.0)
    Tg = _gpu(torch, np.zeros_like(Q0))
    dg(Tg, _gpu(torch, Q0), 0.0, 1.0, 0.0)
    Tn = Tg.cpu().numpy()
    for s in range(4):
        assert rel_linf(Tn[:, s], To[:, s]) < TOL, s
    Q = dg.init_ode_state(0.0)
    solver = cm.odesolvers.LSRK144NiegemannDiehlBusch(dg, Q, dt=120.0)
    solver.dostep(Q, nsteps=30)
    dg.synchronize()
    assert check_gyre_refvals(Q.cpu().numpy(), dg.state_auxiliary.cpu().numpy()) == 32
    dg.set_rhs_hooks()
    for f in keep:
        f.close()
    dg.close()


def test_windstress_short_reference_regression_on_the_gpu(cm, torch):
    O = cm.ocean
    law, grid = ocean_windstress_setup()
    dg = cm.dgmodel.DGModel(law, grid)
    keep = O.install_hydrostatic_boussinesq_hooks(dg)
    Q = dg.init_ode_state(0.0)
    solver = cm.odesolvers.LSRK144NiegemannDiehlBusch(dg, Q, dt=180.0)
    solver.dostep(Q, nsteps=20)
    dg.synchronize()
    table = [r for r in WIND["explicit_cpu"] if r[1] != "θ"]
    assert check_gyre_refvals(Q.cpu().numpy(), dg.state_auxiliary.cpu().numpy(), table=table, rtol=2e-11) == 28
    dg.set_rhs_hooks()
    for f in keep:
        f.close()
    dg.close()


def test_ocean_courant_numbers_and_dt_match_oracle(cm, oracle, torch):
    """src/Ocean/HydrostaticBoussinesq/Courant.jl: the four local Courant numbers in the
    directions calculate_dt uses them, and the resulting time step."""
    O, F = cm.ocean, cm.mesh.filters
    law, grid = ocean_gyre_setup()
    dg = cm.dgmodel.DGModel(law, grid)
    odg = oracle.OracleDGModel(law, grid)
    Q0 = law.init_state_prognostic(grid, odg.state_auxiliary, 0.0)
    rng = np.random.default_rng(4)
    Q0[:, 0:2] = 0.05 * rng.standard_normal(Q0[:, 0:2].shape)
    w = 1e-3 * rng.standard_normal(Q0[:, 0].shape)
    odg.state_auxiliary[:, 1] = w
    dg.state_auxiliary[:, 1] = _gpu(torch, w)
    Qg = _gpu(torch, Q0)
    for kind in (0, 1, 2, 3):
        for d in (0, 1, 2):
            o = oracle.courant(kind, odg, Q0, 7.0, 0.0, d)
            g = dg.courant(kind, Qg, 7.0, 0.0, d)
            assert abs(g - o) <= 1e-13 * abs(o), (kind, d, g, o)
    dt_o = law.calculate_dt(lambda k, d: oracle.courant(k, odg, Q0, 1.0, 0.0, d), 0.4)
    dt_g = law.calculate_dt(lambda k, d: dg.courant(k, Qg, 1.0, 0.0, d), 0.4)
    assert abs(dt_g - dt_o) <= 1e-13 * dt_o
    dg.close()


def test_ocean_gyre_long_reference_regression_on_the_gpu(cm, torch):
    """test_ocean_gyre_long.jl: 20^3 elements, 4e6 x 4e6 x 1000 m, one simulated day of explicit
    LSRK144 steps with dt = calculate_dt(Courant number 0.4) adjusted to divide the day (620
    steps of 139.35 s), against the `long` StateCheck rows."""
    O = cm.ocean
    law, grid = ocean_gyre_setup(Nx=20, Ny=20, Nz=20, L=4e6)
    dg = cm.dgmodel.DGModel(law, grid)
    keep = O.install_hydrostatic_boussinesq_hooks(dg)
    Q = dg.init_ode_state(0.0)
    dt = law.calculate_dt(lambda k, d: dg.courant(k, Q, 1.0, 0.0, d), 0.4)
    nsteps = int(np.ceil(86400.0 / dt))            # solver_configs.jl:247-249
    assert nsteps == 620
    dt = 86400.0 / nsteps
    solver = cm.odesolvers.LSRK144NiegemannDiehlBusch(dg, Q, dt=dt)
    solver.dostep(Q, nsteps=nsteps)
    dg.synchronize()
    n = check_gyre_refvals(Q.cpu().numpy(), dg.state_auxiliary.cpu().numpy(), rtol=1e-10,
                           table=GYRE["long"])
    assert n == 32
    dg.set_rhs_hooks()
    for f in keep:
        f.close()
    dg.close()


def test_fused_column_operators_equal_the_recorded_sequence(cm, torch, monkeypatch):
    """The column operators of the recorded update_auxiliary_state_gradient! composition run as
    one launch (columns.h k_column_chain: copy + upward integrals + reverse integral + surface
    value) and the flow deviation as one (k_flow_deviation); CMDG_FUSED_COLUMNS=0 issues them one
    by one.  Same arithmetic in the same order: state and every auxiliary column bit for bit,
    for the uncoupled gyre and for the coupled box with its flow deviation."""
    from helpers import split_explicit_setup
    O = cm.ocean
    for coupled in (False, True):
        res = []
        for fused in ("0", "2"):
            monkeypatch.setenv("CMDG_FUSED_COLUMNS", fused)
            if coupled:
                law, grid, _, _ = split_explicit_setup(True, Nx=4, Ny=3, Nz=3)
            else:
                law, grid = ocean_gyre_setup()
            dg = cm.dgmodel.DGModel(law, grid)
            keep = O.install_hydrostatic_boussinesq_hooks(dg)
            rng = np.random.default_rng(9)
            Q0 = law.init_state_prognostic(grid, dg.state_auxiliary.cpu().numpy(), 0.0)
            Q0[:, 0:2] += 0.05 * rng.standard_normal(Q0[:, 0:2].shape)
            Q0[:, 3] += 0.1 * rng.standard_normal(Q0[:, 3].shape)
            Q = _gpu(torch, Q0)
            solver = cm.odesolvers.LSRK144NiegemannDiehlBusch(dg, Q, dt=60.0)
            solver.dostep(Q, nsteps=3)
            dg.synchronize()
            res.append((Q.cpu().numpy().copy(), dg.state_auxiliary.cpu().numpy().copy()))
            dg.set_rhs_hooks()
            for f in keep:
                f.close()
            dg.close()
        nr = grid.nreal
        assert np.isfinite(res[0][0][:nr]).all()
        assert np.array_equal(res[0][0][:nr], res[1][0][:nr])
        assert np.array_equal(res[0][1][:nr], res[1][1][:nr])
