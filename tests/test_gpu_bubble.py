"""BASELINE config 2 -- dry rising thermal bubble (experiments/TestCase/risingbubble.jl):
SmagorinskyLilly + HydrostaticState(DryAdiabaticProfile) + LSRK144 on the GPU against the
oracle, the multi-rank path against the single-rank one, and the reference's own acceptance
check (norm ratio within 1.5e-3 of one at t = 1000 s) on the reference's 20 x 1 x 20 mesh.
``-m gpu``."""
import numpy as np
import pytest

from helpers import rel_linf, rising_bubble_setup

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


def _moving_state(law, grid, aux, seed=3):
    Q = law.init_state_prognostic(grid, aux, 0.0)
    rng = np.random.default_rng(seed)
    Q[:, 1:4] += Q[:, 0:1] * 3.0 * rng.standard_normal(Q[:, 1:4].shape)
    Q[:, 4] *= 1 + 1e-3 * rng.standard_normal(Q[:, 4].shape)
    return Q


def test_bubble_tendency_matches_oracle(cm, oracle, torch):
    law, grid = rising_bubble_setup(nx=4, ny=2, nz=4)
    odg = oracle.OracleDGModel(law, grid)
    dg = cm.dgmodel.DGModel(law, grid)
    assert rel_linf(dg.state_auxiliary.cpu().numpy(), odg.state_auxiliary) == 0
    Q0 = _moving_state(law, grid, odg.state_auxiliary)
    T0 = np.random.default_rng(1).standard_normal(Q0.shape)
    for alpha, beta in ((1.0, 0.0), (0.5, 2.0)):
        To = T0.copy()
        odg(To, Q0.copy(), 0.3, alpha, beta)
        Tg = _gpu(torch, T0)
        dg(Tg, _gpu(torch, Q0), 0.3, alpha, beta)
        Tn = Tg.cpu().numpy()
        for s in range(5):
            assert rel_linf(Tn[:, s], To[:, s]) < TOL, s
        gfg = dg.state_gradient_flux.cpu().numpy()
        for s in range(law.ngradflux):
            sc = max(np.abs(odg.state_gradient_flux[:, s]).max(), 1e-300)
            assert np.abs(gfg[:, s] - odg.state_gradient_flux[:, s]).max() / sc < TOL, s
    # the refreshed auxiliary entries (theta_v, T) as the reference leaves them
    auxg = dg.state_auxiliary.cpu().numpy()
    assert rel_linf(auxg[:, law.off_moist:], odg.state_auxiliary[:, law.off_moist:]) < TOL
    # Courant numbers with the eddy viscosity of this state
    Qg = _gpu(torch, Q0)
    for kind in (0, 1, 2):
        for d in (0, 1, 2):
            o = oracle.courant(kind, odg, Q0, 0.4, 0.0, d)
            g = dg.courant(kind, Qg, 0.4, 0.0, d)
            assert abs(g - o) <= 1e-11 * abs(o), (kind, d, g, o)
    dg.close()


def test_bubble_lsrk144_matches_oracle(cm, oracle, torch):
    law, grid = rising_bubble_setup(nx=4, nz=4)
    odg = oracle.OracleDGModel(law, grid)
    dg = cm.dgmodel.DGModel(law, grid)
    Q0 = _moving_state(law, grid, odg.state_auxiliary)
    RKA, RKB, RKC = cm.odesolvers.LSRK144_COEFFICIENTS
    Qo, dQo = Q0.copy(), np.zeros_like(Q0)
    dt = 0.1            # node-to-node noise of 3 m/s: keep the acoustic Courant number ~0.4
    for i in range(2):
        oracle.lsrk_step(odg, Qo, dQo, i * dt, dt, RKA, RKB, RKC)
    Q = _gpu(torch, Q0)
    solver = cm.odesolvers.LSRK144NiegemannDiehlBusch(dg, Q, dt=dt)
    solver.dostep(Q, nsteps=2)
    dg.synchronize()
    Qn = Q.cpu().numpy()
    assert np.isfinite(Qo).all()
    for s in range(5):
        sc = np.abs(Qo[:, s]).max()
        assert np.abs(Qn[:, s] - Qo[:, s]).max() / sc < 1e-11, s
    dg.close()


def test_bubble_local_multirank_matches_single_rank(cm, torch):
    law, grid = rising_bubble_setup(nx=4, ny=2, nz=4)
    dg1 = cm.dgmodel.DGModel(law, grid)
    aux1 = dg1.state_auxiliary.cpu().numpy()
    Q1h = _moving_state(law, grid, aux1)
    gl1 = grid.topology.globalelems
    byglobal = {int(g): Q1h[i] for i, g in enumerate(gl1[:grid.nreal])}
    Q1 = _gpu(torch, Q1h)
    RKA, RKB, RKC = cm.odesolvers.LSRK144_COEFFICIENTS
    dQ1 = torch.zeros_like(Q1)
    dg1.lsrk_run(Q1, dQ1, 0.0, 0.1, 2, RKA, RKB, RKC)
    dg1.synchronize()
    ref = {int(g): Q1[i].cpu().numpy() for i, g in enumerate(gl1[:grid.nreal])}
    size = 3
    dgs, Qs, grids = [], [], []
    for r in range(size):
        lawr, gridr = rising_bubble_setup(nx=4, ny=2, nz=4, rank=r, size=size)
        d = cm.dgmodel.DGModel(lawr, gridr)
        q = np.full((gridr.nelem, 5, gridr.Np), np.nan)
        for i, g in enumerate(gridr.topology.globalelems[:gridr.nreal]):
            q[i] = byglobal[int(g)]
        dgs.append(d)
        grids.append(gridr)
        Qs.append(_gpu(torch, q))
    cm.dgmodel.connect_local(dgs)
    dQs = [torch.zeros_like(q) for q in Qs]
    torch.cuda.synchronize()
    cm.dgmodel.group_lsrk_run(dgs, Qs, dQs, 0.0, 0.1, 2, RKA, RKB, RKC)
    for d in dgs:
        d.synchronize()
    for gr, q in zip(grids, Qs):
        qn = q.cpu().numpy()
        for i, g in enumerate(gr.topology.globalelems[:gr.nreal]):
            for s in range(5):
                sc = max(np.abs(ref[int(g)][s]).max(), 1e-3)
                assert np.abs(qn[i, s] - ref[int(g)][s]).max() / sc < 1e-11
    for d in dgs + [dg1]:
        d.close()


def test_rising_bubble_reference_acceptance(cm, torch):
    """The reference's own check of this case (risingbubble.jl:186-233): N = 4, 20 x 1 x 20
    elements, LSRK144 at Courant number 1.7, t_end = 1000 s,
    ``isapprox(norm(Q_end) / norm(Q_0), 1; atol = 1.5e-3)``."""
    law, grid = rising_bubble_setup()
    assert grid.nreal == 400
    dg = cm.dgmodel.DGModel(law, grid)
    Q = dg.init_ode_state(0.0)
    n0 = dg.norm(Q)
    t0, timeend = 0.0, 1000.0
    dt = dg.calculate_dt(Q, 1.7)
    nsteps = int(np.ceil((timeend - t0) / dt))      # cld(timeend - t0, ode_dt)
    dt = (timeend - t0) / nsteps                    # timeend_dt_adjust
    assert 2000 < nsteps < 3000
    solver = cm.odesolvers.LSRK144NiegemannDiehlBusch(dg, Q, dt=dt)
    solver.dostep(Q, nsteps=nsteps)
    dg.synchronize()
    assert bool(torch.isfinite(Q).all())
    ratio = dg.norm(Q) / n0
    assert abs(ratio - 1.0) <= 1.5e-3, ratio
    w = (Q[:, 3] / Q[:, 0])
    assert 5.0 < float(w.max()) < 30.0              # the thermal has risen
    dg.close()
