"""Element filters on the GPU (through ``cmdg_filter_*`` of the C ABI) against the oracle's
restatement of the reference kernels.  The kernels keep the reference's summation order and
launch-boundary arithmetic, so agreement is to the last bit; the asserted bound is the
north-star 1e-12.  Needs a real MI355X: ``-m gpu``."""
import numpy as np
import pytest

from helpers import held_suarez_setup, pseudo1d_setup, rel_linf
from test_filters_oracle import _filter_test_state, _weightedsum

pytestmark = pytest.mark.gpu
TOL = 1e-12
EVERY, HORZ, VERT = 0, 1, 2


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "no HIP device: the product path has no CPU fallback"
    return t


def _gpu(torch, a):
    x = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    torch.cuda.synchronize()
    return x


@pytest.fixture(scope="module")
def box(cm, torch):
    law, grid, _ = pseudo1d_setup(Ne=3)
    dg = cm.dgmodel.DGModel(law, grid)
    yield law, grid, dg
    dg.close()


@pytest.fixture(scope="module")
def sphere(cm, torch):
    law, grid, d, dd = held_suarez_setup(n_horz=3, n_vert=2)
    dg = cm.dgmodel.DGModel(law, grid, direction=d, diffusion_direction=dd)
    yield law, grid, dg
    dg.close()


@pytest.mark.parametrize("kind", ["CutoffFilter", "ExponentialFilter", "BoydVandevenFilter",
                                  "MassPreservingCutoffFilter"])
@pytest.mark.parametrize("direction", [EVERY, HORZ, VERT])
def test_spectral_filters_match_oracle(cm, oracle, torch, box, direction, kind):
    """filter.jl:199-330 at N = 4: states 1 and 3 lose their high modes, 2 and 4 stay."""
    F = cm.mesh.filters
    law, grid, dg = box
    filt = {"CutoffFilter": lambda: F.CutoffFilter(grid, 2),
            "MassPreservingCutoffFilter": lambda: F.MassPreservingCutoffFilter(grid, 2),
            "ExponentialFilter": lambda: F.ExponentialFilter(grid, 1, 8),
            "BoydVandevenFilter": lambda: F.BoydVandevenFilter(grid, 1, 8)}[kind]()
    Q0 = _filter_test_state(grid, None)
    Q0 += 1e-3 * np.random.default_rng(3).standard_normal(Q0.shape)
    Qo = Q0.copy()
    oracle.apply_filter(Qo, F.FilterIndices(1, 3), grid, filt, direction=direction)
    Q = _gpu(torch, Q0)
    F.apply(Q, (1, 3), dg, filt, direction=direction)
    Qg = Q.cpu().numpy()
    assert rel_linf(Qg, Qo) < TOL
    assert np.array_equal(Qg, Qo), np.abs(Qg - Qo).max()       # same order of operations
    assert np.array_equal(Qg[:, 1], Q0[:, 1]) or kind == "MassPreservingCutoffFilter"


@pytest.mark.parametrize("kind", ["CutoffFilter", "MassPreservingCutoffFilter"])
@pytest.mark.parametrize("direction", [EVERY, HORZ, VERT])
def test_cutoff_filter_analytic(cm, torch, direction, kind):
    """the analytic statement of filter.jl:199-330 on one element [-1, 1]^3 (N = 4)."""
    F = cm.mesh.filters
    law, grid, _ = pseudo1d_setup(Ne=1)
    dg = cm.dgmodel.DGModel(law, grid)
    filt = getattr(F, kind)(grid, 2)
    Qa = _gpu(torch, _filter_test_state(grid, None))
    F.apply(Qa, (1, 3), dg, filt, direction=direction)
    P = _filter_test_state(grid, direction)
    assert np.abs(Qa.cpu().numpy() - P).max() < 5e-13
    dg.close()


def test_filter_many_states_and_colon_target(cm, oracle, torch, box):
    """more filtered states than one launch stages in LDS (chunks of 16); target ``:``."""
    F = cm.mesh.filters
    law, grid, dg = box
    filt = F.ExponentialFilter(grid, 0, 4)
    Q0 = np.random.default_rng(5).standard_normal((grid.nelem, 21, grid.Np))
    Qo = Q0.copy()
    oracle.apply_filter(Qo, F.FilterIndices(range(1, 22)), grid, filt)
    Q = _gpu(torch, Q0)
    F.apply(Q, None, dg, filt)
    assert np.array_equal(Q.cpu().numpy(), Qo)
    with pytest.raises(cm._lib.CmdgError):
        F.apply(Q, (1, 22), dg, filt)             # index beyond nstate


@pytest.mark.parametrize("target", [(1,), None])
def test_tmar_filter_matches_oracle(cm, oracle, torch, box, target):
    """filter.jl:349-399: non-negative afterwards, weighted sum kept."""
    F = cm.mesh.filters
    law, grid, dg = box
    x = grid.vgeo[:, 12, :]
    Q0 = np.ascontiguousarray(np.stack([np.abs(x) - 0.1, 0.3 - np.abs(x)], axis=1))
    idx = F.FilterIndices(*target) if target else F.FilterIndices(range(1, 3))
    Qo = Q0.copy()
    oracle.apply_filter(Qo, idx, grid, F.TMARFilter())
    Q = _gpu(torch, Q0)
    F.apply(Q, target, dg, F.TMARFilter())
    Qg = Q.cpu().numpy()
    assert np.array_equal(Qg, Qo)
    assert Q0[:, 0].min() < 0 and Qg[:, 0].min() >= 0
    b, a = _weightedsum(grid, Q0, 0), _weightedsum(grid, Qg, 0)
    assert abs(a - b) <= 10 * np.finfo(float).eps * abs(b)
    if target:
        assert np.array_equal(Qg[:, 1], Q0[:, 1])


@pytest.mark.parametrize("direction", [EVERY, HORZ, VERT])
@pytest.mark.parametrize("tname", ["AtmosFilterPerturbations", "AtmosSpecificFilterPerturbations"])
def test_atmos_targets_match_oracle(cm, oracle, torch, sphere, tname, direction):
    """the Held-Suarez every-step filter (heldsuarez.jl:261-272) and its specific variant."""
    F = cm.mesh.filters
    law, grid, dg = sphere
    aux = dg.state_auxiliary.cpu().numpy()
    Q0 = law.init_state_prognostic(grid, aux, 0.0)
    rng = np.random.default_rng(11)
    Q0 = Q0 * (1 + 1e-3 * rng.standard_normal(Q0.shape))
    Q0[:, 1:4] += 5.0 * rng.standard_normal(Q0[:, 1:4].shape)
    filt = F.ExponentialFilter(grid, 0, 20)
    tg = getattr(F, tname)(law)
    Qo = Q0.copy()
    oracle.apply_filter(Qo, tg, grid, filt, direction=direction, state_auxiliary=aux)
    Q = _gpu(torch, Q0)
    F.apply(Q, tg, dg, filt, direction=direction, state_auxiliary=dg.state_auxiliary)
    Qg = Q.cpu().numpy()
    assert rel_linf(Qg, Qo) < TOL
    assert np.array_equal(Qg, Qo), np.abs(Qg - Qo).max()
    assert not np.array_equal(Qg[: grid.nreal], Q0[: grid.nreal])


def test_mass_preserving_filter_on_sphere(cm, oracle, torch, sphere):
    """filter.jl:440-509 on the Held-Suarez grid: element means are restored."""
    F = cm.mesh.filters
    law, grid, dg = sphere
    Q0 = np.random.default_rng(2).standard_normal((grid.nelem, 5, grid.Np)) + 3.0
    for cls, conserved in (("MassPreservingCutoffFilter", True), ("CutoffFilter", False)):
        filt = getattr(F, cls)(grid, 2)
        Qo = Q0.copy()
        oracle.apply_filter(Qo, F.FilterIndices(range(1, 4)), grid, filt)
        Q = _gpu(torch, Q0)
        F.apply(Q, range(1, 4), dg, filt)
        Qg = Q.cpu().numpy()
        assert np.array_equal(Qg, Qo), np.abs(Qg - Qo).max()
        for s in range(3):
            b, a = _weightedsum(grid, Q0, s), _weightedsum(grid, Qg, s)
            assert (abs(a - b) <= 1e-12 * abs(b)) == conserved


def test_gradient_and_tendency_filters_in_the_operator(cm, oracle, torch):
    """DGModel(...; gradient_filter, tendency_filter): DGModel.jl:185-193, 417-425."""
    F = cm.mesh.filters
    law, grid, _ = pseudo1d_setup(Ne=3)
    dg = cm.dgmodel.DGModel(law, grid)
    odg = oracle.OracleDGModel(law, grid)
    gfilt, tfilt = F.CutoffFilter(grid, 3), F.ExponentialFilter(grid, 1, 4)
    gt, tt = F.FilterIndices(range(1, law.ngradflux + 1)), F.FilterIndices(range(1, law.ns + 1))
    odg.gradient_filter, odg.tendency_filter = (gfilt, gt), (tfilt, tt)
    gdev = F.make_device_filter(dg, gfilt, gt, nstate=law.ngradflux)
    tdev = F.make_device_filter(dg, tfilt, tt)
    dg.set_filters(gradient_filter=gdev, tendency_filter=tdev)
    Q0 = law.init_state_prognostic(grid, odg.state_auxiliary, 0.0)
    Q0 = Q0 + 1e-3 * np.random.default_rng(7).standard_normal(Q0.shape)
    T0 = np.random.default_rng(8).standard_normal(Q0.shape)
    To = T0.copy()
    odg(To, Q0.copy(), 0.1, 1.0, 1.0)
    Tg = _gpu(torch, T0)
    dg(Tg, _gpu(torch, Q0), 0.1, 1.0, 1.0)
    assert rel_linf(Tg.cpu().numpy(), To) < TOL
    assert rel_linf(dg.state_gradient_flux.cpu().numpy(), odg.state_gradient_flux) < TOL
    # unfiltered operator differs (the filters are not no-ops here)
    odg2 = oracle.OracleDGModel(law, grid)
    T2 = T0.copy()
    odg2(T2, Q0.copy(), 0.1, 1.0, 1.0)
    assert rel_linf(T2, To) > 1e-6
    # LSRK with a tendency filter: rhs!, filter dQ, update! per stage (unfused path)
    Qo = Q0.copy()
    dQo = np.zeros_like(Qo)
    for i in range(3):
        oracle.lsrk54_step(odg, Qo, dQo, 0.01 * i, 0.01)
    Q = _gpu(torch, Q0)
    dQ = torch.zeros_like(Q)
    dg.lsrk_run(Q, dQ, 0.0, 0.01, 3, oracle.RKA, oracle.RKB, oracle.RKC)
    dg.synchronize()
    assert rel_linf(Q.cpu().numpy(), Qo) < TOL
    assert rel_linf(dQ.cpu().numpy(), dQo) < 1e-10
    dg.set_filters()
    gdev.close()
    tdev.close()
    dg.close()


def test_gradient_filter_on_a_node_major_gradient_flux(cm, oracle, torch):
    """The dry atmosphere's ``state_gradient_flux`` is node-major inside the library (cmdg.h,
    ``cmdg_export_gradient_flux``); a ``gradient_filter`` then runs on a reference-layout copy that
    is folded back.  Operator and exported array against the oracle, then the filter removed."""
    from helpers import rising_bubble_setup
    F = cm.mesh.filters
    law, grid = rising_bubble_setup(nx=3, ny=2, nz=3)
    dg = cm.dgmodel.DGModel(law, grid)
    odg = oracle.OracleDGModel(law, grid)
    gfilt = F.ExponentialFilter(grid, 1, 8)
    gt = F.FilterIndices(range(1, law.ngradflux + 1))
    odg.gradient_filter = (gfilt, gt)
    gdev = F.make_device_filter(dg, gfilt, gt, nstate=law.ngradflux)
    dg.set_filters(gradient_filter=gdev)
    Q0 = law.init_state_prognostic(grid, odg.state_auxiliary, 0.0)
    rng = np.random.default_rng(5)
    Q0[:, 1:4] += Q0[:, 0:1] * 2.0 * rng.standard_normal(Q0[:, 1:4].shape)
    To, T1 = np.zeros_like(Q0), np.zeros_like(Q0)
    odg(To, Q0.copy(), 0.2, 1.0, 0.0)
    Tg = _gpu(torch, np.zeros_like(Q0))
    dg(Tg, _gpu(torch, Q0), 0.2, 1.0, 0.0)
    assert rel_linf(Tg.cpu().numpy(), To) < TOL
    gfg = dg.state_gradient_flux.cpu().numpy()
    for s in range(law.ngradflux):
        sc = max(np.abs(odg.state_gradient_flux[:, s]).max(), 1e-300)
        assert np.abs(gfg[:, s] - odg.state_gradient_flux[:, s]).max() / sc < TOL, s
    dg.set_filters()
    odg.gradient_filter = None
    odg(T1, Q0.copy(), 0.2, 1.0, 0.0)
    dg(Tg, _gpu(torch, Q0), 0.2, 1.0, 0.0)
    assert rel_linf(Tg.cpu().numpy(), T1) < TOL
    assert rel_linf(T1, To) > 1e-8          # the filter was not a no-op
    gdev.close()
    dg.close()


def test_step_filter_in_lsrk_run(cm, oracle, torch):
    """Held-Suarez time loop with the exponential filter after every step
    (heldsuarez.jl:261-272), three steps, against the oracle."""
    F = cm.mesh.filters
    law, grid, d, dd = held_suarez_setup(n_horz=2, n_vert=2)
    dg = cm.dgmodel.DGModel(law, grid, direction=d, diffusion_direction=dd)
    odg = oracle.OracleDGModel(law, grid, direction=d, diffusion_direction=dd)
    filt = F.ExponentialFilter(grid, 0, 20)
    tg = F.AtmosFilterPerturbations(law)
    sdev = F.make_device_filter(dg, filt, tg)
    dg.set_filters(step_filter=sdev)
    Q0 = law.init_state_prognostic(grid, odg.state_auxiliary, 0.0)
    rng = np.random.default_rng(4)
    Q0[:, 1:4] += 2.0 * Q0[:, 0:1] * rng.standard_normal(Q0[:, 1:4].shape)    # u ~ 2 m/s
    Qo, dQo = Q0.copy(), np.zeros_like(Q0)
    dt = 0.2
    for i in range(3):
        oracle.lsrk54_step(odg, Qo, dQo, i * dt, dt, step_filter=(filt, tg, EVERY))
    Q = _gpu(torch, Q0)
    dQ = torch.zeros_like(Q)
    dg.lsrk_run(Q, dQ, 0.0, dt, 3, oracle.RKA, oracle.RKB, oracle.RKC)
    dg.synchronize()
    nr = grid.nreal
    assert np.isfinite(Qo).all()
    assert rel_linf(Q.cpu().numpy()[:nr], Qo[:nr]) < TOL
    # and the filter matters: an unfiltered run differs
    dg.set_filters()
    Q2 = _gpu(torch, Q0)
    dQ2 = torch.zeros_like(Q2)
    dg.lsrk_run(Q2, dQ2, 0.0, dt, 3, oracle.RKA, oracle.RKB, oracle.RKC)
    dg.synchronize()
    assert rel_linf(Q2.cpu().numpy()[:nr], Qo[:nr]) > 1e-9
    sdev.close()
    dg.close()
