"""``courant`` / ``min_node_distance`` / ``calculate_dt`` on the GPU against the oracle and
the analytic values of the reference's test/Numerics/DGMethods/courant.jl.  ``-m gpu``."""
import numpy as np
import pytest

from helpers import courant_test_setup, held_suarez_setup, pseudo1d_setup
from test_courant_oracle import _expected

pytestmark = pytest.mark.gpu
EVERY, HORZ, VERT = 0, 1, 2


@pytest.fixture(scope="module")
def torch():
    import torch as t
    assert t.cuda.is_available(), "no HIP device: the product path has no CPU fallback"
    return t


def test_courant_matches_reference_test_and_oracle(cm, oracle, torch):
    law, grid, setup = courant_test_setup()
    dg = cm.dgmodel.DGModel(law, grid)
    odg = oracle.OracleDGModel(law, grid)
    Q = dg.init_ode_state(0.0)
    Qh = Q.cpu().numpy()
    dt = 1 / 200
    exp = _expected(law, grid, setup, dt)
    D = cm.dgmodel
    assert dg.courant(D.NONDIFFUSIVE_COURANT, Q, dt, 0.0, HORZ) == pytest.approx(exp["c_h"], rel=1e-4)
    assert dg.courant(D.NONDIFFUSIVE_COURANT, Q, dt, 0.0, VERT) == pytest.approx(exp["c_v"], rel=1e-4)
    rt = np.sqrt(np.finfo(float).eps)
    assert dg.courant(D.DIFFUSIVE_COURANT, Q, dt, 0.0, HORZ) == pytest.approx(exp["d_h"], rel=rt)
    assert dg.courant(D.DIFFUSIVE_COURANT, Q, dt, 0.0, VERT) == pytest.approx(exp["d_v"], rel=rt)
    for kind in (0, 1, 2):
        for d in (EVERY, HORZ, VERT):
            o = oracle.courant(kind, odg, Qh, dt, 0.0, d)
            g = dg.courant(kind, Q, dt, 0.0, d)
            assert abs(g - o) <= 1e-12 * abs(o) + 1e-300, (kind, d, g, o)
    for d in (EVERY, HORZ, VERT):
        assert dg.min_node_distance(d) == cm.mesh.grids.min_node_distance(grid, d)
    dt2 = dg.calculate_dt(Q, 0.4)
    assert dt2 == pytest.approx(oracle.calculate_dt(odg, Qh, 0.4), rel=1e-12)
    assert dg.courant(D.NONDIFFUSIVE_COURANT, Q, dt2) == pytest.approx(0.4, rel=1e-13)
    dg.close()


def test_courant_on_held_suarez_sphere(cm, oracle, torch):
    law, grid, d, dd = held_suarez_setup(n_horz=3, n_vert=2)
    dg = cm.dgmodel.DGModel(law, grid, direction=d, diffusion_direction=dd)
    odg = oracle.OracleDGModel(law, grid, direction=d, diffusion_direction=dd)
    Q = dg.init_ode_state(0.0)
    Qh = Q.cpu().numpy()
    for kind in (0, 1, 2):
        for dr in (EVERY, HORZ, VERT):
            o = oracle.courant(kind, odg, Qh, 2.5, 10.0, dr)
            g = dg.courant(kind, Q, 2.5, 10.0, dr)
            assert abs(g - o) <= 1e-12 * abs(o) + 1e-300, (kind, dr, g, o)
    # acoustic limit of the experiment: dt = CFL dx_v / c_s
    assert 0.5 < dg.calculate_dt(Q, 0.5, 0.0, VERT) < 50.0
    dg.close()


def test_law_without_courant_fails_loudly(cm, torch):
    law, grid, _ = pseudo1d_setup(Ne=2)
    dg = cm.dgmodel.DGModel(law, grid)
    Q = dg.init_ode_state(0.0)
    with pytest.raises(cm._lib.CmdgError):
        dg.courant(cm.dgmodel.NONDIFFUSIVE_COURANT, Q, 0.1)
    assert dg.min_node_distance() == cm.mesh.grids.min_node_distance(grid)
    dg.close()
