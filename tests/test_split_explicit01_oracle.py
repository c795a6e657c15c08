"""src/Ocean/SplitExplicit01 (OceanModel + Continuity3dModel + BarotropicModel,
SplitExplicitLSRK2nSolver with fast-step averaging) restated by the oracle: structural checks on
a small box.  The reference's numbers for this path -- the StateCheck table of
test/Ocean/SplitExplicit/simple_box_2dt.jl after five days on 20^3 elements
(tests/golden/ocean_simple_box_2dt_refvals.json) -- take the oracle well over an hour on eight
cores; they are reproduced on the device (tests/test_gpu_split_explicit01.py), which this file's
oracle agrees with step by step."""
import json
import os

import numpy as np

from helpers import cm, simple_box_2dt_fields, simple_box_2dt_setup

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ocean_simple_box_2dt_refvals.json")))


def oracle_pair(O, model, g3, baro, g2):
    o3 = O.OracleDGModel(model, g3)
    oc = O.OracleDGModel(cm.ocean01.Continuity3dModel01(model), g3)
    vf, ef = cm.ocean01.default_filters(g3)
    O.ocean01_hooks(o3, oc, vf, ef)
    o2 = O.OracleDGModel(baro, g2)
    return o3, o2


def test_fixture_has_the_reference_rows():
    assert len(GOLD["varr"]) == 28 and len(GOLD["parr"]) == 28
    names = [(r[0], r[1]) for r in GOLD["varr"]]
    O1 = cm.ocean01
    assert names[:4] == [("oce Q_3D", n) for n in O1.STATE_NAMES_3D]
    assert names[4:12] == [("oce aux", n) for n in O1.AUX_NAMES_3D]
    assert names[12:15] == [("baro Q_2D", n) for n in O1.STATE_NAMES_2D]
    assert names[15:] == [("baro aux", n) for n in O1.AUX_NAMES_2D]


def test_two_slow_steps_keep_the_coupling_identities(oracle):
    model, g3, baro, g2 = simple_box_2dt_setup(3, 3, 3)
    o3, o2 = oracle_pair(oracle, model, g3, baro, g2)
    Q3 = model.init_state_prognostic(g3, o3.state_auxiliary, 0.0)
    Q2 = baro.init_state_prognostic(g2, o2.state_auxiliary, 0.0)
    se = oracle.SplitExplicit01Oracle(o3, o2, Q3, Q2, 5400.0, 240.0)
    for s in range(2):
        se.dostep(Q3, Q2, s * 5400.0)
    A3, A2 = o3.state_auxiliary, o2.state_auxiliary
    assert np.isfinite(Q3).all() and np.isfinite(Q2).all() and np.isfinite(A3).all()
    f = simple_box_2dt_fields(Q3, A3, Q2, A2, g2)
    H, nv, Nqh = model.problem.H, 3, 25
    # the 3-D eta is the averaged barotropic eta, the barotropic state restarts from the saved one
    v3 = lambda a: a.reshape(g3.nelem // nv, nv, a.shape[1], 5, Nqh)
    assert np.array_equal(v3(Q3)[:, :, 2], np.broadcast_to(f[("baro aux", "η_c")][:, None, None, :],
                                                          v3(Q3)[:, :, 2].shape))
    assert np.array_equal(f[("baro Q_2D", "η")], f[("baro aux", "η_s")])
    assert np.array_equal(f[("baro Q_2D", "U[1]")], f[("baro aux", "U_s[1]")])
    # dG_u = -G_U / H through the column (tendency_from_slow_to_fast!)
    for c in (0, 1):
        G = f[("baro aux", "Gᵁ[%d]" % (c + 1))]
        assert np.array_equal(v3(A3)[:, :, 5 + c], np.broadcast_to((-G / H)[:, None, None, :],
                                                                  v3(A3)[:, :, 5 + c].shape))
    # Delta_eta = eta_c - eta_diag; wz0 is w at the surface; pkin vanishes at the surface
    assert np.array_equal(f[("baro aux", "Δη")], f[("baro aux", "η_c")] - f[("baro aux", "η_diag")])
    assert np.array_equal(v3(A3)[:, :, 2], np.broadcast_to(v3(A3)[:, -1, 0, -1, :][:, None, None, :],
                                                          v3(A3)[:, :, 2].shape))
    assert np.abs(v3(A3)[:, -1, 1, -1, :]).max() == 0.0
    # the flow deviation has no vertical mean: int u_d dz = 0
    top = o3.integrate_velocity(A3[:, 3:5, :].copy())
    assert np.abs(top).max() <= 1e-13 * H * np.abs(A3[:, 3:5, :]).max()
    # the wind spins the box up: a few cm/s after three hours
    assert 1e-3 < np.abs(Q3[:, 0:2]).max() < 1.0
