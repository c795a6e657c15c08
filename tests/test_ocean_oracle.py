"""Hydrostatic Boussinesq ocean model in the oracle against the reference's own regression
values: test/Ocean/HydrostaticBoussinesq/test_3D_spindown.jl (one simulated day, 720 LSRK144
steps of 120 s on 5 x 5 x 8 elements, N = 4) with the StateCheck table
test/Ocean/refvals/3D_hydrostatic_spindown_refvals.jl (min / max / std to 12 digits) and the
analytic-solution error the reference prints (0.0011289879366523504).  This run composes the
right-hand side, the element filters, the column integrals and LSRK144.  CPU only (~1 min)."""
import json
import os

import numpy as np
import pytest

from cmdg_loader import cm
from helpers import ocean_gyre_setup, ocean_spindown_setup, ocean_windstress_setup

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ocean_spindown_refvals.json")))
GYRE = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ocean_gyre_short_refvals.json")))
F = cm.mesh.filters


def statecheck(a):
    """``scstats`` (src/Diagnostics/Debug/StateCheck.jl:231-283): unweighted statistics of the
    real data."""
    a = np.asarray(a).reshape(-1)
    m = a.mean()
    return a.min(), a.max(), m, np.sqrt(((a - m) ** 2).sum() / (a.size - 1))


def check_against_refvals(Q, aux, rtol):
    fields = {("state", "u[1]"): Q[:, 0], ("state", "u[2]"): Q[:, 1], ("state", "η"): Q[:, 2],
              ("state", "θ"): Q[:, 3], ("aux", "y"): aux[:, 0], ("aux", "w"): aux[:, 1],
              ("aux", "pkin"): aux[:, 2], ("aux", "wz0"): aux[:, 3]}
    for row in GOLD["explicit"]:
        lab, name, rmin, rmax, _rmean, rstd = row
        if (lab, name) not in fields:
            continue
        prec = GOLD["precision"][name]
        mn, mx, _, sd = statecheck(fields[(lab, name)])
        for got, ref, p in ((mn, rmin, prec[0]), (mx, rmax, prec[1]), (sd, rstd, prec[3])):
            if p == 0:
                continue
            if ref == 0:
                assert got == 0, (name, got)
            else:
                assert abs(got - ref) <= rtol * abs(ref), (name, got, ref)


def test_spindown_matches_reference_statecheck(oracle):
    law, grid = ocean_spindown_setup()
    dg = oracle.OracleDGModel(law, grid)
    oracle.hydrostatic_boussinesq_hooks(dg, F.CutoffFilter(grid, grid.N[-1] - 1),
                                        F.ExponentialFilter(grid, 1, 8))
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    RKA, RKB, RKC = cm.odesolvers.LSRK144_COEFFICIENTS
    dQ = np.zeros_like(Q)
    dt, t = 120.0, 0.0
    for _ in range(720):
        oracle.lsrk_step(dg, Q, dQ, t, dt, RKA, RKB, RKC)
        t += dt
    Qe = law.init_state_prognostic(grid, dg.state_auxiliary, 86400.0)
    err = np.sqrt(oracle.weighted_norm2_local(grid, Q, Qe) / oracle.weighted_norm2_local(grid, Qe))
    assert err < 0.005                                              # test_3D_spindown.jl:133
    assert abs(err - GOLD["error_printed_by_reference"]) < 1e-10    # :149 (comment)
    # 12 digits in the reference's own run; the filter matrices here differ from Julia's in the
    # last bit and are applied 10 080 times, which is where the observed 1e-12 .. 3e-12 comes from
    check_against_refvals(Q, dg.state_auxiliary, rtol=5e-12)
    # v stays at rounding level, theta and pkin identically zero (alpha_T = 0)
    assert np.abs(Q[:, 1]).max() < 1e-12 and not Q[:, 3].any() and not dg.state_auxiliary[:, 2].any()


WIND = json.load(open(os.path.join(os.path.dirname(__file__), "golden",
                                   "ocean_windstress_short_refvals.json")))


def check_gyre_refvals(Q, aux, rtol=2e-12, table=None):
    """all four statistics of the seven live fields against the `short` table; values that are
    small by cancellation (means, theta at the sea floor) are compared on the field's scale."""
    fields = {"u[1]": Q[:, 0], "u[2]": Q[:, 1], "η": Q[:, 2], "θ": Q[:, 3], "y": aux[:, 0],
              "w": aux[:, 1], "pkin": aux[:, 2], "wz0": aux[:, 3]}
    n = 0
    for lab, name, rmin, rmax, rmean, rstd in (table or GYRE["short"]):
        if name not in fields:
            continue
        st = statecheck(fields[name])
        scale = max(abs(rmin), abs(rmax))
        for k, (got, ref) in enumerate(zip(st, (rmin, rmax, rmean, rstd))):
            floor = scale if k == 2 else 0.05 * scale     # the mean is a cancelling sum
            assert abs(got - ref) <= rtol * max(abs(ref), floor), (name, got, ref)
            n += 1
    return n


def test_ocean_gyre_short_matches_reference_statecheck(oracle):
    """test_ocean_gyre_short.jl: wind stress + temperature flux at the surface, no-slip walls,
    beta-plane Coriolis force, stratified theta with convective adjustment; 30 LSRK144 steps."""
    law, grid = ocean_gyre_setup()
    dg = oracle.OracleDGModel(law, grid)
    oracle.hydrostatic_boussinesq_hooks(dg, F.CutoffFilter(grid, 3), F.ExponentialFilter(grid, 1, 8))
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    RKA, RKB, RKC = cm.odesolvers.LSRK144_COEFFICIENTS
    dQ = np.zeros_like(Q)
    t = 0.0
    for _ in range(30):
        oracle.lsrk_step(dg, Q, dQ, t, 120.0, RKA, RKB, RKC)
        t += 120.0
    assert check_gyre_refvals(Q, dg.state_auxiliary) == 32


def test_windstress_short_matches_reference_statecheck(oracle):
    """explicit run of test_windstress_short.jl: 20 LSRK144 steps of 180 s."""
    law, grid = ocean_windstress_setup()
    dg = oracle.OracleDGModel(law, grid)
    oracle.hydrostatic_boussinesq_hooks(dg, F.CutoffFilter(grid, 3), F.ExponentialFilter(grid, 1, 8))
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    RKA, RKB, RKC = cm.odesolvers.LSRK144_COEFFICIENTS
    dQ = np.zeros_like(Q)
    t = 0.0
    for _ in range(20):
        oracle.lsrk_step(dg, Q, dQ, t, 180.0, RKA, RKB, RKC)
        t += 180.0
    table = [r for r in WIND["explicit_cpu"] if r[1] != "θ"]
    # observed: 1e-13 except wz0 (surface w, small by cancellation: 7e-12); the reference's own
    # CPU and GPU tables differ by 1e-11 in two entries (test_windstress_refvals.jl:47-49)
    assert check_gyre_refvals(Q, dg.state_auxiliary, table=table, rtol=2e-11) == 28
    # theta stays 20 to rounding (the reference's std of theta is 2.6e-13 and unchecked)
    assert np.abs(Q[:, 3] - 20.0).max() < 1e-11
