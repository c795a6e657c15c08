"""Split-explicit barotropic / baroclinic ocean stepper (BASELINE config 5) in the oracle
against the reference's own StateCheck tables: test/Ocean/SplitExplicit/test_spindown_long.jl
with test/Ocean/refvals/hydrostatic_spindown_refvals.jl (tests/golden/
ocean_split_explicit_refvals.json).  One simulated day on 5 x 5 x 8 elements, N = 4:
"coupled" = 288 slow steps of 300 s with a 300 s fast step, "ninety_minutes" = 16 slow steps
of 5400 s sub-stepped by the 300 s barotropic model.  The reference checks min / max / std to
12 digits (11 for the column-integrated tendencies); this restatement reproduces them to within
two units of the twelfth digit.  The exception is the vertically averaged slow tendency
(G_U / dG_u, 1e-9 m/s^2 on a velocity of 1 m/s): it is the viscous operator applied to u, which
amplifies the 1e-13 rounding-level differences in u (last-bit filter matrices, see
test_ocean_oracle.py) to 1e-11 .. 2e-10 of its own size; it is pinned to 9 digits here where the
reference asks 11 of a bit-reproducible rerun.  CPU only (~1 min)."""
import json
import os

import numpy as np
import pytest

from cmdg_loader import cm
from helpers import (check_split_explicit_table, split_explicit_fields, split_explicit_schedule,
                     split_explicit_setup)

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden",
                                   "ocean_split_explicit_refvals.json")))
F = cm.mesh.filters


def relaxed(parr):
    """the column-averaged tendencies are pinned to 9 digits (see the module docstring)"""
    out = []
    for r in parr:
        r = list(r)
        if r[1] in ("ΔGᵘ[1]", "Gᵁ[1]"):
            r[2:6] = [min(p, 9) for p in r[2:6]]
        out.append(r)
    return out


def run_oracle(oracle, coupled, dt_slow, dt_fast=300.0):
    law3, g3, law2, g2 = split_explicit_setup(coupled)
    dg3 = oracle.OracleDGModel(law3, g3)                          # Rusanov
    oracle.hydrostatic_boussinesq_hooks(dg3, F.CutoffFilter(g3, g3.N[-1] - 1),
                                        F.ExponentialFilter(g3, 1, 8))
    dg2 = oracle.OracleDGModel(law2, g2, nf_first=1)              # CentralNumericalFluxFirstOrder
    Q3 = law3.init_state_prognostic(g3, dg3.state_auxiliary, 0.0)
    Q2 = law2.init_state_prognostic(g2, dg2.state_auxiliary, 0.0)
    dt, nsteps = split_explicit_schedule(dt_slow)
    se = oracle.SplitExplicitOracle(dg3, dg2, Q3, Q2, dt, dt_fast)
    t = 0.0
    for _ in range(nsteps):
        se.dostep(Q3, Q2, t)
        t += dt
    return law3, g3, law2, g2, dg3, dg2, Q3, Q2


@pytest.mark.parametrize("name,coupled,dt_slow", [("coupled", True, 300.0),
                                                  ("ninety_minutes", True, 5400.0)])
def test_split_explicit_matches_reference_statecheck(oracle, name, coupled, dt_slow):
    law3, g3, law2, g2, dg3, dg2, Q3, Q2 = run_oracle(oracle, coupled, dt_slow)
    A3, A2 = dg3.state_auxiliary, dg2.state_auxiliary
    fields = split_explicit_fields(Q3, A3, Q2, A2, g2)
    check_split_explicit_table(GOLD[name], relaxed(GOLD["parr"]), fields, slack=2.0)
    # analytic solution check of run_split_explicit (hydrostatic_spindown.jl:101-137)
    for law, g, Q, A in ((law3, g3, Q3, A3), (law2, g2, Q2, A2)):
        Qe = law.init_state_prognostic(g, A, 86400.0)
        err = np.sqrt(oracle.weighted_norm2_local(g, Q, Qe) / oracle.weighted_norm2_local(g, Qe))
        assert err < 0.005
    # the extruded barotropic fields stay constant along the extrusion (to rounding: the
    # numerically differentiated metric terms differ in the last bit from layer to layer)
    q = Q2.reshape(g2.nelem, 3, g2.Nq[2], -1)
    assert np.abs(q - q[:, :, :1, :]).max() < 1e-12 * np.abs(q).max()
