"""Split-explicit ocean box (BASELINE configs[4] shape on one GPU): wall time per slow step,
per-kernel HIP-event breakdown of the slow (3-D) and fast (barotropic) handles, and the oracle
timed on the host on a bounded sample.
Usage: python scripts/measure_split_explicit.py [Nx] [Nz] [dt_slow] > profiles/...json"""
import json
import sys
import time

import numpy as np
import torch

import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cmdg_loader import cm                   # noqa: E402
from helpers import split_explicit_setup     # noqa: E402

Nx = int(sys.argv[1]) if len(sys.argv) > 1 else 48
Nz = int(sys.argv[2]) if len(sys.argv) > 2 else 16
# the reference's 5 x 5 x 8 runs use dt_slow = 5400 s over dt_fast = 300 s; both are scaled with
# the horizontal element size so that the finer box stays inside the barotropic CFL limit
dt_slow = (float(sys.argv[3]) if len(sys.argv) > 3 else 5400.0) * 5.0 / Nx
dt_fast = 300.0 * 5.0 / Nx
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
cpu = "--no-cpu" not in sys.argv
# polynomial order of the barotropic grid along its extrusion (--extrusion=1: two nodes)
ext = [int(a.split("=")[1]) for a in sys.argv if a.startswith("--extrusion=")]
N_ext = ext[0] if ext else None
O = cm.ocean
t0 = time.time()
law3, g3, law2, g2 = split_explicit_setup(True, Nx=Nx, Ny=Nx, Nz=Nz, N_extrusion=N_ext)
print("grids: %.1f s" % (time.time() - t0), file=sys.stderr, flush=True)
dg3 = cm.dgmodel.DGModel(law3, g3)
keep = O.install_hydrostatic_boussinesq_hooks(dg3)
dg2 = cm.dgmodel.DGModel(law2, g2, numerical_flux_first_order=cm.balancelaws.CentralNumericalFluxFirstOrder)
Q3, Q2 = dg3.init_ode_state(0.0), dg2.init_ode_state(0.0)
se = O.SplitExplicitSolver(dg3, dg2, Q3, Q2, dt_slow, dt_fast)
se.dostep(Q3, Q2, 2)
torch.cuda.synchronize()
t0 = time.perf_counter()
se.dostep(Q3, Q2, steps)
el = time.perf_counter() - t0
for d in (dg3, dg2):
    d.profile_reset()
    d.profile_enable(True)
t0 = time.perf_counter()
se.dostep(Q3, Q2, steps)
el_ev = time.perf_counter() - t0
RKC = se.RKC
nsub = sum(int(np.ceil(((1 - RKC[s]) if s == 4 else (RKC[s + 1] - RKC[s])) * dt_slow / dt_fast))
           for s in range(5))
out = {"workload": "split-explicit ocean box %dx%dx%d elements, N=4 (%d 3-D elements, %d columns), "
                   "dt_slow=%g s, dt_fast<=%g s (%d barotropic LSRK54 steps per slow step), Coupled; "
                   "barotropic grid %d nodes per element"
                   % (Nx, Nx, Nz, g3.nreal, g2.nreal, dt_slow, dt_fast, nsub, g2.Np),
       "ms_per_slow_step": 1e3 * el / steps, "ms_per_slow_step_with_events": 1e3 * el_ev / steps,
       "node_updates_per_s_3d": g3.nreal * g3.Np * 5 * steps / el,
       "state_finite": bool(torch.isfinite(Q3).all().item() and torch.isfinite(Q2).all().item())}
for name, d in (("slow", dg3), ("fast", dg2)):
    k = {}
    for kn in ("GRADIENTS", "TENDENCY", "FILTER", "STACK_INTEGRAL"):
        ms, n = d.profile_get(kn)
        if n:
            k[kn] = {"avg_ms": ms / n, "launches_per_slow_step": n / steps,
                     "ms_per_slow_step": ms / steps}
    out["kernels_" + name] = k
if cpu:
    from oracle import oracle as OR
    F = cm.mesh.filters
    o3 = OR.OracleDGModel(law3, g3)
    OR.hydrostatic_boussinesq_hooks(o3, F.CutoffFilter(g3, 3), F.ExponentialFilter(g3, 1, 8))
    o2 = OR.OracleDGModel(law2, g2, nf_first=1)
    q3 = law3.init_state_prognostic(g3, o3.state_auxiliary, 0.0)
    q2 = law2.init_state_prognostic(g2, o2.state_auxiliary, 0.0)
    so = OR.SplitExplicitOracle(o3, o2, q3, q2, dt_slow, dt_fast)
    t0 = time.perf_counter()
    so.dostep(q3, q2, 0.0)
    c = time.perf_counter() - t0
    out["cpu_baseline"] = {"ms_per_slow_step": 1e3 * c, "cores": OR.get_max_threads(), "kind": "port",
                           "sample": "1 slow step of the same workload"}
    out["speedup_vs_oracle"] = c / (el / steps)
print(json.dumps(out, indent=1))
