"""Per-launch timing of the operators around the RHS (filters, Courant number, column
integrals) on the Held-Suarez grid: HIP events through cmdg_profile_*, algorithmic bytes per
node stated next to each.  Usage: python scripts/measure_ops.py [n_horz] > profiles/...json"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from cmdg_loader import cm            # noqa: E402
from helpers import held_suarez_setup  # noqa: E402

n_horz = int(sys.argv[1]) if len(sys.argv) > 1 else 30
law, grid, d, dd = held_suarez_setup(n_horz=n_horz, n_vert=8)
dg = cm.dgmodel.DGModel(law, grid, direction=d, diffusion_direction=dd)
Q = dg.init_ode_state(0.0)
F = cm.mesh.filters
nodes = grid.nreal * grid.Np
out = {"workload": "Held-Suarez grid 6x%dx%dx8, N=4, %d elements" % (n_horz, n_horz, grid.nreal)}


def timed(name, fn, bytes_per_node, reps=20):
    fn()
    dg.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    dg.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out[name] = {"ms": 1e3 * dt, "algorithmic_bytes_per_node": bytes_per_node,
                 "GBs": bytes_per_node * nodes / dt / 1e9}


filt = F.make_device_filter(dg, F.ExponentialFilter(grid, 0, 20), F.AtmosFilterPerturbations(law))
timed("exponential_filter_atmos_perturbations", lambda: filt.apply(Q), 8 * 12)
cut = F.make_device_filter(dg, F.CutoffFilter(grid, 3), F.FilterIndices(range(1, 6)))
timed("cutoff_filter_5_states", lambda: cut.apply(Q), 8 * 10)
tm = F.make_device_filter(dg, F.TMARFilter(), F.FilterIndices(1))
Qp = Q.clone()
timed("tmar_filter_1_state", lambda: tm.apply(Qp), 8 * 3)
# Courant: Q (5) + aux (Phi grad 3, Phi, ...) + coordinates (3); host sync included
timed("nondiffusive_courant_incl_host_sync", lambda: dg.courant(1, Q, 1.0), 8 * (5 + 4 + 3), reps=10)
timed("min_node_distance_incl_host_sync", lambda: dg.min_node_distance(), 8 * 3, reps=10)
aux = torch.zeros((grid.nelem, 4, grid.Np), dtype=torch.float64, device=Q.device)
timed("stack_integral_2_fields", lambda: dg.indefinite_stack_integral(
    Q, aux, [(1, 0), (1, 4)], [0, 1]), 8 * (2 + 1 + 2))
timed("reverse_stack_integral_2_fields", lambda: dg.reverse_indefinite_stack_integral(
    aux, [0, 1], [2, 3]), 8 * 4)
print(json.dumps(out, indent=1))
