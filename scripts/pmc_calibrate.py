"""Known-byte-count launches in this library's access pattern (8 B per lane, coalesced
columns) for calibrating FETCH_SIZE / WRITE_SIZE: the exponential filter on the atmos target
reads 7 and writes 5 fields of every real node; the reverse column integral reads 2 and
writes 2.  Run under rocprofv3 --pmc (scripts/pmc_any.sh)."""
import sys

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
filt = F.make_device_filter(dg, F.ExponentialFilter(grid, 0, 20), F.AtmosFilterPerturbations(law))
aux = torch.zeros((grid.nelem, 4, grid.Np), dtype=torch.float64, device=Q.device)
for _ in range(5):
    filt.apply(Q)
    dg.reverse_indefinite_stack_integral(aux, [0, 1], [2, 3])
dg.synchronize()
nodes = grid.nreal * grid.Np
print("nodes", nodes, "filter read B", 56 * nodes, "write B", 40 * nodes,
      "reverse integral read B", 16 * nodes, "write B", 16 * nodes)
