"""Diagnostic for the stream-priority ordering failure of the partitioned split-explicit ocean
(tests/test_gpu_split_explicit.py::test_partitioned_split_explicit_matches_single_rank[2]): runs the
2-rank local-transport pair against the one-rank run and prints, per rank and field, the largest
deviation and which elements carry it.  Environment: CMDG_HALO_PRIORITY, CMDG_DBG_* (engine.h).
usage: python scripts/probe/priority_order_diag.py [nsteps] [size]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from cmdg_loader import cm  # noqa: E402
from helpers import split_explicit_setup  # noqa: E402

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
size = int(sys.argv[2]) if len(sys.argv) > 2 else 2
O = cm.ocean
central = cm.balancelaws.CentralNumericalFluxFirstOrder
gpu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
law3, g3, law2, g2 = split_explicit_setup(True, Nx=4, Ny=3, Nz=3)
dg3 = cm.dgmodel.DGModel(law3, g3)
keep1 = O.install_hydrostatic_boussinesq_hooks(dg3)
dg2 = cm.dgmodel.DGModel(law2, g2, numerical_flux_first_order=central)
rng = np.random.default_rng(11)
Q3h = law3.init_state_prognostic(g3, dg3.state_auxiliary.cpu().numpy(), 600.0)
Q3h[:, 0:2] += 0.02 * rng.standard_normal(Q3h[:, 0:2].shape)
Q2h = law2.init_state_prognostic(g2, dg2.state_auxiliary.cpu().numpy(), 600.0)
by3 = {int(g): Q3h[i] for i, g in enumerate(g3.topology.globalelems[:g3.nreal])}
by2 = {int(g): Q2h[i] for i, g in enumerate(g2.topology.globalelems[:g2.nreal])}
Q3, Q2 = gpu(Q3h), gpu(Q2h)
se1 = O.SplitExplicitSolver(dg3, dg2, Q3, Q2, 1800.0, 300.0)
se1.dostep(Q3, Q2, nsteps)
ref3 = {int(g): Q3[i].cpu().numpy() for i, g in enumerate(g3.topology.globalelems[:g3.nreal])}
ref2 = {int(g): Q2[i].cpu().numpy() for i, g in enumerate(g2.topology.globalelems[:g2.nreal])}
if os.environ.get("DIAG_WARM_PRIO_STREAMS"):
    # use four high-priority streams once before libcmdg creates its own (the runtime then holds
    # their hardware queues already)
    n = int(os.environ["DIAG_WARM_PRIO_STREAMS"])
    ws = [torch.cuda.Stream(priority=-1) for _ in range(n)]
    x = torch.zeros(1 << 20, device="cuda:0")
    for w in ws:
        with torch.cuda.stream(w):
            x.add_(1.0)
    torch.cuda.synchronize()
    if os.environ.get("DIAG_WARM_KEEP") != "1":
        del ws
    print("warmed %d high-priority streams" % n, flush=True)
for trial in range(3):
    slows, fasts, Q3s, Q2s, grids, keeps = [], [], [], [], [], []
    for r in range(size):
        l3, gr3, l2, gr2 = split_explicit_setup(True, Nx=4, Ny=3, Nz=3, rank=r, size=size)
        d3 = cm.dgmodel.DGModel(l3, gr3)
        keeps.append(O.install_hydrostatic_boussinesq_hooks(d3))
        d2 = cm.dgmodel.DGModel(l2, gr2, numerical_flux_first_order=central)
        q3 = np.full((gr3.nelem, 4, gr3.Np), np.nan)
        for i, g in enumerate(gr3.topology.globalelems[:gr3.nreal]):
            q3[i] = by3[int(g)]
        q2 = np.full((gr2.nelem, 3, gr2.Np), np.nan)
        for i, g in enumerate(gr2.topology.globalelems[:gr2.nreal]):
            q2[i] = by2[int(g)]
        slows.append(d3), fasts.append(d2), grids.append((gr3, gr2))
        Q3s.append(gpu(q3)), Q2s.append(gpu(q2))
    cm.dgmodel.connect_local(slows)
    cm.dgmodel.connect_local(fasts)
    solvers = [O.SplitExplicitSolver(d3, d2, q3, q2, 1800.0, 300.0)
               for d3, d2, q3, q2 in zip(slows, fasts, Q3s, Q2s)]
    torch.cuda.synchronize()
    O.SplitExplicitSolver.group_dostep(solvers, Q3s, Q2s, nsteps)
    worst = 0.0
    for r, ((gr3, gr2), q3, q2) in enumerate(zip(grids, Q3s, Q2s)):
        q3n, q2n = q3.cpu().numpy(), q2.cpu().numpy()
        ext3 = set(int(e) - 1 for e in gr3.exteriorelems)
        ext2 = set(int(e) - 1 for e in gr2.exteriorelems)
        for name, qn, ref, gr, ext, cols in (("slow", q3n, ref3, gr3, ext3, (0, 1, 2, 3)),
                                             ("fast", q2n, ref2, gr2, ext2, (0, 1, 2))):
            for s in cols:
                errs = []
                for i, g in enumerate(gr.topology.globalelems[:gr.nreal]):
                    sc = max(np.abs(ref[int(g)][s]).max(), 1e-3)
                    errs.append(np.abs(qn[i, s] - ref[int(g)][s]).max() / sc)
                errs = np.array(errs)
                bad = np.nonzero(~(errs < 1e-11))[0]
                worst = max(worst, np.nanmax(errs) if np.isfinite(errs).any() else np.inf)
                if len(bad):
                    print("trial %d rank %d %s state %d: max %.3e, %d of %d elements off (%d of them exterior); nan: %d"
                          % (trial, r, name, s, np.nanmax(errs), len(bad), gr.nreal,
                             sum(1 for b in bad if b in ext), int(np.isnan(errs).sum())), flush=True)
    print("trial %d: worst %.3e %s" % (trial, worst, "OK" if worst < 1e-11 else "MISMATCH"), flush=True)
    for d3, k in zip(slows, keeps):
        d3.set_rhs_hooks()
        for f in k:
            f.close()
        d3.close()
    for d2 in fasts:
        d2.close()
