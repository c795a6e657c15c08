"""The N > 1 path on CPU: real processes (torch.distributed, gloo) each build their rank's view
of the Hilbert-partitioned mesh, evaluate the oracle RHS with the ghost exchange carried by
isend/irecv of the packed face-node buffers (MPIStateArrays.jl:411-514), and must reproduce the
single-rank result bit for bit: two ranks on config 1's advection-diffusion brick (two
exchanges per evaluation), three ranks on the Held-Suarez cubed sphere (the reference's five
exchange points of a hyperdiffusive law, DGModel.jl:126-412, with orientation-3 faces of the
cube edges crossing ranks)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(name, rank=0, size=1):
    """(law, grid, direction, diffusion_direction, dt, nsteps)"""
    from helpers import held_suarez_setup, pseudo1d_setup
    if name == "advdiff":
        law, grid, dt = pseudo1d_setup(direction=0, rank=rank, size=size)
        return law, grid, 0, 0, dt, 2
    law, grid, d, dd = held_suarez_setup(3, 2, rank=rank, size=size)
    return law, grid, d, dd, 2.0, 2


def _worker(rank, size, port, outdir, case):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    O.set_num_threads(1)
    law, grid, direction, diffdir, dt, nsteps = _case(case, rank, size)
    nexchanges = [0]

    class GlooExchange:
        """begin/end_ghost_exchange! with pack/unpack done by the oracle's restatement of
        kernel_fillsendbuf!/kernel_transferrecvbuf!."""

        def begin(self, arr, nvar):
            import ctypes as C
            nexchanges[0] += 1
            L = O.lib()
            send = np.zeros((len(grid.vmapsend), nvar))
            recv = np.zeros((len(grid.vmaprecv), nvar))
            vs = np.ascontiguousarray(grid.vmapsend, dtype=np.int64)
            L.orc_fillsendbuf(O._p(send), O._p(arr), O._p(vs), C.c_int64(len(vs)), grid.Np, nvar)
            reqs = []
            ts, tr = torch.from_numpy(send), torch.from_numpy(recv)
            for n, nbr in enumerate(grid.nabrtorank):
                a, b = grid.nabrtovmaprecv[n]
                reqs.append(dist.irecv(tr[a - 1:b], src=nbr))
            for n, nbr in enumerate(grid.nabrtorank):
                a, b = grid.nabrtovmapsend[n]
                reqs.append(dist.isend(ts[a - 1:b].contiguous(), dst=nbr))
            return reqs, recv, (ts, tr)

        def end(self, arr, nvar, token):
            import ctypes as C
            reqs, recv, _keep = token
            for r in reqs:
                r.wait()
            vr = np.ascontiguousarray(grid.vmaprecv, dtype=np.int64)
            O.lib().orc_transferrecvbuf(O._p(arr), O._p(recv), O._p(vr), C.c_int64(len(vr)),
                                        grid.Np, nvar)

    dg = O.OracleDGModel(law, grid, nf_first=0, direction=direction, diffusion_direction=diffdir,
                         exchange=GlooExchange())
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    Q[grid.nreal:] = np.nan
    dQ = np.zeros_like(Q)
    for s in range(nsteps):
        O.lsrk54_step(dg, Q, dQ, s * dt, dt)
    np.savez(os.path.join(outdir, "r%d.npz" % rank), Q=Q[:grid.nreal],
             gl=grid.topology.globalelems[:grid.nreal], nex=nexchanges[0],
             orient3=int((np.asarray(grid.topology.elemtoordr)[:grid.nreal] == 3).sum()),
             nghost=grid.nelem - grid.nreal)
    dist.barrier()
    dist.destroy_process_group()


def _run(tmp_path, case, size):
    import torch.multiprocessing as mp
    from oracle import oracle as O
    O.build()
    port = 29500 + (os.getpid() % 2000) + (7 if case == "advdiff" else 13)
    mp.spawn(_worker, args=(size, port, str(tmp_path), case), nprocs=size, join=True)
    law, grid, direction, diffdir, dt, nsteps = _case(case)
    dg = O.OracleDGModel(law, grid, nf_first=0, direction=direction, diffusion_direction=diffdir)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    dQ = np.zeros_like(Q)
    for s in range(nsteps):
        O.lsrk54_step(dg, Q, dQ, s * dt, dt)
    ref = {int(g): Q[i] for i, g in enumerate(grid.topology.globalelems[:grid.nreal])}
    seen, out = 0, []
    for r in range(size):
        z = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        for q, g in zip(z["Q"], z["gl"]):
            assert np.array_equal(q, ref[int(g)])      # same kernels, same order: bit exact
            seen += 1
        out.append(z)
    assert seen == grid.nreal
    return out, nsteps


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_rank(tmp_path):
    ranks, nsteps = _run(tmp_path, "advdiff", 2)
    assert all(int(z["nex"]) == 2 * 5 * nsteps for z in ranks)      # Q and the gradient flux


@pytest.mark.timeout(600)
def test_three_rank_gloo_held_suarez_matches_single_rank(tmp_path):
    """The headline law partitioned over three processes: five exchange points per evaluation
    (Q, gradient flux, hyperdiffusion gradients, Laplacians, gradients of Laplacians -- the
    reference's, the oracle keeps its order), every rank with ghosts, faces of orientation 3 in
    the partition."""
    ranks, nsteps = _run(tmp_path, "heldsuarez", 3)
    assert all(int(z["nex"]) == 5 * 5 * nsteps for z in ranks)
    assert all(int(z["nghost"]) > 0 for z in ranks)
    assert sum(int(z["orient3"]) for z in ranks) > 0
