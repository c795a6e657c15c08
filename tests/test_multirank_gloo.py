"""The N > 1 path on CPU: two real processes (torch.distributed, gloo) each build their
rank's view of the Hilbert-partitioned mesh, evaluate the oracle RHS with the ghost
exchange carried by isend/irecv of the packed face-node buffers
(MPIStateArrays.jl:411-514), and must reproduce the single-rank tendency."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, size, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from helpers import pseudo1d_setup
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=size)
    O.set_num_threads(1)
    law, grid, dt = pseudo1d_setup(direction=0, rank=rank, size=size)

    class GlooExchange:
        """begin/end_ghost_exchange! with pack/unpack done by the oracle's restatement of
        kernel_fillsendbuf!/kernel_transferrecvbuf!."""

        def begin(self, arr, nvar):
            import ctypes as C
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

    dg = O.OracleDGModel(law, grid, nf_first=0, direction=0, exchange=GlooExchange())
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    Q[grid.nreal:] = np.nan
    dQ = np.zeros_like(Q)
    for s in range(2):
        O.lsrk54_step(dg, Q, dQ, s * dt, dt)
    np.savez(os.path.join(outdir, "r%d.npz" % rank), Q=Q[:grid.nreal],
             gl=grid.topology.globalelems[:grid.nreal])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_rank(tmp_path):
    import torch.multiprocessing as mp
    from helpers import pseudo1d_setup
    from oracle import oracle as O
    O.build()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    law, grid, dt = pseudo1d_setup(direction=0)
    dg = O.OracleDGModel(law, grid, nf_first=0, direction=0)
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    dQ = np.zeros_like(Q)
    for s in range(2):
        O.lsrk54_step(dg, Q, dQ, s * dt, dt)
    ref = {int(g): Q[i] for i, g in enumerate(grid.topology.globalelems[:grid.nreal])}
    seen = 0
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        for q, g in zip(z["Q"], z["gl"]):
            assert np.array_equal(q, ref[int(g)])      # same kernels, same order: bit exact
            seen += 1
    assert seen == grid.nreal
