"""Ghost exchange on the device.

1. The reference's known-answer test (test/Arrays/mpi_comm.jl:23-153: three ranks, Np = 9, two
   states, explicit vmaps and expected ghost payloads) on the HIP kernels k_fillsendbuf /
   k_transferrecvbuf through the C ABI (cmdg_fillsendbuf / cmdg_transferrecvbuf).
2. The RCCL transport of the handle-based exchange -- receive offsets, per-neighbour ranges,
   several neighbours in one group -- exercised on ONE GPU: a rank of a multi-rank partition
   whose neighbours are all mapped to itself sends every range to itself, so ghost node
   vmaprecv[i] must end up holding the rank's own node vmapsend[i]; a whole right-hand side
   through that transport must equal the one through the in-process (device copy) transport.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mpi_comm_fixture.json")))


def test_reference_ghost_exchange_fixture_on_hip_kernels(cm, torch):
    L = cm._lib.lib()
    Np, ns, ranks = FX["Np"], FX["nstate"], FX["ranks"]
    dev = "cuda:0"
    arrays, sends = [], []
    for r, fx in enumerate(ranks):
        ne = fx["numreal"] + fx["numghost"]
        Q = np.full((ne, ns, Np), -1.0)
        vals = (r * 1000 + np.arange(1, Np * fx["numreal"] + 1)).reshape(fx["numreal"], Np)
        Q[:fx["numreal"], 0, :] = vals
        Q[:fx["numreal"], 1, :] = vals + FX["shift"]
        Qd = torch.from_numpy(Q).to(dev)
        vs = torch.tensor(fx["vmapsend"], dtype=torch.int64, device=dev)
        send = torch.zeros((len(fx["vmapsend"]), ns), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        cm._lib.check(L.cmdg_fillsendbuf(send.data_ptr(), Qd.data_ptr(), vs.data_ptr(),
                                         len(fx["vmapsend"]), Np, ns))
        arrays.append(Qd)
        sends.append(send.cpu().numpy())
    for r, fx in enumerate(ranks):
        recv = np.zeros((len(fx["vmaprecv"]), ns))
        for n, nbr in enumerate(fx["nabrtorank"]):
            a, b = fx["nabrtovmaprecv"][n]
            k = ranks[nbr]["nabrtorank"].index(r)
            sa, sb = ranks[nbr]["nabrtovmapsend"][k]
            assert sb - sa == b - a
            recv[a - 1:b] = sends[nbr][sa - 1:sb]
        rd = torch.from_numpy(recv).to(dev)
        vr = torch.tensor(fx["vmaprecv"], dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        cm._lib.check(L.cmdg_transferrecvbuf(arrays[r].data_ptr(), rd.data_ptr(), vr.data_ptr(),
                                             len(fx["vmaprecv"]), Np, ns))
    for r, fx in enumerate(ranks):
        Q = arrays[r].cpu().numpy()
        flat0, flat1 = Q[:, 0, :].reshape(-1), Q[:, 1, :].reshape(-1)
        idx = np.asarray(fx["vmaprecv"]) - 1
        exp = np.asarray(fx["expectedghostdata"], dtype=np.float64)
        assert np.array_equal(flat0[idx], exp), r
        assert np.array_equal(flat1[idx], exp + FX["shift"]), r
        ghost = np.ones_like(flat0, dtype=bool)
        ghost[:fx["numreal"] * Np] = False
        ghost[idx] = False
        assert (flat0[ghost] == -1).all() and (flat1[ghost] == -1).all()


def _self_neighbour_grid(cm, rank, size, periodic=True):
    """Rank ``rank`` of a ``size``-rank periodic brick whose neighbour table names the rank itself."""
    M = cm.mesh
    rng = [np.linspace(-1, 1, 7), np.linspace(-1, 1, 4), np.linspace(-1, 1, 3)]
    topl = M.StackedBrickTopology(rng, periodicity=(True, True, False), boundary=((0, 0), (0, 0), (1, 2)),
                                  connectivity="face", rank=rank, size=size)
    grid = M.DiscontinuousSpectralElementGrid(topl, 4)
    return grid


@pytest.mark.parametrize("rank,size", [(0, 2), (1, 3)])
def test_rccl_halo_with_itself_as_every_neighbour(cm, torch, rank, size):
    from helpers import pseudo1d_setup
    grid = _self_neighbour_grid(cm, rank, size)
    nn = len(grid.nabrtorank)
    assert nn >= (1 if size == 2 else 2)
    send = np.asarray(grid.nabrtovmapsend).reshape(nn, 2)      # rows (first, last), 1-based
    recv = np.asarray(grid.nabrtovmaprecv).reshape(nn, 2)
    for n in range(nn):          # a range sent to oneself must fit the range received
        assert send[n][1] - send[n][0] == recv[n][1] - recv[n][0]
    grid.nabrtorank = [0] * nn                              # every neighbour is this process
    law = pseudo1d_setup()[0]
    dg = cm.dgmodel.DGModel(law, grid, direction=0)
    dg.comm_init_rccl(cm.dgmodel.rccl_unique_id(), 0, 1)
    dg.comm_selftest()
    Q = dg.init_ode_state(0.0)
    nr, Np = grid.nreal, grid.Np
    Q[nr:] = float("nan")
    for _ in range(3):                                      # slots are reused across exchanges
        dg.halo_begin(Q)
        dg.halo_end(Q)
        dg.synchronize()
    q = Q.cpu().numpy()[:, 0, :].reshape(-1)
    vs, vr = np.asarray(grid.vmapsend) - 1, np.asarray(grid.vmaprecv) - 1
    for n in range(nn):
        a, b = recv[n][0] - 1, recv[n][1]
        sa, sb = send[n][0] - 1, send[n][1]
        assert np.array_equal(q[vr[a:b]], q[vs[sa:sb]]), n
    ghost = np.ones(q.size, dtype=bool)
    ghost[:nr * Np] = False
    ghost[vr] = False
    assert np.isnan(q[ghost]).all()                         # nothing else was written
    dg.close()


def test_rhs_through_rccl_equals_rhs_through_device_copies(cm, torch):
    """The five-exchange evaluation (hyperdiffusion) of rank 0 of a 2-rank partition talking to
    itself: RCCL transport == local transport, bit for bit -- same kernels, same payloads, only
    the transport differs."""
    from helpers import periodic_hyperdiffusion_setup
    grid = _self_neighbour_grid(cm, 0, 2)
    grid.nabrtorank = [0] * len(grid.nabrtorank)
    out = []
    for transport in ("rccl", "local"):
        law = periodic_hyperdiffusion_setup()[0]
        dg = cm.dgmodel.DGModel(law, grid, numerical_flux_first_order=1, direction=0)
        if transport == "rccl":
            dg.comm_init_rccl(cm.dgmodel.rccl_unique_id(), 0, 1)
        else:
            cm.dgmodel.connect_local([dg])
        Q = dg.init_ode_state(0.0)
        T = dg.create_state()
        if transport == "rccl":
            dg(T, Q, 0.0, 1.0, 0.0)
        else:
            cm.dgmodel.group_rhs([dg], [T], [Q], 0.0, 1.0, 0.0)
        dg.synchronize()
        out.append((T.cpu().numpy()[:grid.nreal], Q.cpu().numpy()))
        dg.close()
    assert np.abs(out[0][0]).max() > 0
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1], out[1][1], equal_nan=True)
