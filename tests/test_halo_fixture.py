"""The reference's ghost-exchange known-answer test (test/Arrays/mpi_comm.jl: three ranks, nine
nodes per element, two states) replayed through the oracle's restatement of
kernel_fillsendbuf! / kernel_transferrecvbuf! (MPIStateArrays.jl:837-871) with the three ranks
held in one process: buffers are (nstate, nvmap) with the state index fastest, `(e, n) =
fldmod1(vmap, Np)`, neighbours own contiguous ranges of the send / receive maps."""
import ctypes as C
import json
import os

import numpy as np

FX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mpi_comm_fixture.json")))


def test_reference_ghost_exchange_fixture(oracle):
    L, Np, ns = oracle.lib(), FX["Np"], FX["nstate"]
    ranks = FX["ranks"]
    arrays, sends = [], []
    for r, fx in enumerate(ranks):
        ne = fx["numreal"] + fx["numghost"]
        Q = np.full((ne, ns, Np), -1.0)                      # numpy view of (Np, nstate, nelem)
        vals = (r * 1000 + np.arange(1, Np * fx["numreal"] + 1)).reshape(fx["numreal"], Np)
        Q[:fx["numreal"], 0, :] = vals
        Q[:fx["numreal"], 1, :] = vals + FX["shift"]
        vs = np.asarray(fx["vmapsend"], dtype=np.int64)
        send = np.zeros((len(vs), ns))
        L.orc_fillsendbuf(oracle._p(send), oracle._p(Q), oracle._p(vs), C.c_int64(len(vs)), Np, ns)
        arrays.append(Q)
        sends.append(send)
    for r, fx in enumerate(ranks):
        vr = np.asarray(fx["vmaprecv"], dtype=np.int64)
        recv = np.zeros((len(vr), ns))
        for n, nbr in enumerate(fx["nabrtorank"]):
            a, b = fx["nabrtovmaprecv"][n]
            # the neighbour's send range that targets this rank
            k = ranks[nbr]["nabrtorank"].index(r)
            sa, sb = ranks[nbr]["nabrtovmapsend"][k]
            # (the fixture's receive ranges may be shorter than what the neighbour sends: rank 1
            # sends 13 nodes of which rank 0 lists 13, rank 2 sends 5 for 5)
            assert sb - sa == b - a
            recv[a - 1:b] = sends[nbr][sa - 1:sb]
        L.orc_transferrecvbuf(oracle._p(arrays[r]), oracle._p(recv), oracle._p(vr),
                              C.c_int64(len(vr)), Np, ns)
    for r, fx in enumerate(ranks):
        Q = arrays[r]
        flat0 = Q[:, 0, :].reshape(-1)          # Q[:, 1, :][:] in the reference's layout
        flat1 = Q[:, 1, :].reshape(-1)
        idx = np.asarray(fx["vmaprecv"]) - 1
        exp = np.asarray(fx["expectedghostdata"], dtype=np.float64)
        assert np.array_equal(flat0[idx], exp), r
        assert np.array_equal(flat1[idx], exp + FX["shift"]), r
        # nothing but the listed ghost nodes was written
        ghost = np.ones_like(flat0, dtype=bool)
        ghost[:fx["numreal"] * Np] = False
        untouched = ghost.copy()
        untouched[idx] = False
        assert (flat0[untouched] == -1).all()
