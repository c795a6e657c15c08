"""Host-side mesh mirror against the reference's own unit-test literals
(test/Numerics/Mesh/BrickMesh.jl, mpi_connect.jl, mpi_connectfull.jl, Elements.jl,
mpi_connect_sphere.jl property checks).  CPU only."""
import json
import os

import numpy as np
import pytest

from cmdg_loader import cm

M = cm.mesh
bm = M.brickmesh
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_linearpartition():  # BrickMesh.jl (test):8-13
    assert list(bm.linearpartition(1, 1, 1)) == [1]
    assert list(bm.linearpartition(20, 1, 1)) == list(range(1, 21))
    assert list(bm.linearpartition(10, 1, 2)) == [1, 2, 3, 4, 5]
    assert list(bm.linearpartition(10, 2, 2)) == [6, 7, 8, 9, 10]


def test_hilbertcode_small():  # BrickMesh.jl (test):15-37
    cases1 = {(0, 0): (0, 0), (0, 1): (0, 1), (1, 1): (1, 0), (1, 0): (1, 1)}
    for k, v in cases1.items():
        assert tuple(bm.hilbertcode(list(k), bits=1)) == v
    cases2 = {(0, 0): (0, 0), (1, 0): (0, 1), (1, 1): (0, 2), (0, 1): (0, 3), (0, 2): (1, 0),
              (0, 3): (1, 1), (1, 3): (1, 2), (1, 2): (1, 3), (2, 2): (2, 0), (2, 3): (2, 1),
              (3, 3): (2, 2), (3, 2): (2, 3), (3, 1): (3, 0), (2, 1): (3, 1), (2, 0): (3, 2),
              (3, 0): (3, 3)}
    for k, v in cases2.items():
        assert tuple(bm.hilbertcode(list(k), bits=2)) == v
    h = bm.hilbertcode(np.array([14, 3, 4], dtype=np.uint64))
    assert [int(x) for x in h] == [0x0, 0x0, 0xE25]


def test_centroidtocode():  # BrickMesh.jl (test):39-69
    vals = [[2, 3, 2, 3, 4, 4, 5, 5], [3, 4, 3, 4, 4, 4, 5, 5], [4, 5, 4, 5, 4, 4, 5, 5],
            [2, 3, 2, 3, 5, 5, 6, 6], [3, 4, 3, 4, 5, 5, 6, 6], [4, 5, 4, 5, 5, 5, 6, 6]]
    etc = np.zeros((6, 4, 2))
    for e, v in enumerate(vals):
        etc[e, :, 0], etc[e, :, 1] = v[:4], v[4:]
    code = bm.centroidtocode(etc)
    exp = [[0x0, 0x1555555555555555, 0xFFFFFFFFFFFFFFFF, 0x5555555555555555,
            0x6AAAAAAAAAAAAAAA, 0xAAAAAAAAAAAAAAAA],
           [0x0, 0x5555555555555555, 0xFFFFFFFFFFFFFFFF, 0x5555555555555555,
            0xAAAAAAAAAAAAAAAA, 0xAAAAAAAAAAAAAAAA]]
    assert [[int(x) for x in r] for r in code] == exp
    one = bm.centroidtocode(etc[:1])
    assert not one.any()


def test_vertsortandorder():  # BrickMesh.jl (test):71-95
    v = bm.vertsortandorder
    assert v(1) == ((1,), 1)
    assert v(1, 2) == ((1, 2), 1) and v(2, 1) == ((1, 2), 2)
    for args, o in (((1, 2, 3), 1), ((3, 1, 2), 2), ((2, 3, 1), 3), ((2, 1, 3), 4),
                    ((3, 2, 1), 5), ((1, 3, 2), 6)):
        assert v(*args) == ((1, 2, 3), o)
    for args, o in (((1, 2, 3, 4), 1), ((1, 3, 2, 4), 2), ((2, 1, 3, 4), 3), ((2, 4, 1, 3), 4),
                    ((3, 1, 4, 2), 5), ((3, 4, 1, 2), 6), ((4, 2, 3, 1), 7), ((4, 3, 2, 1), 8)):
        assert v(*args) == ((1, 2, 3, 4), o)
    with pytest.raises(ValueError):
        v(1, 3, 3, 1)


@pytest.mark.parametrize("key,conn", [("mpi_connect", "face"), ("mpi_connectfull", "full")])
def test_three_rank_connectivity_tables(key, conn):
    fx = json.load(open(os.path.join(GOLD, "mesh_connect.json")))
    base = fx["mpi_connect"]
    ll = np.array(base["globalelemtocoord_x0y0"])
    gbndy = np.array(base["globalelemtobndy"]).T
    gface = np.array(base["globalelemtoface"]).T
    for r, exp in enumerate(fx[key]["ranks"]):
        t = M.BrickTopology((np.arange(0, 5), np.arange(5, 10)), boundary=((1, 2), (3, 4)),
                            periodicity=(False, True), connectivity=conn, rank=r, size=3)
        ge = np.array(exp["globalelems"]) - 1
        assert t.nreal == exp["nrealelem"] and t.nelem == len(ge)
        assert np.array_equal(t.elemtocoord[:, 0, :], ll[ge])          # lower-left corners
        assert np.array_equal(t.elemtocoord[:, 3, :], ll[ge] + 1)      # upper-right corners
        assert np.array_equal(t.elemtoelem.T, np.array(exp["elemtoelem"]))
        assert np.array_equal(t.elemtobndy, gbndy[ge])
        assert np.array_equal(t.elemtoface[:t.nreal], gface[ge[:t.nreal]])
        if "elemtoface" in exp:
            assert np.array_equal(t.elemtoface.T, np.array(exp["elemtoface"]))
        assert (t.elemtoordr == 1).all()
        assert t.nabrtorank == exp["nabrtorank"]
        assert [list(x) for x in t.nabrtorecv] == exp["nabrtorecv"]
        assert [list(x) for x in t.nabrtosend] == exp["nabrtosend"]
        both = np.sort(np.concatenate([t.exteriorelems, t.interiorelems]))
        assert np.array_equal(both, np.arange(1, t.nreal + 1))
        assert np.array_equal(np.unique(t.sendelems), t.exteriorelems)


def test_two_rank_stacked_3d_connectivity_tables():
    """test/Numerics/Mesh/mpi_connect_stacked_3d.jl: 3 x 3 x 3 stacked brick on two ranks,
    periodic in y, face connectivity -- element order, coordinates, neighbours, faces,
    boundary tags and the neighbour send / receive ranges."""
    fx = json.load(open(os.path.join(GOLD, "mesh_connect.json")))["mpi_connect_stacked_3d"]
    gcoord = np.array(fx["globalelemtocoord"])            # (27, 3, 8)
    gbndy = np.array(fx["globalelemtobndy"]).T            # (27, 6)
    gface = np.array(fx["globalelemtoface"]).T
    for r, exp in enumerate(fx["ranks"]):
        t = M.StackedBrickTopology((np.arange(1, 5), np.arange(5, 9), np.arange(9, 13)),
                                   periodicity=(False, True, False),
                                   boundary=((1, 2), (3, 4), (5, 6)), connectivity="face",
                                   rank=r, size=2)
        ge = np.array(exp["globalelems"]) - 1
        assert t.nreal == exp["nrealelem"] and t.nelem == len(ge)
        # elemtocoord here is (nelem, nvert, dim); the fixture (dim, nvert) per element
        assert np.array_equal(np.transpose(t.elemtocoord, (0, 2, 1)), gcoord[ge])
        assert np.array_equal(t.elemtoelem.T, np.array(exp["elemtoelem"]))
        assert np.array_equal(t.elemtobndy, gbndy[ge])
        assert np.array_equal(t.elemtoface[:t.nreal], gface[ge[:t.nreal]])
        assert (t.elemtoordr == 1).all()
        assert t.nabrtorank == exp["nabrtorank"]
        assert [list(x) for x in t.nabrtorecv] == exp["nabrtorecv"]
        assert [list(x) for x in t.nabrtosend] == exp["nabrtosend"]
        both = np.sort(np.concatenate([t.exteriorelems, t.interiorelems]))
        assert np.array_equal(both, np.arange(1, t.nreal + 1))
        assert np.array_equal(np.unique(t.sendelems), t.exteriorelems)
        assert len(np.intersect1d(t.exteriorelems, t.interiorelems)) == 0


def test_lgl_and_derivative():  # Elements.jl (test): exact N=4 rule, D exact on P_N
    x, w = M.elements.lglpoints(4)
    assert np.allclose(x, [-1, -np.sqrt(3 / 7), 0, np.sqrt(3 / 7), 1], atol=1e-15)
    assert np.allclose(w, [1 / 10, 49 / 90, 32 / 45, 49 / 90, 1 / 10], atol=1e-15)
    for N in (1, 2, 3, 4, 5, 6, 7):
        x, w = M.elements.lglpoints(N)
        assert abs(w.sum() - 2) < 1e-14
        D = M.elements.spectralderivative(x)
        for p in range(N + 1):
            assert np.allclose(D @ x ** p, p * x ** max(p - 1, 0) if p else 0 * x, atol=1e-12)
        # quadrature exact to degree 2N-1
        for p in range(2 * N):
            assert abs(w @ x ** p - (0 if p % 2 else 2 / (p + 1))) < 1e-13


def _exchange_all(grids, arrays):
    """In-process restatement of the ghost exchange (MPIStateArrays.jl:411-514) between
    the emulated ranks: send face nodes per neighbour range, scatter into ghosts."""
    size = len(grids)
    for r, g in enumerate(grids):
        for n, nbr in enumerate(g.nabrtorank):
            gn = grids[nbr]
            m = gn.nabrtorank.index(r)
            a, b = gn.nabrtovmapsend[m]
            src = gn.vmapsend[a - 1:b] - 1
            c, d = g.nabrtovmaprecv[n]
            dst = g.vmaprecv[c - 1:d] - 1
            assert len(src) == len(dst)
            A, B = arrays[nbr], arrays[r]
            Np = g.Np
            B[dst // Np, :, dst % Np] = A[src // Np, :, src % Np]


@pytest.mark.parametrize("size", [1, 2, 3])
@pytest.mark.parametrize("kind", ["brick", "sphere"])
def test_face_coordinates_match_across_ranks(kind, size):
    """mpi_connect_sphere.jl:60-110: x[vmap-] == x[vmap+] on interior faces, before and
    after a ghost exchange of the coordinates (orientation flips included)."""
    grids = []
    for r in range(size):
        if kind == "brick":
            rng = [np.linspace(0, 1, 4), np.linspace(0, 2, 3), np.linspace(0, 1, 3)]
            t = M.StackedBrickTopology(rng, periodicity=(True, False, False),
                                       boundary=((0, 0), (1, 2), (3, 4)), rank=r, size=size)
            g = M.DiscontinuousSpectralElementGrid(t, 3)
        else:
            R = np.cumsum(np.arange(1.0, 4.0))
            t = M.StackedCubedSphereTopology(3, R, boundary=(1, 2), rank=r, size=size)
            g = M.DiscontinuousSpectralElementGrid(t, 3, meshwarp=M.equiangular_cubed_sphere_warp)
        grids.append(g)
    total = sum(g.nreal for g in grids)
    assert total == (3 * 2 * 2 if kind == "brick" else 6 * 9 * 2)
    xs = []
    for g in grids:
        x = np.stack([g.vgeo[:, 12 + d, :] for d in range(3)], axis=1).copy()
        x[g.nreal:] = np.nan          # ghosts only get what the exchange delivers
        xs.append(x)
    _exchange_all(grids, xs)
    for g, x in zip(grids, xs):
        Np = g.Np
        real_faces = np.zeros(g.vmapM.shape[:2], dtype=bool)
        real_faces[:g.nreal] = g.elemtobndy[:g.nreal] == 0
        vm = g.vmapM[real_faces] - 1
        vp = g.vmapP[real_faces] - 1
        a = x[vm // Np, :, vm % Np]
        b = x[vp // Np, :, vp % Np]
        if kind == "brick":   # periodic direction wraps: compare modulo the period
            d = np.abs(a - b)
            d[..., 0] = np.minimum(d[..., 0], np.abs(d[..., 0] - 1.0))
            assert np.nanmax(d) < 1e-13 and not np.isnan(d).any()
        else:
            assert np.allclose(a, b, atol=1e-12) and not np.isnan(b).any()
        act = np.zeros(Np * g.nelem, dtype=bool)
        act[:Np * g.nreal] = True
        act[g.vmaprecv - 1] = True
        assert np.array_equal(act, g.activedofs)


def test_metric_identities():  # Grids.jl (test):9-97 style: mass sums to the volume
    rng = [np.linspace(0, 2, 3), np.linspace(-1, 1, 4), np.linspace(0, 3, 2)]
    t = M.StackedBrickTopology(rng)
    g = M.DiscontinuousSpectralElementGrid(t, 4)
    assert abs(g.vgeo[:, 9, :].sum() - 2 * 2 * 3) < 1e-12
    assert np.allclose(g.vgeo[:, 9, :] * g.vgeo[:, 10, :], 1)
    # surface mass sums to the face areas of every element
    sM = g.sgeo[..., 3]
    assert abs(sM[:, 0].sum() - t.nreal * (2 / 3) * 3) < 1e-12
    R = np.array([1.0, 2.0])
    ts = M.StackedCubedSphereTopology(4, R)
    gs = M.DiscontinuousSpectralElementGrid(ts, 4, meshwarp=M.equiangular_cubed_sphere_warp)
    vol = 4 / 3 * np.pi * (2 ** 3 - 1)
    assert abs(gs.vgeo[:, 9, :].sum() - vol) / vol < 1e-6
    n = gs.sgeo[..., :3]
    assert np.allclose(np.linalg.norm(n, axis=-1), 1)


@pytest.mark.parametrize("size", [2, 5])
def test_partition_covers_everything_once(size):
    seen = []
    for r in range(size):
        t = M.StackedCubedSphereTopology(3, np.array([1.0, 2.0, 3.0]), rank=r, size=size)
        seen += list(t.globalelems[:t.nreal])
        assert len(t.interiorelems) + len(t.exteriorelems) == t.nreal
    assert sorted(seen) == list(range(1, 6 * 9 * 2 + 1))


@pytest.mark.parametrize("N", [(2, 2, 2), (2, 3, 4), (4, 3, 2)])
def test_metric_terms_of_two_affine_elements(N):
    """test/Numerics/Mesh/Metrics.jl:560-690 ("3-D Metric terms", linear test, Float64): two
    2 x 2 x 2 cubes, the second rotated by a quarter turn about x3, with a different polynomial
    order per direction: coordinates, mass, the nine metric terms, surface mass and normals."""
    G = M.grids
    Nq = [n + 1 for n in N]
    xw = [M.elements.lglpoints(n) for n in N]
    xi, om = [p[0] for p in xw], [p[1] for p in xw]
    D = [M.elements.spectralderivative(x) for x in xi]
    e2c = np.zeros((2, 8, 3))
    e2c[0] = np.array([[0, 2, 0, 2, 0, 2, 0, 2], [0, 0, 2, 2, 0, 0, 2, 2],
                       [0, 0, 0, 0, 2, 2, 2, 2]]).T
    e2c[1] = np.array([[2, 2, 0, 0, 2, 2, 0, 0], [0, 2, 0, 2, 0, 2, 0, 2],
                       [0, 0, 0, 0, 2, 2, 2, 2]]).T
    vgeo, sgeo = G.computegeometry(e2c, D, xi, om)
    col = lambda c: vgeo[:, c, :].reshape(2, Nq[2], Nq[1], Nq[0])   # (e, k, j, i)
    r1, r2, r3 = xi[0][None, None, :], xi[1][None, :, None], xi[2][:, None, None]
    one = np.ones((Nq[2], Nq[1], Nq[0]))
    tol = dict(rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(col(G._x1)[0], (1 + r1) * one, **tol)
    np.testing.assert_allclose(col(G._x1)[1], (1 - r2) * one, **tol)
    np.testing.assert_allclose(col(G._x2)[0], (1 + r2) * one, **tol)
    np.testing.assert_allclose(col(G._x2)[1], (1 + r1) * one, **tol)
    for e in range(2):
        np.testing.assert_allclose(col(G._x3)[e], (1 + r3) * one, **tol)
        np.testing.assert_allclose(col(G._M)[e],
                                   om[2][:, None, None] * om[1][None, :, None] * om[0][None, None, :],
                                   **tol)
    exact = {G._xi1x1: (1, 0), G._xi1x2: (0, 1), G._xi2x1: (0, -1), G._xi2x2: (1, 0),
             G._xi3x3: (1, 1), G._xi1x3: (0, 0), G._xi2x3: (0, 0), G._xi3x1: (0, 0),
             G._xi3x2: (0, 0)}
    for c, (a, b) in exact.items():
        assert np.abs(col(c)[0] - a).max() <= 100 * np.finfo(float).eps
        assert np.abs(col(c)[1] - b).max() <= 100 * np.finfo(float).eps
    normals = np.zeros((2, 6, 3))
    normals[0, 0, 0], normals[0, 1, 0], normals[1, 2, 0], normals[1, 3, 0] = -1, 1, 1, -1
    normals[0, 2, 1], normals[0, 3, 1], normals[1, 0, 1], normals[1, 1, 1] = -1, 1, -1, 1
    normals[:, 4, 2], normals[:, 5, 2] = -1, 1
    Np = int(np.prod(Nq))
    for d in range(3):
        others = [j for j in range(3) if j != d]
        Mf = np.kron(om[others[1]], om[others[0]])          # lower direction fastest
        nfp = Np // Nq[d]
        for f in (2 * d, 2 * d + 1):
            for e in range(2):
                np.testing.assert_allclose(sgeo[e, f, :nfp, G._sM], Mf, **tol)
                for c, k in enumerate((G._n1, G._n2, G._n3)):
                    np.testing.assert_allclose(sgeo[e, f, :nfp, k], normals[e, f, c], **tol)


def _check_rank_tables(t, fx, rk, nface=4):
    ex = fx["ranks"][rk]
    ge = np.array(ex["globalelems"]) - 1
    nr = ex["nrealelem"]
    assert t.nreal == nr and t.nelem == len(ge)
    assert np.array_equal(np.asarray(t.elemtoelem).T, np.array(ex["elemtoelem"]))
    gf = np.array(fx["globalelemtoface"])
    assert np.array_equal(np.asarray(t.elemtoface).T[:, :nr], gf[:, ge[:nr]])
    gb = np.array(fx.get("globalelemtobndy", np.zeros_like(gf)))
    assert np.array_equal(np.asarray(t.elemtobndy).T, gb[:, ge])
    assert np.all(np.asarray(t.elemtoordr) == 1)
    assert list(t.nabrtorank) == ex["nabrtorank"]
    assert [list(r) for r in t.nabrtorecv] == ex["nabrtorecv"]
    assert [list(r) for r in t.nabrtosend] == ex["nabrtosend"]
    real = np.arange(1, nr + 1)
    assert np.array_equal(np.sort(np.union1d(t.exteriorelems, t.interiorelems)), real)
    assert np.array_equal(np.unique(t.sendelems), np.asarray(t.exteriorelems))
    assert len(np.intersect1d(t.exteriorelems, t.interiorelems)) == 0
    return ge


@pytest.mark.parametrize("rk", [0, 1])
def test_two_rank_periodic_brick_tables(rk):
    """test/Numerics/Mesh/mpi_connect_ell.jl: 3 x 2 doubly periodic 2-D brick on two ranks."""
    fx = json.load(open(os.path.join(GOLD, "mesh_connect.json")))["mpi_connect_ell"]
    t = M.BrickTopology([np.linspace(0, 1, 4), np.linspace(0, 1, 3)], periodicity=(True, True),
                        boundary=((1, 2), (3, 4)), connectivity="face", rank=rk, size=2)
    _check_rank_tables(t, fx, rk)


@pytest.mark.parametrize("rk", [0, 1, 2])
def test_three_rank_stacked_brick_tables(rk):
    """test/Numerics/Mesh/mpi_connect_stacked.jl: 3 x 2 stacked 2-D brick, periodic in the
    stacking direction, on three ranks (whole columns per rank)."""
    fx = json.load(open(os.path.join(GOLD, "mesh_connect.json")))["mpi_connect_stacked"]
    t = M.StackedBrickTopology([np.arange(2, 6), np.arange(4, 7)], periodicity=(False, True),
                               boundary=((1, 2), (3, 4)), connectivity="face", rank=rk, size=3)
    ge = _check_rank_tables(t, fx, rk)
    gc = np.array(fx["globalelemtocoord"])             # (elem, dim, vertex)
    ec = np.asarray(t.elemtocoord)                     # (elem, vertex, dim)
    assert np.array_equal(ec[:, :, :2].transpose(0, 2, 1), gc[ge])


def test_hilbert_partition_of_the_reference_brick():
    """test/Numerics/Mesh/mpi_partition.jl:14-92: the 4 x 4 brick (0:4, 5:9) on three ranks;
    each rank's elements in space-filling-curve order (lower-left corners of `etc`) and the
    boundary tags of the non-periodic direction (`etb`)."""
    corners = [[(0, 5), (1, 5), (1, 6), (0, 6), (0, 7)],
               [(0, 8), (1, 8), (1, 7), (2, 7), (2, 8)],
               [(3, 8), (3, 7), (3, 6), (2, 6), (2, 5), (3, 5)]]
    etb = [[[1, 0, 0, 1, 1], [0] * 5], [[1, 0, 0, 0, 0], [0] * 5],
           [[0] * 6, [2, 2, 2, 0, 0, 2]]]
    for rk in range(3):
        t = M.BrickTopology([np.arange(0, 5), np.arange(5, 10)], periodicity=(False, True),
                            boundary=((1, 2), (3, 4)), connectivity="face", rank=rk, size=3)
        nr = t.nreal
        ec = np.asarray(t.elemtocoord)[:nr]
        assert [tuple(int(v) for v in ec[e, 0, :2]) for e in range(nr)] == corners[rk]
        for e in range(nr):                      # vertex order: x fastest (etc[:, :, e])
            x0, y0 = corners[rk][e]
            assert ec[e, :, :2].tolist() == [[x0, y0], [x0 + 1, y0], [x0, y0 + 1], [x0 + 1, y0 + 1]]
        assert np.asarray(t.elemtobndy)[:nr, :2].T.tolist() == etb[rk]
        assert np.all(np.asarray(t.elemtobndy)[:nr, 2:] == 0)
