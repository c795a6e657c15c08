"""Host-side mirror of the reference's ``Topologies`` module.

Reference: ``src/Numerics/Mesh/Topologies.jl`` -- BoxElementTopology :222-292,
BrickTopology :469-532, StackedBrickTopology :631-1020, CubedShellTopology
:1066-1138, cubedshellmesh :1183-1239, cubed_sphere_warp :1253-1298,
StackedCubedSphereTopology :1522-1700.

Every constructor takes ``rank``/``size`` instead of an MPI communicator and
returns that rank's view (see ``brickmesh.py``).  Integer tables are 1-based;
numpy shapes are the reversed Julia shapes.
"""
import math

import numpy as np

from . import brickmesh as bm

__all__ = [
    "Topology", "BrickTopology", "StackedBrickTopology", "CubedShellTopology",
    "StackedCubedSphereTopology", "cubedshellmesh", "equiangular_cubed_sphere_warp",
]


class Topology:
    """Plain record of the reference's ``BoxElementTopology`` fields
    (Topologies.jl:35-292).  ``interiorelems``/``exteriorelems`` follow :250-251."""

    def __init__(self, dim, conn, stacksize=None, periodicstack=False,
                 bndytoelem=(), bndytoface=(), rank=0, size=1):
        self.dim = dim
        self.rank, self.size = rank, size
        self.nelem = conn["nelem"]
        self.nreal = conn["nreal"]
        self.nghost = conn["nghost"]
        self.ghostfaces = conn["ghostfaces"]
        self.sendelems = np.asarray(conn["sendelems"], dtype=np.int64)
        self.sendfaces = conn["sendfaces"]
        self.elemtocoord = conn["elemtocoord"]
        self.elemtoelem = conn["elemtoelem"]
        self.elemtoface = conn["elemtoface"]
        self.elemtoordr = conn["elemtoordr"]
        self.elemtobndy = conn["elemtobndy"]
        self.nabrtorank = list(conn["nabrtorank"])
        self.nabrtorecv = list(conn["nabrtorecv"])      # (first, last) 1-based
        self.nabrtosend = list(conn["nabrtosend"])
        self.globalelems = conn.get("globalelems")
        self.stacksize = stacksize
        self.periodicstack = periodicstack
        self.bndytoelem, self.bndytoface = bndytoelem, bndytoface
        ext = np.unique(self.sendelems)
        self.exteriorelems = ext.astype(np.int64)
        self.interiorelems = np.setdiff1d(
            np.arange(1, self.nreal + 1, dtype=np.int64), ext)

    @property
    def isstacked(self):
        return self.stacksize is not None

    @property
    def realelems(self):
        return range(1, self.nreal + 1)

    @property
    def ghostelems(self):
        return range(self.nreal + 1, self.nelem + 1)


def BrickTopology(elemrange, boundary=None, periodicity=None, connectivity="face",
                  rank=0, size=1):
    """Reference: Topologies.jl:469-532."""
    d = len(elemrange)
    boundary = boundary or tuple((1, 1) for _ in range(d))
    periodicity = periodicity or tuple(False for _ in range(d))
    mesh = bm.brickmesh(elemrange, periodicity, boundary)
    if connectivity == "face":
        conn = bm.connectmesh(mesh, rank, size)
    else:
        conn = bm.connectmeshfull(mesh, rank, size)
    b2e, b2f = bm.enumerateboundaryfaces(conn["elemtoelem"], conn["elemtobndy"],
                                         periodicity, boundary)
    return Topology(d, conn, bndytoelem=b2e, bndytoface=b2f, rank=rank, size=size)


def _stack(base, stacksize, dim, elemtocoord, elemtoordr_map, vert_nbr, vert_bndy,
           periodicity, boundary, rank, size, periodicstack):
    """Shared stacking logic of StackedBrickTopology (Topologies.jl:666-806) and
    StackedCubedSphereTopology (:1545-1685)."""
    nface = 2 * dim
    nh = 2 * (dim - 1)
    nreal = base.nreal * stacksize
    nghost = base.nghost * stacksize
    nelem = nreal + nghost
    j = np.arange(1, stacksize + 1)
    sendelems = (stacksize * (base.sendelems[:, None] - 1) + j[None, :]).reshape(-1)
    ghostfaces = np.zeros((nghost, nface), dtype=bool)
    ghostfaces[:, :nh] = np.repeat(base.ghostfaces, stacksize, axis=0)
    sendfaces = np.zeros((len(sendelems), nface), dtype=bool)
    sendfaces[:, :nh] = np.repeat(base.sendfaces, stacksize, axis=0)
    elemtoelem = np.repeat(np.arange(1, nelem + 1)[:, None], nface, axis=1)
    elemtoface = np.repeat(np.arange(1, nface + 1)[None, :], nelem, axis=0)
    elemtoordr = np.ones((nelem, nface), dtype=np.int64)
    elemtobndy = np.zeros((nelem, nface), dtype=np.int64)
    for i in range(1, base.nreal + 1):
        e1 = stacksize * (i - 1) + j                      # (stacksize,)
        for f in range(nh):
            elemtoelem[e1 - 1, f] = stacksize * (base.elemtoelem[i - 1, f] - 1) + j
            elemtoface[e1 - 1, f] = base.elemtoface[i - 1, f]
            elemtoordr[e1 - 1, f] = elemtoordr_map(base.elemtoordr[i - 1, f])
        for jj in range(1, stacksize + 1):
            e = stacksize * (i - 1) + jj
            (eb, fb), (et, ft) = vert_nbr(i, jj, e)
            elemtoelem[e - 1, nh] = eb
            elemtoelem[e - 1, nh + 1] = et
            elemtoface[e - 1, nh] = fb
            elemtoface[e - 1, nh + 1] = ft
    for i in range(1, base.nelem + 1):
        e1 = stacksize * (i - 1) + j
        vert_bndy(i, e1, elemtobndy, base)
    nabrtorecv = [(stacksize * (a - 1) + 1, stacksize * b) for a, b in base.nabrtorecv]
    nabrtosend = [(stacksize * (a - 1) + 1, stacksize * b) for a, b in base.nabrtosend]
    b2e, b2f = bm.enumerateboundaryfaces(elemtoelem, elemtobndy, periodicity, boundary)
    gl = None
    if base.globalelems is not None:      # global id of a stacked element (test bookkeeping)
        gl = (stacksize * (np.asarray(base.globalelems)[:, None] - 1) + j[None, :]).reshape(-1)
    conn = dict(nelem=nelem, nreal=nreal, nghost=nghost, ghostfaces=ghostfaces,
                sendelems=sendelems, sendfaces=sendfaces, elemtocoord=elemtocoord,
                elemtoelem=elemtoelem, elemtoface=elemtoface, elemtoordr=elemtoordr,
                elemtobndy=elemtobndy, nabrtorank=base.nabrtorank,
                nabrtorecv=nabrtorecv, nabrtosend=nabrtosend,
                globalelems=gl)
    return Topology(dim, conn, stacksize=stacksize, periodicstack=periodicstack,
                    bndytoelem=b2e, bndytoface=b2f, rank=rank, size=size)


def StackedBrickTopology(elemrange, boundary=None, periodicity=None,
                         connectivity="full", rank=0, size=1):
    """Reference: Topologies.jl:631-1020 (DSS tables omitted: out of scope)."""
    dim = len(elemrange)
    assert dim > 1, "Stacked brick topology works for 2D and 3D"
    boundary = boundary or tuple((1, 1) for _ in range(dim))
    periodicity = periodicity or tuple(False for _ in range(dim))
    base = BrickTopology(elemrange[:dim - 1], boundary[:dim - 1], periodicity[:dim - 1],
                         connectivity, rank, size)
    stack = np.asarray(elemrange[dim - 1])
    stacksize = len(stack) - 1
    nvb = 2 ** (dim - 1)
    T = np.result_type(base.elemtocoord.dtype, stack.dtype)
    elemtocoord = np.zeros((base.nelem * stacksize, 2 * nvb, dim), dtype=T)
    ec = elemtocoord.reshape(base.nelem, stacksize, 2 * nvb, dim)
    ec[:, :, :nvb, :dim - 1] = base.elemtocoord[:, None, :, :]
    ec[:, :, nvb:, :dim - 1] = base.elemtocoord[:, None, :, :]
    ec[:, :, :nvb, dim - 1] = stack[None, :-1, None]
    ec[:, :, nvb:, dim - 1] = stack[None, 1:, None]
    nh = 2 * (dim - 1)
    per = periodicity[dim - 1]

    def vert_nbr(i, jj, e1):
        et, eb = e1 + 1, e1 - 1
        ft, fb = nh + 1, nh + 2
        if jj == stacksize:
            et = stacksize * (i - 1) + 1 if per else e1
            ft = ft if per else nh + 2
        if jj == 1:
            eb = stacksize * (i - 1) + stacksize if per else e1
            fb = fb if per else nh + 1
        return (eb, fb), (et, ft)

    def vert_bndy(i, e1, elemtobndy, b):
        for f in range(nh):
            elemtobndy[e1 - 1, f] = b.elemtobndy[i - 1, f]
        if not per:
            elemtobndy[e1[0] - 1, nh] = boundary[dim - 1][0]
            elemtobndy[e1[-1] - 1, nh + 1] = boundary[dim - 1][1]

    def ordr(o):
        assert o == 1
        return o

    return _stack(base, stacksize, dim, elemtocoord, ordr, vert_nbr, vert_bndy,
                  periodicity, boundary, rank, size, per)


def cubedshellmesh(Ne):
    """Global cubed-shell mesh on the flattened net.  Returns ``(elemtovert,
    elemtocoord_flat, elemtobndy, faceconnections)``.
    Reference: Topologies.jl:1183-1239."""
    nelem = 6 * Ne * Ne
    elemtovert = np.zeros((nelem, 4), dtype=np.int64)
    elemtocoord = np.zeros((nelem, 4, 2), dtype=np.int64)
    bx = [0, Ne, 2 * Ne, Ne, Ne, Ne]
    by = [0, 0, 0, Ne, 2 * Ne, 3 * Ne]
    n1 = Ne + 1

    def vertmap(a, b, c):               # 1-based LinearIndices((Ne+1,)*3)
        return a + n1 * ((b - 1) + n1 * (c - 1))

    for e in range(nelem):
        i = e % Ne + 1
        j = (e // Ne) % Ne + 1
        blck = e // (Ne * Ne) + 1
        elemtocoord[e, :, 0] = bx[blck - 1] + np.array([i - 1, i, i - 1, i])
        elemtocoord[e, :, 1] = by[blck - 1] + np.array([j - 1, j - 1, j, j])
        for n in range(1, 5):
            ix = i + (n - 1) % 2
            jx = j + (n - 1) // 2
            if blck == 1:
                v = vertmap(1, Ne + 2 - ix, jx)
            elif blck == 2:
                v = vertmap(ix, 1, jx)
            elif blck == 3:
                v = vertmap(Ne + 1, ix, jx)
            elif blck == 4:
                v = vertmap(ix, jx, Ne + 1)
            elif blck == 5:
                v = vertmap(ix, Ne + 1, Ne + 2 - jx)
            else:
                v = vertmap(ix, Ne + 2 - jx, 1)
            elemtovert[e, n - 1] = v
    elemtobndy = np.zeros((nelem, 4), dtype=np.int64)
    return elemtovert, elemtocoord, elemtobndy, []


def CubedShellTopology(Neside, T=np.float64, connectivity="full", rank=0, size=1):
    """Reference: Topologies.jl:1066-1138.  The Hilbert partition is computed on
    the flattened-net integer coordinates, the 3-D corner coordinates come from
    the vertex lattice ``(2(i-1) - Ne)/Ne``."""
    elemtovert, flat, elemtobndy, fc = cubedshellmesh(Neside)
    nelem = elemtovert.shape[0]
    n1 = Neside + 1
    v = elemtovert - 1
    ijk = np.stack([v % n1, (v // n1) % n1, v // (n1 * n1)], axis=-1)  # 0-based i-1
    elemtocoord = ((2 * ijk - Neside) / Neside).astype(T)
    mesh = (elemtovert, elemtocoord, elemtobndy, fc)
    if connectivity == "face":
        conn = bm.connectmesh(mesh, rank, size, dim=2, partcoord=flat)
    else:
        conn = bm.connectmeshfull(mesh, rank, size, dim=2, partcoord=flat)
    return Topology(2, conn, rank=rank, size=size)


def StackedCubedSphereTopology(Nhorz, Rrange, boundary=(1, 1), connectivity="full",
                               rank=0, size=1):
    """Reference: Topologies.jl:1522-1700 (DSS tables omitted: out of scope)."""
    Rrange = np.asarray(Rrange)
    T = Rrange.dtype
    base = CubedShellTopology(Nhorz, T, connectivity, rank, size)
    dim = 3
    stacksize = len(Rrange) - 1
    elemtocoord = np.zeros((base.nelem * stacksize, 8, 3), dtype=T)
    ec = elemtocoord.reshape(base.nelem, stacksize, 8, 3)
    ec[:, :, :4, :] = base.elemtocoord[:, None, :, :] * Rrange[None, :-1, None, None]
    ec[:, :, 4:, :] = base.elemtocoord[:, None, :, :] * Rrange[None, 1:, None, None]
    nh = 4

    def vert_nbr(i, jj, e1):
        eb = e1 if jj == 1 else e1 - 1
        et = e1 if jj == stacksize else e1 + 1
        fb = nh + 1 if jj == 1 else nh + 2
        ft = nh + 2 if jj == stacksize else nh + 1
        return (eb, fb), (et, ft)

    def vert_bndy(i, e1, elemtobndy, b):
        elemtobndy[e1[0] - 1, nh] = boundary[0]
        elemtobndy[e1[-1] - 1, nh + 1] = boundary[1]

    def ordr(o):
        assert o in (1, 2)
        return 1 if o == 1 else 3

    return _stack(base, stacksize, dim, elemtocoord, ordr, vert_nbr, vert_bndy,
                  (False,), (tuple(boundary),), rank, size, False)


def equiangular_cubed_sphere_warp(a, b, c, R=None):
    """Vectorised ``cubed_sphere_warp(EquiangularCubedSphere(), a, b, c, R)``.
    Reference: Topologies.jl:1253-1298.  The face branch order (``argmax`` picks
    the first maximal |.|) and the operation order are kept."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64)
    if R is None:
        R = np.maximum(np.maximum(np.abs(a), np.abs(b)), np.abs(c))
    absv = np.stack([np.abs(a), np.abs(b), np.abs(c)])
    fdim = np.argmax(absv, axis=0)        # first maximum, like Julia's argmax
    x1 = np.zeros_like(a)
    x2 = np.zeros_like(a)
    x3 = np.zeros_like(a)

    def f(sR, xi, eta):
        X = np.tan(math.pi * xi / 4)
        Y = np.tan(math.pi * eta / 4)
        z1 = sR / np.sqrt(X ** 2 + Y ** 2 + 1)
        return z1, X * z1, Y * z1

    with np.errstate(divide="ignore", invalid="ignore"):
        m = (fdim == 0) & (a < 0)
        z = f(-R, b / a, c / a)
        x1[m], x2[m], x3[m] = z[0][m], z[1][m], z[2][m]
        m = (fdim == 1) & (b < 0)
        z = f(-R, a / b, c / b)
        x2[m], x1[m], x3[m] = z[0][m], z[1][m], z[2][m]
        m = (fdim == 0) & (a > 0)
        z = f(R, b / a, c / a)
        x1[m], x2[m], x3[m] = z[0][m], z[1][m], z[2][m]
        m = (fdim == 1) & (b > 0)
        z = f(R, a / b, c / b)
        x2[m], x1[m], x3[m] = z[0][m], z[1][m], z[2][m]
        m = (fdim == 2) & (c > 0)
        z = f(R, b / c, a / c)
        x3[m], x2[m], x1[m] = z[0][m], z[1][m], z[2][m]
        m = (fdim == 2) & (c < 0)
        z = f(-R, b / c, a / c)
        x3[m], x2[m], x1[m] = z[0][m], z[1][m], z[2][m]
    return x1, x2, x3
