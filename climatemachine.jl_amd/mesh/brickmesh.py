"""Host-side mirror of the reference's ``BrickMesh`` module (mesh generation,
Hilbert-curve partition, face / vertex connectivity).

Reference: ``src/Numerics/Mesh/BrickMesh.jl`` (linearpartition :17, hilbertcode :40,
centroidtocode :112, brickmesh :272, getpartition :449, partition :531,
vertsortandorder :667-791, connectmesh :827, enumerateboundaryfaces! :1096,
connectmeshfull :1168).

The reference builds the mesh with MPI collectives (sample sort, Alltoallv).  The
result on every rank is a deterministic function of the *global* mesh and of
``(rank, size)``: the elements sorted by Hilbert code are cut into ``size`` equal
contiguous chunks, each rank re-sorts its chunk by code, ghost elements are
numbered by (owner rank, owner-local id).  This module computes that function
directly, so any rank's view can be produced in any process with no
communication -- which is also how the multi-rank tests emulate 2-5 ranks.

Conventions: integer tables hold the reference's **1-based** values; arrays are
numpy C-ordered with the reference's (column-major) dimensions reversed, so the
memory image equals the Julia array's (e.g. ``elemtoelem[e-1, f-1]``).
"""
from fractions import Fraction

import numpy as np

__all__ = [
    "linearpartition", "hilbertcode", "centroidtocode", "brickmesh",
    "partition_order", "vertsortandorder", "connectmesh", "connectmeshfull",
    "enumerateboundaryfaces", "build_fmask",
]

_U64 = (1 << 64) - 1


def linearpartition(n, p, nparts):
    """1-based inclusive range (as a python ``range``) of piece ``p`` of ``1:n``.
    Reference: BrickMesh.jl:17-18."""
    return range((p - 1) * n // nparts + 1, p * n // nparts + 1)


def _hilbert_ints(X, bits):
    """Scalar (python int) version for arbitrary ``bits``; ``X`` list of ints."""
    X = list(X)
    n = len(X)
    M = 1 << (bits - 1)
    Q = M
    for _ in range(bits - 1):
        P = Q - 1
        for i in range(n):
            if X[i] & Q:
                X[0] ^= P
            else:
                t = (X[0] ^ X[i]) & P
                X[0] ^= t
                X[i] ^= t
        Q >>= 1
    for i in range(1, n):
        X[i] ^= X[i - 1]
    t = 0
    Q = M
    for _ in range(bits - 1):
        if X[n - 1] & Q:
            t ^= Q - 1
        Q >>= 1
    X = [x ^ t for x in X]
    H = [0] * n
    for i in range(n):
        for j in range(bits):
            k = i * bits + j
            bit = (X[n - 1 - (k % n)] >> (k // n)) & 1
            H[n - 1 - i] |= bit << j
    return H


def _hilbert_u64(X):
    """Vectorised 64-bit version; ``X`` is ``(n, nelem)`` uint64."""
    X = np.array(X, dtype=np.uint64, copy=True)
    n, ne = X.shape
    bits = 64
    Q = 1 << (bits - 1)
    for _ in range(bits - 1):
        q, p = np.uint64(Q), np.uint64(Q - 1)
        for i in range(n):
            sel = (X[i] & q) != 0
            t = (X[0] ^ X[i]) & p
            x0 = np.where(sel, X[0] ^ p, X[0] ^ t)
            if i != 0:
                X[i] = np.where(sel, X[i], X[i] ^ t)
            X[0] = x0
        Q >>= 1
    for i in range(1, n):
        X[i] = X[i] ^ X[i - 1]
    t = np.zeros(ne, dtype=np.uint64)
    Q = 1 << (bits - 1)
    for _ in range(bits - 1):
        sel = (X[n - 1] & np.uint64(Q)) != 0
        t = np.where(sel, t ^ np.uint64(Q - 1), t)
        Q >>= 1
    X = X ^ t[None, :]
    H = np.zeros((n, ne), dtype=np.uint64)
    one = np.uint64(1)
    for i in range(n):
        for j in range(bits):
            k = i * bits + j
            bit = (X[n - 1 - (k % n)] >> np.uint64(k // n)) & one
            H[n - 1 - i] |= bit << np.uint64(j)
    return H


def hilbertcode(Y, bits=64):
    """Skilling's AxestoTranspose followed by the bit transposition.  ``Y`` is
    ``(n,)`` or ``(n, nelem)``; the result has the same shape.
    Reference: BrickMesh.jl:40-93."""
    Y = np.asarray(Y)
    if bits == 64:
        H = _hilbert_u64(Y.reshape(Y.shape[0], -1))
        return H[:, 0] if Y.ndim == 1 else H
    if Y.ndim == 1:
        return np.array(_hilbert_ints([int(v) for v in Y], bits))
    cols = [_hilbert_ints([int(v) for v in Y[:, e]], bits) for e in range(Y.shape[1])]
    return np.array(cols).T


def centroidtocode(elemtocoord, cmin=None, cmax=None):
    """Hilbert code (``(d, nelem)`` uint64) of every element centroid.

    ``elemtocoord`` has numpy shape ``(nelem, nvert, d)``.  ``cmin/cmax`` are the
    global bounding box of the centroids (the reference Allreduces them,
    BrickMesh.jl:123-130); default: the box of the given elements.
    The integer coordinate is ``floor(typemax(UInt64) * BigFloat(c, 128 bits))``
    (BrickMesh.jl:146-151) which is exact, so it is evaluated in rationals."""
    e2c = np.asarray(elemtocoord, dtype=np.float64)
    nelem, nvert, d = e2c.shape
    # Julia: sum(elemtocorner, dims = 2) ./ nvert  (sequential sum over vertices)
    cent = np.zeros((nelem, d))
    for v in range(nvert):
        cent += e2c[:, v, :]
    cent = cent / nvert
    if cmin is None:
        cmin = cent.min(axis=0) if nelem else np.full(d, np.inf)
        cmax = cent.max(axis=0) if nelem else np.full(d, -np.inf)
    size = np.asarray(cmax, dtype=np.float64) - np.asarray(cmin, dtype=np.float64)
    if not size.any():
        size = np.ones(d)
    else:
        mx = size.max()
        size = np.where(size == 0, mx, size)
    c = (cent - cmin) / size
    X = np.zeros((d, nelem), dtype=np.uint64)
    for e in range(nelem):
        for i in range(d):
            fr = Fraction(float(c[e, i])) * _U64
            X[i, e] = fr.numerator // fr.denominator
    return hilbertcode(X, bits=64)


def _cart_strides(dims):
    s = [1]
    for n in dims[:-1]:
        s.append(s[-1] * n)
    return s


def build_fmask(d):
    """Face -> local-vertex mask, ``fmask[f][:]`` 0-based vertex ids (vertices in
    Cartesian order).  Reference: BrickMesh.jl:303-311 / :1495-1507."""
    nvert = 2 ** d
    p = np.arange(nvert).reshape([2] * d, order="F")
    fm = []
    for f in range(2 * d):
        idx = [slice(None)] * d
        idx[f // 2] = f % 2
        fm.append(p[tuple(idx)].flatten(order="F"))
    return fm


def brickmesh(x, periodic, boundary=None):
    """Global brick mesh.  Returns ``(elemtovert (nelem,nvert), elemtocoord
    (nelem,nvert,d), elemtobndy (nelem,nface), faceconnections list)`` with the
    reference's 1-based vertex/element/face numbers.  Reference:
    BrickMesh.jl:272-348 (called with part=1, numparts=1: the final partition does
    not depend on the initial linear split)."""
    d = len(x)
    if boundary is None:
        boundary = tuple((1, 1) for _ in range(d))
    x = [np.asarray(xi) for xi in x]
    nelemdim = [len(xi) - 1 for xi in x]
    nvertdim = [len(xi) for xi in x]
    nelem = int(np.prod(nelemdim))
    nvert = 2 ** d
    nface = 2 * d
    vs = _cart_strides(nvertdim)
    T = np.result_type(*[xi.dtype for xi in x])
    elemtovert = np.zeros((nelem, nvert), dtype=np.int64)
    elemtocoord = np.zeros((nelem, nvert, d), dtype=T)
    elemtobndy = np.zeros((nelem, nface), dtype=np.int64)
    faceconnections = []
    fmask = build_fmask(d)
    # element cartesian indices, first dim fastest
    ec = np.stack(np.unravel_index(np.arange(nelem), nelemdim, order="F"), axis=1)
    for v in range(nvert):
        off = [(v >> j) & 1 for j in range(d)]
        vid = np.zeros(nelem, dtype=np.int64)
        for j in range(d):
            vid += (ec[:, j] + off[j]) * vs[j]
            elemtocoord[:, v, j] = x[j][ec[:, j] + off[j]]
        elemtovert[:, v] = vid + 1
    for i in range(d):
        if not periodic[i]:
            elemtobndy[ec[:, i] == 0, 2 * i] = boundary[i][0]
            elemtobndy[ec[:, i] == nelemdim[i] - 1, 2 * i + 1] = boundary[i][1]
    for e in range(nelem):
        for i in range(d):
            if periodic[i] and ec[e, i] == nelemdim[i] - 1:
                verts = []
                for v in fmask[2 * i]:       # low face of the wrapped neighbour
                    vid = 0
                    for j in range(d):
                        o = (int(v) >> j) & 1
                        cj = o if j == i else ec[e, j] + o
                        vid += cj * vs[j]
                    verts.append(vid + 1)
                faceconnections.append([e + 1, 2 * i + 2] + verts)
    return elemtovert, elemtocoord, elemtobndy, faceconnections


def partition_order(elemtocoord, size):
    """Global Hilbert ordering and the per-rank chunks.

    Returns ``(order, starts)``: ``order`` = 0-based global element ids sorted by
    code; rank ``r`` owns ``order[starts[r]:starts[r+1]]`` in that (code) order.
    Reference: BrickMesh.jl:449-522 (global sort, equal split by sorted position)
    and :621-632 (local re-sort by code)."""
    code = centroidtocode(elemtocoord)
    d, nelem = code.shape
    keys = [np.arange(nelem)] + [code[i] for i in range(d - 1, -1, -1)]
    order = np.lexsort(keys)
    starts = [0]
    for r in range(size):
        starts.append(linearpartition(nelem, r + 1, size)[-1] if nelem else 0)
    return order, starts


def _minmaxflip(x, y):
    return (y, x, True) if y < x else (x, y, False)


def vertsortandorder(*v):
    """Sorted vertex tuple and orientation code.  Reference: BrickMesh.jl:667-791."""
    if len(v) == 1:
        return (v[0],), 1
    if len(v) == 2:
        a, b, s1 = _minmaxflip(*v)
        return (a, b), (2 if s1 else 1)
    if len(v) == 3:
        a, b, c = v
        b, c, s1 = _minmaxflip(b, c)
        a, c, s2 = _minmaxflip(a, c)
        a, b, s3 = _minmaxflip(a, b)
        table = {(0, 0, 0): 1, (0, 1, 1): 2, (1, 0, 1): 3, (0, 0, 1): 4,
                 (1, 1, 1): 5, (1, 0, 0): 6}
        key = (int(s1), int(s2), int(s3))
        if key not in table:
            raise ValueError("Problem finding vertex ordering %s" % ((a, b, c),))
        return (a, b, c), table[key]
    if len(v) == 4:
        a, b, c, d = v
        a, b, s1 = _minmaxflip(a, b)
        c, d, s2 = _minmaxflip(c, d)
        a, c, s3 = _minmaxflip(a, c)
        b, d, s4 = _minmaxflip(b, d)
        b, c, s5 = _minmaxflip(b, c)
        table = {(0, 0, 0, 0, 0): 1, (0, 0, 0, 0, 1): 2, (1, 0, 0, 0, 0): 3,
                 (0, 0, 1, 1, 1): 4, (1, 1, 0, 0, 1): 5, (0, 0, 1, 1, 0): 6,
                 (1, 1, 1, 1, 1): 7, (1, 1, 1, 1, 0): 8}
        key = tuple(int(s) for s in (s1, s2, s3, s4, s5))
        if key not in table:
            raise ValueError("Problem finding vertex ordering %s" % ((a, b, c, d),))
        return (a, b, c, d), table[key]
    raise ValueError("unsupported face size")


class _Part:
    """Global mesh + ownership bookkeeping shared by the two connect routines."""

    def __init__(self, mesh, size, order=None, starts=None, partcoord=None):
        self.elemtovert, self.elemtocoord, self.elemtobndy, self.fc = mesh
        self.size = size
        if order is None:
            order, starts = partition_order(
                self.elemtocoord if partcoord is None else partcoord, size)
        self.order, self.starts = order, starts
        nelem = len(order)
        self.owner = np.zeros(nelem, dtype=np.int64)
        self.lid = np.zeros(nelem, dtype=np.int64)      # 1-based owner-local id
        for r in range(size):
            g = order[starts[r]:starts[r + 1]]
            self.owner[g] = r
            self.lid[g] = np.arange(1, len(g) + 1)

    def local(self, rank):
        return self.order[self.starts[rank]:self.starts[rank + 1]]


def _face_keys(elemtovert, fc, d):
    """Per (elem, face): sorted vertex tuple + orientation (with periodic
    replacement).  Reference: BrickMesh.jl:856-882."""
    nelem, nvert = elemtovert.shape
    fmask = build_fmask(d)
    nface = 2 * d
    keys = {}
    orient = np.ones((nelem, nface), dtype=np.int64)
    for e in range(nelem):
        for f in range(nface):
            fv, o = vertsortandorder(*[int(elemtovert[e, v]) for v in fmask[f]])
            keys[(e, f)] = fv
            orient[e, f] = o
    for c in fc:
        e, f, v = c[0] - 1, c[1] - 1, c[2:]
        fv, o = vertsortandorder(*[int(q) for q in v])
        keys[(e, f)] = fv
        orient[e, f] = o
    return keys, orient


def connectmesh(mesh, rank=0, size=1, dim=None, part=None, partcoord=None):
    """Face-connected view of rank ``rank``.  ``mesh`` is the *global*
    ``(elemtovert, elemtocoord, elemtobndy, faceconnections)``.
    Reference: BrickMesh.jl:827-1088."""
    part = part or _Part(mesh, size, partcoord=partcoord)
    elemtovert, elemtocoord, elemtobndy = part.elemtovert, part.elemtocoord, part.elemtobndy
    d = dim if dim is not None else elemtocoord.shape[2]
    nface = 2 * d
    keys, orient = _face_keys(elemtovert, part.fc, d)
    match = {}
    for (e, f), k in keys.items():
        match.setdefault(k, []).append((e, f))
    mine = part.local(rank)
    nelem = len(mine)
    # neighbour info for every local (e, f)
    nbr = {}
    ghosts = set()
    sendpairs = set()
    for le, g in enumerate(mine):
        for f in range(nface):
            lst = match[keys[(g, f)]]
            other = None
            if len(lst) >= 2:
                # the reference pairs adjacent equal columns after a sort; for
                # conforming meshes there are exactly two
                other = lst[1] if lst[0] == (g, f) else lst[0]
            if other is None:
                nbr[(le, f)] = None
            else:
                ge, gf = other
                nbr[(le, f)] = (ge, gf)
                r = int(part.owner[ge])
                if r != rank:
                    ghosts.add((r, int(part.lid[ge]), ge))
                    sendpairs.add((r, le + 1))
    ghostlist = sorted(ghosts)
    gidx = {ge: nelem + i + 1 for i, (_, _, ge) in enumerate(ghostlist)}
    nghost = len(ghostlist)
    sendlist = sorted(sendpairs)
    sendelems = np.array([e for _, e in sendlist], dtype=np.int64)
    sidx = {p: i for i, p in enumerate(sendlist)}
    sendfaces = np.zeros((len(sendlist), nface), dtype=bool)
    ghostfaces = np.zeros((nghost, nface), dtype=bool)
    ntot = nelem + nghost
    elemtoelem = np.repeat(np.arange(1, ntot + 1)[:, None], nface, axis=1)
    elemtoface = np.repeat(np.arange(1, nface + 1)[None, :], ntot, axis=0)
    elemtoordr = np.ones((ntot, nface), dtype=np.int64)
    for le, g in enumerate(mine):
        for f in range(nface):
            o = nbr[(le, f)]
            if o is None:
                continue
            ge, gf = o
            r = int(part.owner[ge])
            if r != rank:
                elemtoelem[le, f] = gidx[ge]
                sendfaces[sidx[(r, le + 1)], f] = True
                ghostfaces[gidx[ge] - nelem - 1, gf] = True
            else:
                elemtoelem[le, f] = part.lid[ge]
            elemtoface[le, f] = gf + 1
            mo, no = orient[g, f], orient[ge, gf]
            if d == 2:
                elemtoordr[le, f] = 1 if no == mo else 2
            else:
                if no != 1 or mo != 1:
                    raise NotImplementedError("TODO add support for other orientations")
                elemtoordr[le, f] = 1
    allg = list(mine) + [ge for _, _, ge in ghostlist]
    newcoord = elemtocoord[allg]
    newbndy = elemtobndy[allg]
    ranks_recv = sorted({r for r, _, _ in ghostlist})
    ranks_send = sorted({r for r, _ in sendlist})
    assert ranks_recv == ranks_send
    nabrtorecv, nabrtosend = [], []
    for r in ranks_recv:
        ids = [i + 1 for i, (rr, _, _) in enumerate(ghostlist) if rr == r]
        nabrtorecv.append((ids[0], ids[-1]))
        ids = [i + 1 for i, (rr, _) in enumerate(sendlist) if rr == r]
        nabrtosend.append((ids[0], ids[-1]))
    return dict(
        nelem=ntot, nreal=nelem, nghost=nghost,
        ghostfaces=ghostfaces, sendelems=sendelems, sendfaces=sendfaces,
        elemtocoord=newcoord, elemtovert=None,
        elemtoelem=elemtoelem, elemtoface=elemtoface, elemtoordr=elemtoordr,
        elemtobndy=newbndy.copy(), nabrtorank=ranks_recv,
        nabrtorecv=nabrtorecv, nabrtosend=nabrtosend,
        globalelems=np.array(allg, dtype=np.int64) + 1,
    )


def _global_vertex_classes(part, d):
    """Periodic vertex identification ``gldofv`` (1-based ids, index 0 unused).
    Reference: BrickMesh.jl:1212-1261 (connections gathered in rank order)."""
    fmask = build_fmask(d)
    nfvert = len(fmask[0])
    elemtovert = part.elemtovert
    nvertg = int(elemtovert.max())
    vconng = []
    fc_by_elem = {}
    for c in part.fc:
        fc_by_elem.setdefault(c[0] - 1, []).append(c)
    for r in range(part.size):
        seen = set()
        for g in part.local(r):
            for c in fc_by_elem.get(int(g), []):
                f, v = c[1] - 1, c[2:]
                fv, _ = vertsortandorder(*[int(elemtovert[g, q]) for q in fmask[f]])
                vv, _ = vertsortandorder(*[int(q) for q in v])
                for i in range(nfvert):
                    pr = (fv[i], vv[i])
                    if pr not in seen:
                        seen.add(pr)
                        vconng.append(pr)
    gldofv = -np.ones(nvertg + 1, dtype=np.int64)
    pmarker = [-1] * len(vconng)
    for i, (v1, v2) in enumerate(vconng):
        if gldofv[v1] == -1 and gldofv[v2] == -1:
            gldofv[v1] = gldofv[v2] = min(v1, v2)
            pmarker[i] = 1
    for i, (v1, v2) in enumerate(vconng):
        if pmarker[i] == -1:
            idv = min(gldofv[v1], gldofv[v2])
            gldofv[v1] = gldofv[v2] = idv
    for i in range(1, nvertg + 1):
        if gldofv[i] == -1:
            gldofv[i] = i
    return gldofv


def connectmeshfull(mesh, rank=0, size=1, dim=2, part=None, partcoord=None):
    """Vertex-connected view of rank ``rank`` of a 2-D (possibly embedded) mesh.
    Reference: BrickMesh.jl:1168-1475."""
    assert dim == 2
    part = part or _Part(mesh, size, partcoord=partcoord)
    elemtovert, elemtocoord, elemtobndy = part.elemtovert, part.elemtocoord, part.elemtobndy
    nvert = elemtovert.shape[1]
    nface = 2 * dim
    fmask = build_fmask(dim)
    nfvert = len(fmask[0])
    gldofv = _global_vertex_classes(part, dim)
    gvert = gldofv[elemtovert]                   # (nelemg, nvert) mapped ids
    # vertex -> set of owning ranks
    vprocs = {}
    velems = {}
    for g in range(elemtovert.shape[0]):
        for v in range(nvert):
            gv = int(gvert[g, v])
            vprocs.setdefault(gv, set()).add(int(part.owner[g]))
            velems.setdefault(gv, []).append(g)
    mine = part.local(rank)
    nelem = len(mine)
    recv = {}
    send = {}
    interior, exterior = [], []
    for le, g in enumerate(mine):
        is_int = True
        for v in range(nvert):
            gv = int(gvert[g, v])
            for ge in velems[gv]:
                r = int(part.owner[ge])
                if r != rank:
                    recv.setdefault(r, set()).add(int(part.lid[ge]))
                    send.setdefault(r, set()).add(le + 1)
                    is_int = False
        (interior if is_int else exterior).append(le + 1)
    nabrtorank = sorted(send.keys())
    assert nabrtorank == sorted(recv.keys())
    sendlist, ghostlist = [], []
    nabrtosend, nabrtorecv = [], []
    for r in nabrtorank:
        s = sorted(send[r])
        nabrtosend.append((len(sendlist) + 1, len(sendlist) + len(s)))
        sendlist += [(r, e) for e in s]
        g = sorted(recv[r])
        nabrtorecv.append((len(ghostlist) + 1, len(ghostlist) + len(g)))
        ghostlist += [(r, e) for e in g]
    nsend, nghost = len(sendlist), len(ghostlist)
    sendfaces = np.zeros((nsend, nface), dtype=bool)
    ghostfaces = np.zeros((nghost, nface), dtype=bool)
    gids = []
    for i, (r, lid) in enumerate(ghostlist):
        ge = int(part.order[part.starts[r] + lid - 1])
        gids.append(ge)
        vm = [rank in vprocs[int(gvert[ge, v])] for v in range(nvert)]
        for f in range(nface):
            if any(vm[q] for q in fmask[f]):
                ghostfaces[i, f] = True
    for i, (r, le) in enumerate(sendlist):
        g = int(mine[le - 1])
        vm = [r in vprocs[int(gvert[g, v])] for v in range(nvert)]
        for f in range(nface):
            if any(vm[q] for q in fmask[f]):
                sendfaces[i, f] = True
    allg = [int(g) for g in mine] + gids
    ntot = len(allg)
    # face matching among local real + ghost elements
    fkeys = {}
    forient = np.ones((ntot, nface), dtype=np.int64)
    for le, g in enumerate(allg):
        for f in range(nface):
            fv, o = vertsortandorder(*[int(elemtovert[g, q]) for q in fmask[f]])
            key = tuple(int(gldofv[q]) for q in fv)
            fkeys.setdefault(key, []).append((le, f))
            forient[le, f] = o
    elemtoelem = np.repeat(np.arange(1, ntot + 1)[:, None], nface, axis=1)
    elemtoface = np.repeat(np.arange(1, nface + 1)[None, :], ntot, axis=0)
    elemtoordr = np.ones((ntot, nface), dtype=np.int64)
    for key, lst in fkeys.items():
        # the reference walks the sorted columns pairing neighbours two by two
        j = 0
        while j < len(lst):
            if j + 1 < len(lst):
                (e1, f1), (e2, f2) = lst[j], lst[j + 1]
                elemtoelem[e1, f1] = e2 + 1
                elemtoface[e1, f1] = f2 + 1
                elemtoelem[e2, f2] = e1 + 1
                elemtoface[e2, f2] = f1 + 1
                o = 1 if forient[e1, f1] == forient[e2, f2] else 2
                elemtoordr[e1, f1] = elemtoordr[e2, f2] = o
                j += 2
            else:
                j += 1
    # locally unique vertex numbers (for completeness; DSS is out of scope)
    ev = gldofv[elemtovert[allg]]
    uniq = np.unique(ev)
    remap = {int(v): i + 1 for i, v in enumerate(uniq)}
    elemtouvert = np.vectorize(remap.get)(ev) if ev.size else ev
    return dict(
        nelem=ntot, nreal=nelem, nghost=nghost,
        ghostfaces=ghostfaces, sendelems=np.array([e for _, e in sendlist], dtype=np.int64),
        sendfaces=sendfaces, elemtocoord=elemtocoord[allg], elemtovert=elemtouvert,
        elemtoelem=elemtoelem, elemtoface=elemtoface, elemtoordr=elemtoordr,
        elemtobndy=elemtobndy[allg].copy(), nabrtorank=nabrtorank,
        nabrtorecv=nabrtorecv, nabrtosend=nabrtosend,
        globalelems=np.array(allg, dtype=np.int64) + 1,
    )


def enumerateboundaryfaces(elemtoelem, elemtobndy, periodicity, boundary):
    """Overwrite ``elemtoelem`` on boundary faces with a running per-tag counter and
    return ``(bndytoelem, bndytoface)``.  Reference: BrickMesh.jl:1096-1128."""
    nb = 0
    for i, per in enumerate(periodicity):
        if not per:
            nb = max(nb, *boundary[i])
    assert nb <= 6
    bndytoelem = [[] for _ in range(nb)]
    bndytoface = [[] for _ in range(nb)]
    N = [0] * nb
    nelem, nface = elemtoelem.shape
    for e in range(nelem):
        for f in range(nface):
            dd = int(elemtobndy[e, f])
            assert 0 <= dd <= nb
            if dd != 0:
                N[dd - 1] += 1
                elemtoelem[e, f] = N[dd - 1]
                bndytoelem[dd - 1].append(e + 1)
                bndytoface[dd - 1].append(f + 1)
    return bndytoelem, bndytoface
