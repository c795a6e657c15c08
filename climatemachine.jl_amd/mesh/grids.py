"""Host-side mirror of ``DiscontinuousSpectralElementGrid``: the producer of every
table the DG kernels read (``vgeo, sgeo, vmap-/+, elemtobndy, vmapsend/recv,
interior/exterior lists, activedofs, D, omega``).

Reference: ``src/Numerics/Mesh/Grids.jl`` -- vgeo/sgeo ids :76-146, constructor
:267-413, mappings :559-637, commmapping :761-811, computegeometry :1028-1124,
horizontal_metrics! :1135-1154; ``Metrics.jl`` creategrid! :85-114, jacobian
:215-264, computemetric! (2-D :328-398, 3-D :431-722); ``GeometricFactors.jl``
:18-119 (25-column vgeo, 5-row sgeo).

Layouts (numpy C order = reversed Julia dims): ``vgeo (nelem, 25, Np)``,
``sgeo (nelem, nface, Nfp, 5)``, ``vmapM/vmapP (nelem, nface, Nfp)`` int64 1-based,
``elemtobndy (nelem, nface)``.
"""
import numpy as np

from . import elements

# vgeo column ids (0-based; reference ids are these + 1)
(_xi1x1, _xi2x1, _xi3x1, _xi1x2, _xi2x2, _xi3x2, _xi1x3, _xi2x3, _xi3x3,
 _M, _MI, _MH, _x1, _x2, _x3, _JcV,
 _x1xi1, _x2xi1, _x3xi1, _x1xi2, _x2xi2, _x3xi2, _x1xi3, _x2xi3, _x3xi3) = range(25)
NVGEO = 25
_n1, _n2, _n3, _sM, _vMI = range(5)
NSGEO = 5

__all__ = ["DiscontinuousSpectralElementGrid", "mappings", "commmapping",
           "computegeometry", "NVGEO", "NSGEO"]


def _fmask(Nq):
    """fmask[f] = 0-based node ids of face f (Grids.jl:573-597)."""
    d = len(Nq)
    p = np.arange(int(np.prod(Nq))).reshape(Nq, order="F")
    fm = []
    for f in range(2 * d):
        idx = [slice(None)] * d
        idx[f // 2] = 0 if f % 2 == 0 else Nq[f // 2] - 1
        fm.append(p[tuple(idx)].flatten(order="F"))
    return fm


def mappings(N, elemtoelem, elemtoface, elemtoordr):
    """``vmap-``/``vmap+`` (numpy ``(nelem, nface, maxNfp)``, 1-based linear DOF ids,
    0 padding).  Reference: Grids.jl:559-637."""
    nelem, nfaces = elemtoelem.shape
    d = nfaces // 2
    Nq = [n + 1 for n in N]
    Np = int(np.prod(Nq))
    Nfp = [Np // q for q in Nq]
    fmask = _fmask(Nq)
    maxfp = max(Nfp)
    vmapM = np.zeros((nelem, nfaces, maxfp), dtype=np.int64)
    vmapP = np.zeros((nelem, nfaces, maxfp), dtype=np.int64)
    # flipped (first face index reversed) masks for orientation 3
    flipped = []
    for f in range(nfaces):
        dims = [Nq[j] for j in range(d) if j != f // 2]
        if d == 3:
            inds = np.arange(len(fmask[f])).reshape(dims, order="F")
            flipped.append(fmask[f][inds[::-1, :].flatten(order="F")])
        else:
            flipped.append(None)
    e1 = np.arange(nelem)
    for f1 in range(nfaces):
        d1 = f1 // 2
        n1 = Nfp[d1]
        vmapM[:, f1, :n1] = Np * e1[:, None] + fmask[f1][None, :n1] + 1
        e2 = elemtoelem[:, f1] - 1
        f2 = elemtoface[:, f1] - 1
        o2 = elemtoordr[:, f1]
        for ff in range(nfaces):
            for oo in (1, 3) if d == 3 else (1,):
                sel = (f2 == ff) & (o2 == oo)
                if not sel.any():
                    continue
                assert Nfp[ff // 2] == n1
                fm = fmask[ff] if oo == 1 else flipped[ff]
                vmapP[sel, f1, :n1] = Np * e2[sel, None] + fm[None, :n1] + 1
        bad = ~np.isin(o2, (1, 3) if d == 3 else (1,))
        if bad.any():
            raise NotImplementedError(
                "Orientation '%d' with dim '%d' not supported yet" % (o2[bad][0], d))
    return vmapM, vmapP


def commmapping(N, commelems, commfaces, nabrtocomm):
    """Face-node communication map.  Returns ``(vmapC, nabrtovmapC)`` with
    1-based DOF ids and (first, last) 1-based inclusive ranges.
    Reference: Grids.jl:761-811."""
    commelems = np.asarray(commelems)
    nelem = len(commelems)
    nface = commfaces.shape[1] if commfaces.ndim == 2 else 0
    d = len(N)
    Nq = [n + 1 for n in N]
    Np = int(np.prod(Nq))
    ci = np.stack(np.unravel_index(np.arange(Np), Nq, order="F"), axis=1)
    vmapC = []
    ranges = []
    e = 0
    for (a, b) in nabrtocomm:
        rbegin = len(vmapC) + 1
        for ne in range(a, b + 1):
            ce = commelems[ne - 1]
            add = np.zeros(Np, dtype=bool)
            for j in range(d):
                if commfaces[e, 2 * j]:
                    add |= ci[:, j] == 0
                if commfaces[e, 2 * j + 1]:
                    add |= ci[:, j] == Nq[j] - 1
            vmapC.extend(((ce - 1) * Np + np.nonzero(add)[0] + 1).tolist())
            e += 1
        ranges.append((rbegin, len(vmapC)))
    return np.array(vmapC, dtype=np.int64), ranges


def _creategrid(elemtocoord, xi):
    """Tri/bi-linear blend of the element corners (Metrics.jl:46-114).  Returns
    x (nelem, Np, 3) (unused components zero)."""
    nelem, nvert, dc = elemtocoord.shape
    d = len(xi)
    Nq = [len(x) for x in xi]
    Np = int(np.prod(Nq))
    ci = np.stack(np.unravel_index(np.arange(Np), Nq, order="F"), axis=1)
    r = [xi[j][ci[:, j]] for j in range(d)]
    x = np.zeros((nelem, Np, 3))
    e2c = np.asarray(elemtocoord, dtype=np.float64)
    for n in range(dc):
        acc = np.zeros((nelem, Np))
        for v in range(nvert):
            w = np.ones(Np)
            for j in range(d):
                w = w * ((1 + r[j]) if (v >> j) & 1 else (1 - r[j]))
            acc = acc + w[None, :] * e2c[:, v, n][:, None]
        x[:, :, n] = acc / (2 ** d)
    return x


def _apply_D(D, u, axis, Nq):
    """out[..., i, ...] = sum_n D[i, n] u[..., n, ...] along tensor axis."""
    shp = u.shape
    v = u.reshape(shp[0], *Nq[::-1])            # (nelem, k, j, i)
    ax = len(Nq) - axis                          # numpy axis of tensor dim `axis`
    out = np.zeros_like(v)
    n_ax = Nq[axis]
    vm = np.moveaxis(v, ax, -1)
    om = np.moveaxis(out, ax, -1)
    for n in range(n_ax):                        # sequential sum like the reference
        om += vm[..., n:n + 1] * D[:, n]
    return out.reshape(shp)


def computegeometry(elemtocoord, D, xi, omega, meshwarp=None):
    """Returns ``(vgeo (nelem,25,Np), sgeo (nelem,nface,maxNfp,5))``.
    Reference: Grids.jl:1028-1124 and Metrics.jl (see module docstring)."""
    d = len(D)
    nelem = elemtocoord.shape[0]
    Nq = [Dj.shape[0] for Dj in D]
    assert all(q > 1 for q in Nq), "FV (N = 0) geometry is out of scope"
    Np = int(np.prod(Nq))
    Nfp = [Np // q for q in Nq]
    nface = 2 * d
    maxfp = max(Nfp)
    vgeo = np.zeros((nelem, NVGEO, Np))
    sgeo = np.zeros((nelem, nface, maxfp, NSGEO))
    x = _creategrid(elemtocoord, xi)
    x1, x2, x3 = x[:, :, 0], x[:, :, 1], x[:, :, 2]
    if meshwarp is not None:
        x1, x2, x3 = meshwarp(x1, x2, x3)
    vgeo[:, _x1], vgeo[:, _x2], vgeo[:, _x3] = x1, x2, x3
    X = (x1, x2, x3)
    # c) d x_a / d xi_b
    xr = [[_apply_D(D[b], X[a], b, Nq) if a < max(d, 3) else None for b in range(d)]
          for a in range(3)]
    fmask = _fmask(Nq)
    if d == 3:
        x1r, x1s, x1t = xr[0]
        x2r, x2s, x2t = xr[1]
        x3r, x3s, x3t = xr[2]
        JcV = np.sqrt(x1t ** 2 + x2t ** 2 + x3t ** 2)
        J = (x1r * (x2s * x3t - x3s * x2t) + x2r * (x3s * x1t - x1s * x3t)
             + x3r * (x1s * x2t - x2s * x1t))
        JI2 = 1 / (2 * J)
        yzr, yzs, yzt = x2 * x3r - x3 * x2r, x2 * x3s - x3 * x2s, x2 * x3t - x3 * x2t
        zxr, zxs, zxt = x3 * x1r - x1 * x3r, x3 * x1s - x1 * x3s, x3 * x1t - x1 * x3t
        xyr, xys, xyt = x1 * x2r - x2 * x1r, x1 * x2s - x2 * x1s, x1 * x2t - x2 * x1t
        A = lambda b, u: _apply_D(D[b], u, b, Nq)
        # curl-invariant form (Metrics.jl:543-576), same accumulation order
        xi2x1 = -A(0, yzt) + A(2, yzr)
        xi3x1 = A(0, yzs) - A(1, yzr)
        xi2x2 = -A(0, zxt) + A(2, zxr)
        xi3x2 = A(0, zxs) - A(1, zxr)
        xi2x3 = -A(0, xyt) + A(2, xyr)
        xi3x3 = A(0, xys) - A(1, xyr)
        xi1x1 = A(1, yzt) - A(2, yzs)
        xi1x2 = A(1, zxt) - A(2, zxs)
        xi1x3 = A(1, xyt) - A(2, xys)
        (xi1x1, xi2x1, xi3x1, xi1x2, xi2x2, xi3x2, xi1x3, xi2x3, xi3x3) = [
            u * JI2 for u in (xi1x1, xi2x1, xi3x1, xi1x2, xi2x2, xi3x2,
                              xi1x3, xi2x3, xi3x3)]
        # inverse of d xi/dx for the stored dx/dxi (Metrics.jl:581-667)
        a11 = xi2x2 * xi3x3 - xi2x3 * xi3x2
        a12 = xi1x3 * xi3x2 - xi1x2 * xi3x3
        a13 = xi1x2 * xi2x3 - xi1x3 * xi2x2
        a21 = xi2x3 * xi3x1 - xi2x1 * xi3x3
        a22 = xi1x1 * xi3x3 - xi1x3 * xi3x1
        a23 = xi1x3 * xi2x1 - xi1x1 * xi2x3
        a31 = xi2x1 * xi3x2 - xi2x2 * xi3x1
        a32 = xi1x2 * xi3x1 - xi1x1 * xi3x2
        a33 = xi1x1 * xi2x2 - xi1x2 * xi2x1
        det = xi1x1 * a11 + xi2x1 * a12 + xi3x1 * a13
        inv = 1.0 / det
        vgeo[:, _x1xi1] = inv * (a11 * a11 + a12 * a12 + a13 * a13)
        vgeo[:, _x1xi2] = inv * (a11 * a21 + a12 * a22 + a13 * a23)
        vgeo[:, _x1xi3] = inv * (a11 * a31 + a12 * a32 + a13 * a33)
        vgeo[:, _x2xi1] = inv * (a21 * a11 + a22 * a12 + a23 * a13)
        vgeo[:, _x2xi2] = inv * (a21 * a21 + a22 * a22 + a23 * a23)
        vgeo[:, _x2xi3] = inv * (a21 * a31 + a22 * a32 + a23 * a33)
        vgeo[:, _x3xi1] = inv * (a31 * a11 + a32 * a12 + a33 * a13)
        vgeo[:, _x3xi2] = inv * (a31 * a21 + a32 * a22 + a33 * a23)
        vgeo[:, _x3xi3] = inv * (a31 * a31 + a32 * a32 + a33 * a33)
        mets = {0: (xi1x1, xi1x2, xi1x3), 1: (xi2x1, xi2x2, xi2x3),
                2: (xi3x1, xi3x2, xi3x3)}
        for col, u in zip((_xi1x1, _xi2x1, _xi3x1, _xi1x2, _xi2x2, _xi3x2,
                           _xi1x3, _xi2x3, _xi3x3),
                          (xi1x1, xi2x1, xi3x1, xi1x2, xi2x2, xi3x2,
                           xi1x3, xi2x3, xi3x3)):
            vgeo[:, col] = u
    elif d == 2:
        x1r, x1s = xr[0]
        x2r, x2s = xr[1]
        JcV = np.hypot(x1s, x2s)
        J = x1r * x2s - x2r * x1s
        xi1x1, xi2x1 = x2s / J, -x2r / J
        xi1x2, xi2x2 = -x1s / J, x1r / J
        z = np.zeros_like(J)
        vgeo[:, _xi1x1], vgeo[:, _xi2x1] = xi1x1, xi2x1
        vgeo[:, _xi1x2], vgeo[:, _xi2x2] = xi1x2, xi2x2
        vgeo[:, _x1xi1], vgeo[:, _x2xi1] = x1r, x2r
        vgeo[:, _x1xi2], vgeo[:, _x2xi2] = x1s, x2s
        mets = {0: (xi1x1, xi1x2, z), 1: (xi2x1, xi2x2, z)}
    else:
        raise NotImplementedError("1-D grids are out of scope")
    vgeo[:, _JcV] = JcV
    # surface normals / surface Jacobian (Metrics.jl:370-396, :670-718)
    sgeo[:] = np.nan
    for f in range(nface):
        dd = f // 2
        sign = -1.0 if f % 2 == 0 else 1.0
        fm = fmask[f]
        nfp = Nfp[dd]
        nn = [sign * J[:, fm] * m[:, fm] for m in mets[dd]]
        if d == 3:
            sJ = np.sqrt(nn[0] ** 2 + nn[1] ** 2 + nn[2] ** 2)
        else:
            sJ = np.hypot(nn[0], nn[1])
        sgeo[:, f, :nfp, _n1] = nn[0] / sJ
        sgeo[:, f, :nfp, _n2] = nn[1] / sJ
        sgeo[:, f, :nfp, _n3] = nn[2] / sJ if d == 3 else 0.0
        sgeo[:, f, :nfp, _sM] = sJ
    # mass terms (Grids.jl:1094-1114): xi1 fastest => kron of reversed weights
    # kron(1, reverse(w)...) == w_d (x) ... (x) w_1 with xi1 fastest
    Mfull = omega[0].copy()
    for j in range(1, d):
        Mfull = np.kron(omega[j], Mfull)
    vgeo[:, _M] = J * Mfull[None, :]
    vgeo[:, _MI] = 1.0 / vgeo[:, _M]
    for f in range(nface):
        dd = f // 2
        sgeo[:, f, :Nfp[dd], _vMI] = vgeo[:, _MI][:, fmask[f]]
        others = [j for j in range(d) if j != dd]
        if d > 1:
            wf = omega[others[0]].copy()
            for j in others[1:]:
                wf = np.kron(omega[j], wf)
        else:
            wf = np.ones(1)
        sgeo[:, f, :Nfp[dd], _sM] *= wf[None, :]
    # horizontal mass matrix (Grids.jl:1135-1154)
    MHw = omega[0].copy()
    for j in range(1, d - 1):
        MHw = np.kron(omega[j], MHw)
    MH = np.kron(np.ones(Nq[d - 1]), MHw)
    Jn = vgeo[:, _M] / Mfull[None, :]
    vm = mets[d - 1]
    if d == 3:
        vgeo[:, _MH] = MH[None, :] * np.sqrt((Jn * vm[0]) ** 2 + (Jn * vm[1]) ** 2
                                             + (Jn * vm[2]) ** 2)
    else:
        vgeo[:, _MH] = MH[None, :] * np.hypot(Jn * vm[0], Jn * vm[1])
    return vgeo, sgeo


def indefinite_integral_interpolation_matrix(r, omega):
    """Reference: Grids.jl:1184-1207.  ``I @ f(r)`` is the indefinite integral (from ``r[0]``)
    of the interpolant of ``f``, evaluated at the points ``r``."""
    r = np.asarray(r, dtype=np.float64)
    omega = np.asarray(omega, dtype=np.float64)
    Nq = len(r)
    I = np.zeros((Nq, Nq))
    I[0, :] = omega[0] if Nq == 1 else 0.0
    wbary = elements.baryweights(r)
    for n in range(1, Nq):
        rdst = (1 - r) / 2 * r[0] + (1 + r) / 2 * r[n]
        In = elements.interpolationmatrix(r, rdst, wbary)
        delta = (r[n] - r[0]) / 2
        I[n, :] = delta * (omega @ In)
    return I


class DiscontinuousSpectralElementGrid:
    """Reference: Grids.jl:170-413.  ``polynomialorder`` is an int or a tuple
    (a 2-tuple in 3-D means (horizontal, vertical))."""

    def __init__(self, topology, polynomialorder, meshwarp=None):
        dim = topology.dim
        if isinstance(polynomialorder, int):
            N = (polynomialorder,) * dim
        elif len(polynomialorder) == 2 and dim == 3:
            N = (polynomialorder[0], polynomialorder[0], polynomialorder[1])
        else:
            N = tuple(polynomialorder)
        assert len(N) == dim
        self.topology = topology
        self.dim = dim
        self.N = N
        self.Nq = tuple(n + 1 for n in N)
        self.Np = int(np.prod(self.Nq))
        self.Nfp = tuple(self.Np // q for q in self.Nq)
        self.nface = 2 * dim
        t = topology
        self.vmapM, self.vmapP = mappings(N, t.elemtoelem, t.elemtoface, t.elemtoordr)
        ghostidx = np.arange(t.nreal + 1, t.nelem + 1, dtype=np.int64)
        self.vmaprecv, self.nabrtovmaprecv = commmapping(
            N, ghostidx, t.ghostfaces, t.nabrtorecv)
        self.vmapsend, self.nabrtovmapsend = commmapping(
            N, t.sendelems, t.sendfaces, t.nabrtosend)
        xw = [elements.lglpoints(n) for n in N]
        self.xi = [p[0] for p in xw]
        self.omega = [p[1] for p in xw]
        self.D = [elements.spectralderivative(x) for x in self.xi]
        self.Imat = [indefinite_integral_interpolation_matrix(x, w)
                     for x, w in zip(self.xi, self.omega)]
        self.vgeo, self.sgeo = computegeometry(t.elemtocoord, self.D, self.xi,
                                               self.omega, meshwarp)
        act = np.zeros(self.Np * t.nelem, dtype=bool)
        act[: self.Np * t.nreal] = True
        if len(self.vmaprecv):
            act[self.vmaprecv - 1] = True
        self.activedofs = act
        self.elemtobndy = np.ascontiguousarray(t.elemtobndy, dtype=np.int64)
        self.interiorelems = t.interiorelems
        self.exteriorelems = t.exteriorelems
        self.nabrtorank = t.nabrtorank

    @property
    def nelem(self):
        return self.topology.nelem

    @property
    def nreal(self):
        return self.topology.nreal


def min_node_distance(grid, direction=0):
    """Minimum physical distance between neighbouring nodes along the reference
    directions selected by ``direction`` (0 every, 1 horizontal, 2 vertical) over the real
    elements.  Reference: Grids.jl:444-487, kernel_min_neighbor_distance! :1228-1334.
    (The reference all-reduces with ``min`` over ranks; this is the local part.)"""
    d = grid.dim
    Nq = list(grid.Nq)
    x = np.stack([grid.vgeo[:grid.nreal, c, :] for c in (_x1, _x2, _x3)], axis=-1)
    x = x.reshape(grid.nreal, *Nq[::-1], 3)          # (e, k, j, i, 3)
    use = {0: [True] * d, 1: [True] * (d - 1) + [False], 2: [False] * (d - 1) + [True]}[direction]
    md = np.inf
    for ax in range(d):
        if not use[ax]:
            continue
        npax = d - ax                                   # numpy axis of tensor dim ax
        diff = np.diff(x, axis=npax)
        md = min(md, float(np.sqrt((diff ** 2).sum(axis=-1)).min()))
    return md


def auxiliary_field_gradient(grid, a, direction=0):
    """``auxiliary_field_gradient!`` (kernel ``dgsem_auxiliary_field_gradient!``,
    DGModel_kernels.jl:3097-3232): the element-local strong-form gradient of a nodal field
    ``a`` (nelem, Np), no face terms.  ``direction`` 0 = Every (the horizontal launch, then the
    vertical one incrementing), 1 = Horizontal (xi_1, xi_2 terms), 2 = Vertical (xi_3 term).
    Returns (nelem, 3, Np).  One-time host work (orientation gradient, reference-state setup)."""
    vg, Nq = grid.vgeo, list(grid.Nq)
    out = np.zeros((a.shape[0], 3, a.shape[1]))
    if direction in (0, 1):
        d1, d2 = _apply_D(grid.D[0], a, 0, Nq), _apply_D(grid.D[1], a, 1, Nq)
        for d, (c1, c2) in enumerate(((_xi1x1, _xi2x1), (_xi1x2, _xi2x2), (_xi1x3, _xi2x3))):
            out[:, d, :] = vg[:, c1, :] * d1 + vg[:, c2, :] * d2
    if direction in (0, 2):
        d3 = _apply_D(grid.D[2], a, 2, Nq)
        for d, c3 in enumerate((_xi3x1, _xi3x2, _xi3x3)):
            out[:, d, :] = out[:, d, :] + vg[:, c3, :] * d3
    return out
