"""Element filters: host-side mirror of ``ClimateMachine.Mesh.Filters``.

Reference: ``src/Numerics/Mesh/Filters.jl`` -- ``spectral_filter_matrix`` :114-128,
``modified_filter_matrix`` :143-159, ``ExponentialFilter`` :172-210,
``BoydVandevenFilter`` :231-267, ``CutoffFilter`` :275-307,
``MassPreservingCutoffFilter`` :316-347, ``TMARFilter`` :369, targets
``FilterIndices`` :72-100 and ``src/Atmos/Model/filters.jl:4-118``.

The filter matrices are one-time host data (``V diag(sigma) V^-1`` with the orthonormal
Legendre Vandermonde matrix ``V``; the reference takes ``V`` from GaussQuadrature.jl's
``orthonormal_poly``, which evaluates the same three-term recurrence).  Applying a
filter is device work: :func:`apply` goes through ``cmdg_filter_apply`` of libcmdg.
"""
import ctypes as C
import math

import numpy as np

EveryDirection, HorizontalDirection, VerticalDirection = 0, 1, 2   # as balancelaws.py

__all__ = [
    "spectral_filter_matrix", "modified_filter_matrix", "orthonormal_legendre_vandermonde",
    "ExponentialFilter", "BoydVandevenFilter", "CutoffFilter", "MassPreservingCutoffFilter",
    "TMARFilter", "FilterIndices", "AtmosFilterPerturbations",
    "AtmosSpecificFilterPerturbations", "apply", "make_device_filter",
]

FILTER_SPECTRAL, FILTER_MASS_PRESERVING, FILTER_TMAR = 0, 1, 2
TARGET_INDICES, TARGET_ATMOS_PERTURBATIONS, TARGET_ATMOS_SPECIFIC_PERTURBATIONS = 0, 1, 2
MAX_FILTER_STATES = 32


def orthonormal_legendre_vandermonde(r):
    """``V[i, n] = sqrt((2n+1)/2) P_n(r_i)`` by the orthonormal three-term recurrence
    ``sqrt(b_{n+1}) p_{n+1} = x p_n - sqrt(b_n) p_{n-1}``, ``b_n = n^2/(4n^2-1)``."""
    r = np.asarray(r, dtype=np.float64)
    N = len(r) - 1
    V = np.zeros((N + 1, N + 1))
    V[:, 0] = 1.0 / math.sqrt(2.0)
    if N >= 1:
        pm1 = np.zeros_like(r)
        p = V[:, 0].copy()
        sb_prev = 0.0
        for n in range(N):
            sb = math.sqrt((n + 1) ** 2 / (4.0 * (n + 1) ** 2 - 1.0))
            pn = (r * p - sb_prev * pm1) / sb
            V[:, n + 1] = pn
            pm1, p, sb_prev = p, pn, sb
    return V


def _filter_matrix(r, Nc, sigma):
    N = len(r) - 1
    V = orthonormal_legendre_vandermonde(r)
    S = np.ones(N + 1)
    for n in range(Nc, N + 1):
        # Julia: (n - Nc) / (N - Nc); 0/0 = NaN is passed to sigma when Nc == N
        eta = (n - Nc) / (N - Nc) if N != Nc else float("nan")
        S[n] = sigma(eta)
    # V * Diagonal(S) / V   ==  solve(V^T, (V S)^T)^T
    return np.linalg.solve(V.T, (V * S[None, :]).T).T


def spectral_filter_matrix(r, Nc, sigma):
    """Reference: Filters.jl:114-128."""
    N = len(r) - 1
    assert N >= 0 and 0 <= Nc <= N
    return _filter_matrix(r, Nc, sigma)


def modified_filter_matrix(r, Nc, sigma):
    """Reference: Filters.jl:143-159 (identity when ``Nc > N``)."""
    N = len(r) - 1
    assert N >= 0 and 0 <= Nc
    if Nc > N:
        return np.eye(N + 1)
    return _filter_matrix(r, Nc, sigma)


def _expand_Nc(grid, Nc):
    dim = grid.dim
    if isinstance(Nc, (int, np.integer)):
        Nc = (int(Nc),) * dim
    elif len(Nc) == 2 and dim == 3:
        Nc = (Nc[0], Nc[0], Nc[1])
    assert len(Nc) == dim
    assert dim == 2 or grid.N[0] == grid.N[1]
    return tuple(int(n) for n in Nc)


class _SpectralFilter:
    kind = FILTER_SPECTRAL

    def _build(self, grid, Nc, sigma, modified=False):
        f = modified_filter_matrix if modified else spectral_filter_matrix
        if not modified:
            assert all(0 <= nc <= n for nc, n in zip(Nc, grid.N))
        self.filter_matrices = tuple(f(grid.xi[i], Nc[i], sigma) for i in range(grid.dim))


class ExponentialFilter(_SpectralFilter):
    """``ExponentialFilter(grid, Nc=0, s=32, alpha=-log(eps))``: ``sigma(eta) = exp(-alpha eta^s)``."""

    def __init__(self, grid, Nc=0, s=32, alpha=-math.log(np.finfo(np.float64).eps)):
        assert s % 2 == 0
        self._build(grid, _expand_Nc(grid, Nc), lambda eta: math.exp(-alpha * eta ** s))


class BoydVandevenFilter(_SpectralFilter):
    """``BoydVandevenFilter(grid, Nc=0, s=32)`` (Filters.jl:231-267)."""

    def __init__(self, grid, Nc=0, s=32):
        assert s % 2 == 0

        def sigma(eta):
            a = 2 * abs(eta) - 1
            if a == 0:
                chi = 1.0
            elif abs(a) == 1:
                chi = float("inf")
            else:
                chi = math.sqrt(-math.log1p(-a * a) / (a * a))
            x = math.sqrt(s) * chi * a
            return math.erfc(x) / 2

        self._build(grid, _expand_Nc(grid, Nc), sigma)


class CutoffFilter(_SpectralFilter):
    """``CutoffFilter(grid, Nc=polynomialorders(grid))``: zeroes modes ``>= Nc``."""

    def __init__(self, grid, Nc=None):
        Nc = grid.N if Nc is None else Nc
        self._build(grid, _expand_Nc(grid, Nc), lambda eta: 0.0)


class MassPreservingCutoffFilter(_SpectralFilter):
    """Cutoff filter that restores the element average (Filters.jl:316-347, kernel :900-1071)."""
    kind = FILTER_MASS_PRESERVING

    def __init__(self, grid, Nc=None):
        Nc = grid.N if Nc is None else Nc
        self._build(grid, _expand_Nc(grid, Nc), lambda eta: 0.0, modified=True)


class TMARFilter:
    """Truncation-and-mass-aware-rescaling positivity filter (Filters.jl:369, kernel :790-884)."""
    kind = FILTER_TMAR
    filter_matrices = None


# ---- targets ---------------------------------------------------------------------------
class FilterIndices:
    """1-based state indices, ``FilterIndices(1, 3)`` or ``FilterIndices(range(1, 4))``."""
    target_id = TARGET_INDICES

    def __init__(self, *I):
        if len(I) == 1 and not isinstance(I[0], (int, np.integer)):
            I = tuple(I[0])
        self.indices = tuple(int(i) for i in I)

    def aux_offsets(self):
        return (0, 0)


class AtmosFilterPerturbations:
    """Filter ``state - ref_state`` for rho and rho e (src/Atmos/Model/filters.jl:4-48)."""
    target_id = TARGET_ATMOS_PERTURBATIONS

    def __init__(self, atmos):
        self.atmos = atmos
        self.indices = tuple(range(1, atmos.ns + 1))

    def aux_offsets(self):
        o = self.atmos.off_ref
        return (o + 0, o + 3)          # ref_state.rho, ref_state.rho e


class AtmosSpecificFilterPerturbations(AtmosFilterPerturbations):
    """Filter specific quantities ``state / rho`` minus the reference's (filters.jl:50-118)."""
    target_id = TARGET_ATMOS_SPECIFIC_PERTURBATIONS


def _as_target(target, nstate, law=None):
    if hasattr(target, "target_id"):
        return target
    if target is None or target == slice(None) or target == ":":
        return FilterIndices(range(1, nstate + 1))
    if isinstance(target, (tuple, list, range)):
        if len(target) and isinstance(target[0], str):
            names = law.state_names()
            return FilterIndices(*[names.index(t) + 1 for t in target])
        return FilterIndices(*target)
    raise TypeError("unknown filter target %r" % (target,))


# ---- device side ------------------------------------------------------------------------
class CmdgFilterDesc(C.Structure):
    """``cmdg_filter_desc`` of include/cmdg.h."""
    _fields_ = [
        ("kind", C.c_int32), ("target", C.c_int32), ("direction", C.c_int32),
        ("nindices", C.c_int32), ("indices", C.c_int32 * MAX_FILTER_STATES),
        ("aux_ref_rho", C.c_int32), ("aux_ref_rhoe", C.c_int32),
        ("filter_h", C.c_void_p), ("filter_v", C.c_void_p),
    ]


class DeviceFilter:
    """A filter + target + direction bound to one ``DGModel`` (``cmdg_filter_create``)."""

    def __init__(self, dg, filt, target, direction=EveryDirection):
        from .. import _lib
        self.dg = dg
        L = dg.L
        d = CmdgFilterDesc()
        d.kind, d.target, d.direction = filt.kind, target.target_id, int(direction)
        idx = target.indices
        if len(idx) > MAX_FILTER_STATES:
            raise _lib.CmdgError("at most %d filtered states" % MAX_FILTER_STATES)
        d.nindices = len(idx)
        for i, v in enumerate(idx):
            d.indices[i] = v
        d.aux_ref_rho, d.aux_ref_rhoe = target.aux_offsets()
        if filt.filter_matrices is not None:
            # column-major (Nq, Nq): element [i, n] at i + Nq n
            self._fh = np.ascontiguousarray(filt.filter_matrices[0].T, dtype=np.float64)
            self._fv = np.ascontiguousarray(filt.filter_matrices[-1].T, dtype=np.float64)
            d.filter_h, d.filter_v = self._fh.ctypes.data, self._fv.ctypes.data
        h = C.c_void_p()
        _lib.check(L.cmdg_filter_create(dg.handle, C.byref(d), C.byref(h)), dg.handle)
        self.handle = h

    def apply(self, Q):
        from .. import _lib
        self.dg._torch_ready()
        _lib.check(self.dg.L.cmdg_filter_apply(self.dg.handle, self.handle, Q.data_ptr(),
                                               Q.shape[1]), self.dg.handle)

    def close(self):
        if getattr(self, "handle", None) and getattr(self.dg, "handle", None):
            self.dg.L.cmdg_filter_destroy(self.dg.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_device_filter(dg, filt, target=None, direction=EveryDirection, nstate=None):
    ns = dg.balance_law.ns if nstate is None else nstate
    return DeviceFilter(dg, filt, _as_target(target, ns, dg.balance_law), direction)


def apply(Q, target, dg, filt, direction=EveryDirection, state_auxiliary=None):
    """``Filters.apply!(Q, target, grid, filter; direction, state_auxiliary)``
    (Filters.jl:408-421).  ``dg`` stands where the reference passes ``grid``: the library
    handle owns the grid tables and the stream the kernel runs on.  ``state_auxiliary``
    is accepted for signature parity; the handle's auxiliary state is what is read."""
    f = make_device_filter(dg, filt, target, direction, nstate=Q.shape[1])
    try:
        f.apply(Q)
        dg.synchronize()
    finally:
        f.close()
