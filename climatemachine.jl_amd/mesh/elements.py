"""1-D element operators (LGL points / weights, barycentric differentiation).

Reference: ``src/Numerics/Mesh/Elements.jl`` -- lglpoints :11-14, baryweights
:34-46, spectralderivative :60-82.  The reference obtains the Legendre-Gauss-Lobatto
rule from the third-party GaussQuadrature.jl 0.5.5 (``legendre(T, N+1, both)``,
absent from /root/reference); here the published definition is used directly:
the interior nodes are the roots of P'_N, found by Newton iteration, and
``w_i = 2 / (N (N+1) P_N(x_i)^2)``.
"""
import numpy as np

__all__ = ["lglpoints", "baryweights", "spectralderivative", "interpolationmatrix"]


def _legendre(N, x):
    """P_N(x) and P'_N(x) by the three-term recurrence."""
    p0 = np.ones_like(x)
    if N == 0:
        return p0, np.zeros_like(x)
    p1 = x.copy()
    for n in range(1, N):
        p0, p1 = p1, ((2 * n + 1) * x * p1 - n * p0) / (n + 1)
    with np.errstate(divide="ignore", invalid="ignore"):
        dp = N * (x * p1 - p0) / (x * x - 1)      # unused at x = +-1
    return p1, dp


def lglpoints(N):
    """(N+1)-point Legendre-Gauss-Lobatto nodes and weights on [-1, 1]."""
    assert N >= 1
    x = -np.cos(np.pi * np.arange(N + 1) / N)
    xi = x[1:-1].astype(np.longdouble)
    for _ in range(100):
        # q = P'_N ; q' from Legendre's ODE: (1-x^2) P'' = 2x P' - N(N+1) P
        P, dP = _legendre(N, xi)
        ddP = (2 * xi * dP - N * (N + 1) * P) / (1 - xi * xi)
        dx = dP / ddP
        xi = xi - dx
        if np.max(np.abs(dx), initial=0) < 1e-19:
            break
    x = np.concatenate([[-1.0], np.asarray(xi, dtype=np.float64), [1.0]])
    # enforce exact antisymmetry as the eigen-solver based rule does to rounding
    x = (x - x[::-1]) / 2
    xl = x.astype(np.longdouble)
    P, _ = _legendre(N, xl)
    w = np.asarray(2 / (N * (N + 1) * P * P), dtype=np.float64)
    w = (w + w[::-1]) / 2
    return x, w


def baryweights(r):
    """Reference: Elements.jl:34-46."""
    r = np.asarray(r, dtype=np.float64)
    Np = len(r)
    wb = np.ones(Np)
    for j in range(Np):
        for i in range(Np):
            if i != j:
                wb[j] = wb[j] * (r[j] - r[i])
        wb[j] = 1.0 / wb[j]
    return wb


def spectralderivative(r, wb=None):
    """``D[j, k]`` (row j, column k).  Reference: Elements.jl:60-82."""
    r = np.asarray(r, dtype=np.float64)
    wb = baryweights(r) if wb is None else wb
    Np = len(r)
    D = np.zeros((Np, Np))
    for k in range(Np):
        for j in range(Np):
            if k == j:
                for l in range(Np):
                    if l != k:
                        D[j, k] = D[j, k] + 1.0 / (r[k] - r[l])
            else:
                D[j, k] = (wb[k] / wb[j]) / (r[j] - r[k])
    return D


def interpolationmatrix(rsrc, rdst, wbsrc=None):
    """Reference: Elements.jl:94-116."""
    rsrc = np.asarray(rsrc, dtype=np.float64)
    rdst = np.asarray(rdst, dtype=np.float64)
    wbsrc = baryweights(rsrc) if wbsrc is None else wbsrc
    I = np.zeros((len(rdst), len(rsrc)))
    for k in range(len(rdst)):
        for j in range(len(rsrc)):
            with np.errstate(divide="ignore"):
                I[k, j] = wbsrc[j] / (rdst[k] - rsrc[j])
            if not np.isfinite(I[k, j]):
                I[k, :] = 0
                I[k, j] = 1
                break
        I[k, :] = I[k, :] / I[k, :].sum()
    return I
