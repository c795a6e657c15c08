"""Element filters: host filter matrices and the oracle's filter kernels against the
reference's own filter tests (test/Numerics/Mesh/filter.jl).  CPU only."""
import math
import struct

import numpy as np
import pytest

from cmdg_loader import cm

M = cm.mesh
F = cm.mesh.filters
EVERY, HORZ, VERT = 0, 1, 2


def _hex(rows):
    return np.array([[struct.unpack("<d", struct.pack("<Q", w))[0] for w in r] for r in rows])


# filter.jl:20-26 and :51-56 ("Values computed with nodal-dg Filter1D.m")
W_N4_S32 = _hex([
    [0x3fe98f3cd0d725e8, 0x3fddfd863c6c9a44, 0xbfe111110d0fd334, 0x3fddbe357bce0b5c, 0xbfc970267f929618],
    [0x3fb608a150f6f927, 0x3fe99528b1a1cd8d, 0x3fcd41d41f8bae45, 0xbfc987d5fabab8d5, 0x3fb5da1cd858af87],
    [0xbfb333332eb1cd92, 0x3fc666666826f178, 0x3fe999999798faaa, 0x3fc666666826f176, 0xbfb333332eb1cd94],
    [0x3fb5da1cd858af84, 0xbfc987d5fabab8d4, 0x3fcd41d41f8bae46, 0x3fe99528b1a1cd8e, 0x3fb608a150f6f924],
    [0xbfc970267f929618, 0x3fddbe357bce0b5c, 0xbfe111110d0fd333, 0x3fddfd863c6c9a44, 0x3fe98f3cd0d725e8]])
W_N3_NC1_S4 = _hex([
    [0x3fd822e5f54ecb62, 0x3fedd204a0f08ef8, 0xbfc7d3aa58fd6968, 0xbfbf74682ac4d276],
    [0x3fc7db36e726d8c1, 0x3fe59d16feee478b, 0x3fc6745bfbb91e20, 0xbfa30fbb7a645448],
    [0xbfa30fbb7a645455, 0x3fc6745bfbb91e26, 0x3fe59d16feee478a, 0x3fc7db36e726d8c4],
    [0xbfbf74682ac4d280, 0xbfc7d3aa58fd6962, 0x3fedd204a0f08ef7, 0x3fd822e5f54ecb62]])


def _brick(dim, Ne, N, periodic=True, lo=-1.0):
    rng = [np.linspace(lo, 1.0, Ne + 1)] * dim
    topl = M.BrickTopology(rng, periodicity=(periodic,) * dim)
    return M.DiscontinuousSpectralElementGrid(topl, N)


@pytest.mark.parametrize("W,N,Nc,s", [(W_N4_S32, 4, 0, 32), (W_N3_NC1_S4, 3, 1, 4)])
def test_exponential_filter_matrix_golden(W, N, Nc, s):
    """filter.jl:15-76: isapprox against the stored matrices."""
    grid = _brick(2, 1, N)
    filt = F.ExponentialFilter(grid, Nc, s)
    for A in filt.filter_matrices:
        assert np.linalg.norm(A - W) <= np.sqrt(np.finfo(float).eps) * np.linalg.norm(W)
        assert np.abs(A - W).max() < 4e-15       # observed ~5e-16


@pytest.mark.parametrize("cls", ["CutoffFilter", "MassPreservingCutoffFilter"])
def test_cutoff_filter_matrix(cls):
    """filter.jl:78-150: N = (5, 3), Nc = (4, 2); W = V diag(Sigma) V^-1."""
    grid = _brick(2, 1, (5, 3))
    filt = getattr(F, cls)(grid, (4, 2))
    for d, (N, Nc) in enumerate(((5, 4), (3, 2))):
        x = grid.xi[d]
        V = np.polynomial.legendre.legvander(x, N)          # unnormalised: same W
        S = np.ones(N + 1)
        S[Nc:] = 0
        W = V @ np.diag(S) @ np.linalg.inv(V)
        assert np.abs(filt.filter_matrices[d] - W).max() < 1e-13
    # identity branch of modified_filter_matrix (Filters.jl:150)
    assert np.array_equal(F.modified_filter_matrix(grid.xi[0], 6, lambda e: 0.0), np.eye(6))


def test_boyd_vandeven_matrix_properties():
    grid = _brick(2, 1, 4)
    A = F.BoydVandevenFilter(grid, 1, 8).filter_matrices[0]
    x = grid.xi[0]
    # constants and linears (modes < Nc... sigma(0) = 1/2 at Nc) -- mode 0 untouched
    assert np.allclose(A @ np.ones(5), 1.0, atol=1e-14)
    # highest mode is annihilated: sigma(1) = erfc(inf)/2 = 0
    P4 = np.polynomial.legendre.legval(x, [0, 0, 0, 0, 1])
    assert np.abs(A @ P4).max() < 1e-14


# ---- application tests (filter.jl:152-330) ----------------------------------------------
def l0(r): return 1 + 0 * r
def l1(r): return r
def l2(r): return (3 * r ** 2 - 1) / 2
def l3(r): return (5 * r ** 3 - 3 * r) / 2


def low(x, y, z): return l0(x) * l0(y) + 4 * l1(x) * l1(y) + 5 * l1(z) + 6 * l1(z) * l1(x)
def high(x, y, z): return l2(x) * l3(y) + l3(x) + l2(y) + l3(z) * l1(y)


def filtered(direction, dim, x, y, z):
    if direction == EVERY:
        return high(x, y, z)
    if direction == VERT:
        return l2(x) * l3(y) + l2(y) if dim == 2 else l3(z) * l1(y)
    return l2(x) * l3(y) + l3(x) if dim == 2 else l2(x) * l3(y) + l3(x) + l2(y)


def _coords(grid):
    x = grid.vgeo[:, 12, :]
    y = grid.vgeo[:, 13, :]
    z = grid.vgeo[:, 14, :]
    return x, y, z


def _filter_test_state(grid, direction):
    """init_state_prognostic! of FilterTestModel{4} (filter.jl:177-197)."""
    x, y, z = _coords(grid)
    Q = np.repeat((low(x, y, z) + high(x, y, z))[:, None, :], 4, axis=1)
    if direction is not None:
        Q[:, 0, :] -= filtered(direction, grid.dim, x, y, z)
        Q[:, 2, :] -= filtered(direction, grid.dim, x, y, z)
    return np.ascontiguousarray(Q)


@pytest.mark.parametrize("cls", ["CutoffFilter", "MassPreservingCutoffFilter"])
@pytest.mark.parametrize("direction", [EVERY, HORZ, VERT])
@pytest.mark.parametrize("dim", [2, 3])
def test_cutoff_filter_application(oracle, dim, direction, cls):
    grid = _brick(dim, 1, 3)
    filt = getattr(F, cls)(grid, 2)
    Q = _filter_test_state(grid, None)
    oracle.apply_filter(Q, F.FilterIndices(1, 3), grid, filt, direction=direction)
    P = _filter_test_state(grid, direction)
    assert np.abs(Q - P).max() < 2e-13, np.abs(Q - P).max()
    # untouched states are bit-identical
    Q0 = _filter_test_state(grid, None)
    assert np.array_equal(Q[:, 1], Q0[:, 1]) and np.array_equal(Q[:, 3], Q0[:, 3])


def _weightedsum(grid, Q, s):
    return float(np.sum(grid.vgeo[: grid.nreal, 9, :] * Q[: grid.nreal, s, :]))


@pytest.mark.parametrize("target", [(1,), None])
@pytest.mark.parametrize("dim", [2, 3])
def test_tmar_filter_application(oracle, dim, target):
    """filter.jl:349-399."""
    grid = _brick(dim, 2, 4)
    x, _, _ = _coords(grid)
    Q = np.ascontiguousarray((np.abs(x) - 0.1)[:, None, :])
    before = _weightedsum(grid, Q, 0)
    assert Q.min() < 0
    tg = F.FilterIndices(*target) if target else F.FilterIndices(range(1, 2))
    oracle.apply_filter(Q, tg, grid, F.TMARFilter())
    assert Q.min() >= 0
    assert abs(_weightedsum(grid, Q, 0) - before) <= 10 * np.finfo(float).eps * abs(before)


def test_mass_preserving_filter_conserves_on_sphere(oracle):
    """filter.jl:440-509: N = 3, 6 elements, shell 1.0 .. 1.2."""
    topl = M.StackedCubedSphereTopology(1, np.array([1.0, 1.2]), boundary=(5, 6))
    grid = M.DiscontinuousSpectralElementGrid(topl, (3, 3),
                                              meshwarp=M.equiangular_cubed_sphere_warp)
    rt = np.sqrt(np.finfo(float).eps)
    for cls, conserved in (("MassPreservingCutoffFilter", True), ("CutoffFilter", False)):
        filt = getattr(F, cls)(grid, 2)
        Q = _filter_test_state(grid, None)
        before = [_weightedsum(grid, Q, s) for s in range(3)]
        oracle.apply_filter(Q, F.FilterIndices(range(1, 4)), grid, filt)
        after = [_weightedsum(grid, Q, s) for s in range(3)]
        for b, a in zip(before, after):
            close = abs(a - b) <= rt * max(abs(a), abs(b))
            assert close == conserved, (cls, b, a)


def test_atmos_perturbation_targets(oracle):
    """AtmosFilterPerturbations / AtmosSpecificFilterPerturbations (src/Atmos/Model/
    filters.jl): a state equal to the reference state plus low modes is a fixed point of the
    cutoff filter; high modes of the perturbation are removed."""
    from helpers import held_suarez_setup
    law, grid, _, _ = held_suarez_setup(n_horz=2, n_vert=2)
    aux = law.init_state_auxiliary(grid)
    Q0 = law.init_state_prognostic(grid, aux, 0.0)
    filt = F.ExponentialFilter(grid, 0, 20)
    for tcls in (F.AtmosFilterPerturbations, F.AtmosSpecificFilterPerturbations):
        Q = Q0.copy()
        oracle.apply_filter(Q, tcls(law), grid, filt, state_auxiliary=aux)
        assert np.isfinite(Q).all()
        # ghost elements untouched, real ones changed only by a small relative amount
        assert np.array_equal(Q[grid.nreal:], Q0[grid.nreal:])
        rel = np.abs(Q - Q0)[: grid.nreal].max(axis=(0, 2)) / np.abs(Q0).max(axis=(0, 2)).clip(1e-300)
        assert rel[0] < 1e-2 and rel[4] < 1e-2       # momenta start near zero
    # the filter of perturbations is exact for a state == reference state + constant offset
    o = law.off_ref
    Q = np.zeros_like(Q0)
    Q[:, 0] = aux[:, o] + 0.25
    Q[:, 4] = aux[:, o + 3] - 3.0
    Q[:, 1:4] = 1.5
    Q1 = Q.copy()
    oracle.apply_filter(Q1, F.AtmosFilterPerturbations(law), grid, filt, state_auxiliary=aux)
    assert np.abs(Q1 - Q)[: grid.nreal].max() <= 1e-9 * np.abs(Q).max()
