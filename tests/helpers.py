"""Shared builders for the tests (configurations of the reference's own tests)."""
import numpy as np

from cmdg_loader import cm

M = cm.mesh
BL = cm.balancelaws


def pseudo1d_setup(Ne=4, N=4, direction=0, flux_bc=False, rank=0, size=1, dim=3):
    """test/Numerics/DGMethods/advection_diffusion/pseudo1D_advection_diffusion.jl:293-368"""
    n = {0: np.ones(3) / np.sqrt(3), 1: np.array([1, 1, 0]) / np.sqrt(2),
         2: np.array([0, 0, 1.0])}[direction]
    rng = [np.linspace(-1, 1, Ne + 1)] * dim
    topl = M.StackedBrickTopology(rng, boundary=((1, 2),) * dim, periodicity=(False,) * dim,
                                  connectivity="full", rank=rank, size=size)
    grid = M.DiscontinuousSpectralElementGrid(topl, N)
    law = BL.AdvectionDiffusion(dim, BL.Pseudo1D(n, 1.0, 1 / 100, -1 / 2, 1 / 10),
                                (BL.InhomogeneousBC(0), BL.InhomogeneousBC(1)), flux_bc=flux_bc)
    dt = (1.0 / 4) / (Ne * N ** 2)
    dt = 1.0 / np.ceil(1.0 / dt)
    return law, grid, dt


def rel_linf(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def periodic_hyperdiffusion_setup(Ne=4, N=4, direction=0, rank=0, size=1):
    """test/Numerics/DGMethods/advection_diffusion/periodic_3D_hyperdiffusion.jl:107-322"""
    from fractions import Fraction as Fr
    D = np.array([[9, 3, 5], [3, 7, 4], [5, 4, 10]], dtype=np.float64) / 50 / 100
    x = np.linspace(0.0, 2 * np.pi, Ne + 1)
    topl = M.StackedBrickTopology([x] * 3, periodicity=(True,) * 3, connectivity="full",
                                  rank=rank, size=size)
    grid = M.DiscontinuousSpectralElementGrid(topl, N)
    law = BL.AdvectionDiffusion(3, BL.ConstantHyperDiffusion(3, direction, D), (),
                                advection=False, diffusion=False, hyperdiffusion=True)
    dx = M.grids.min_node_distance(grid)
    dt = dx ** 4 / 25 / D.sum()
    dt = 1.0 / np.ceil(1.0 / dt)
    return law, grid, dt
