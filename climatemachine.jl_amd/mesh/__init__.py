"""Host-side mirror of the reference's ``ClimateMachine.Mesh`` (``src/Numerics/Mesh``):
the data producers of the DG hot path.  Not a GPU workload (one-time, host)."""
from . import brickmesh, elements, filters, grids, topologies
from .grids import DiscontinuousSpectralElementGrid
from .topologies import (BrickTopology, CubedShellTopology, StackedBrickTopology,
                         StackedCubedSphereTopology, equiangular_cubed_sphere_warp)

__all__ = [
    "brickmesh", "elements", "filters", "grids", "topologies",
    "DiscontinuousSpectralElementGrid", "BrickTopology", "StackedBrickTopology",
    "CubedShellTopology", "StackedCubedSphereTopology",
    "equiangular_cubed_sphere_warp",
]
