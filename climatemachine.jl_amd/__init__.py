"""cmdg-mi355x: MI355X-native DG right-hand side + explicit LSRK time stepping for
ClimateMachine-style balance laws.  Import through
``cmdg_loader`` (alias ``climatemachine_jl_amd``)."""
from . import mesh  # noqa: F401

__all__ = ["mesh"]
