"""cmdg-mi355x: MI355X-native DG right-hand side + explicit LSRK time stepping for
ClimateMachine-style balance laws.  Import through
``cmdg_loader`` (alias ``climatemachine_jl_amd``).

``mesh`` and ``balancelaws`` are host-only (numpy).  ``dgmodel`` / ``odesolvers``
need torch (device memory) and ``libcmdg.so`` (the hand-written HIP kernels); they
are imported lazily so that the host-side pieces work without a GPU."""
from . import atmos, balancelaws, mesh, moist, ocean, ocean01  # noqa: F401

__all__ = ["mesh", "balancelaws", "atmos", "moist", "ocean", "ocean01", "dgmodel", "odesolvers", "plugins"]


def __getattr__(name):
    if name in ("dgmodel", "odesolvers", "_lib", "plugins"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
