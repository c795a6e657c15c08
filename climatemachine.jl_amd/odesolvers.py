"""Host-side mirror of the reference's explicit low-storage Runge-Kutta driver.

Reference: ``LowStorageRungeKutta2N`` / ``dostep!`` / ``update!``
``src/Numerics/ODESolvers/LowStorageRungeKuttaMethod.jl:26-62,102-158``;
``LSRK54CarpenterKennedy`` ``:293-327`` (rational coefficients converted to Float64),
``LSRK144NiegemannDiehlBusch`` ``:349-410``;
``solve!`` / ``general_dostep!`` ``ODESolvers.jl:49-158``;
``StrongStabilityPreservingRungeKutta`` and its four tableaus
``StrongStabilityPreservingRungeKuttaMethod.jl:27-285``.

The stage loop itself runs inside libcmdg (``cmdg_lsrk_run``): five fused
RHS+update passes per step, enqueued without host synchronisation.
"""
from fractions import Fraction

__all__ = ["LSRK54CarpenterKennedy", "LSRK144NiegemannDiehlBusch", "solve",
           "LowStorageRungeKutta2N", "LSRK144_COEFFICIENTS", "StrongStabilityPreservingRungeKutta",
           "SSPRK22Heuns", "SSPRK22Ralstons", "SSPRK33ShuOsher", "SSPRK34SpiteriRuuth",
           "SSPRK_COEFFICIENTS", "LowStorageRungeKutta3N", "LS3NRK44Classic", "LS3NRK33Heuns",
           "LS3N_COEFFICIENTS"]


def _f(num, den):
    return float(Fraction(num, den))


def _advance(solver, nsteps, dt):
    # updatetime!: t += dt once per step (the running sum, not t0 + n * dt)
    for _ in range(int(nsteps)):
        solver.t += dt
    solver.steps += int(nsteps)


class LowStorageRungeKutta2N:
    def __init__(self, dg, RKA, RKB, RKC, Q, dt=0.0, t0=0.0):
        self.dg = dg
        self.RKA, self.RKB, self.RKC = tuple(RKA), tuple(RKB), tuple(RKC)
        self.dt, self.t = dt, t0
        self.steps = 0
        self.dQ = dg.create_state(Q.shape[1])          # zero initialised (:52-53)

    def dostep(self, Q, nsteps=1, dt=None):
        """``nsteps`` steps of size ``dt`` from ``self.t``; the solver's time and step count
        advance as ``general_dostep!`` / ``updatetime!`` do (ODESolvers.jl:56-73, 96-98)."""
        dt = self.dt if dt is None else dt
        self.dg.lsrk_run(Q, self.dQ, self.t, dt, nsteps, self.RKA, self.RKB, self.RKC)
        _advance(self, nsteps, dt)


def LSRK54CarpenterKennedy(dg, Q, dt=0.0, t0=0.0):
    RKA = (0.0, _f(-567301805773, 1357537059087), _f(-2404267990393, 2016746695238),
           _f(-3550918686646, 2091501179385), _f(-1275806237668, 842570457699))
    RKB = (_f(1432997174477, 9575080441755), _f(5161836677717, 13612068292357),
           _f(1720146321549, 2090206949498), _f(3134564353537, 4481467310338),
           _f(2277821191437, 14882151754819))
    RKC = (0.0, _f(1432997174477, 9575080441755), _f(2526269341429, 6820363962896),
           _f(2006345519317, 3224310063776), _f(2802321613138, 2924317926251))
    return LowStorageRungeKutta2N(dg, RKA, RKB, RKC, Q, dt=dt, t0=t0)


# the published 14-stage, 4th-order coefficients (Niegemann, Diehl & Busch 2012), as tabulated
# in LowStorageRungeKuttaMethod.jl:358-407
LSRK144_COEFFICIENTS = (
    (0.0, -0.7188012108672410, -0.7785331173421570, -0.0053282796654044, -0.8552979934029281,
     -3.9564138245774565, -1.5780575380587385, -2.0837094552574054, -0.7483334182761610,
     -0.7032861106563359, 0.0013917096117681, -0.0932075369637460, -0.9514200470875948,
     -7.1151571693922548),
    (0.0367762454319673, 0.3136296607553959, 0.1531848691869027, 0.0030097086818182,
     0.3326293790646110, 0.2440251405350864, 0.3718879239592277, 0.6204126221582444,
     0.1524043173028741, 0.0760894927419266, 0.0077604214040978, 0.0024647284755382,
     0.0780348340049386, 5.5059777270269628),
    (0.0, 0.0367762454319673, 0.1249685262725025, 0.2446177702277698, 0.2476149531070420,
     0.2969311120382472, 0.3978149645802642, 0.5270854589440328, 0.6981269994175695,
     0.8190890835352128, 0.8527059887098624, 0.8604711817462826, 0.8627060376969976,
     0.8734213127600976),
)


def LSRK144NiegemannDiehlBusch(dg, Q, dt=0.0, t0=0.0):
    RKA, RKB, RKC = LSRK144_COEFFICIENTS
    return LowStorageRungeKutta2N(dg, RKA, RKB, RKC, Q, dt=dt, t0=t0)


# (RKA rows, RKB, RKC) of StrongStabilityPreservingRungeKuttaMethod.jl:203-285: Heun, Ralston,
# Shu & Osher (1988) three-stage third-order, Spiteri & Ruuth (2002) four-stage third-order
SSPRK_COEFFICIENTS = {
    "SSPRK22Heuns": (((1.0, 0.0), (1 / 2, 1 / 2)), (1.0, 1 / 2), (0.0, 1.0)),
    "SSPRK22Ralstons": (((1.0, 0.0), (5 / 8, 3 / 8)), (_f(2, 3), 3 / 4), (0.0, _f(2, 3))),
    "SSPRK33ShuOsher": (((1.0, 0.0), (3 / 4, 1 / 4), (_f(1, 3), _f(2, 3))),
                        (1.0, 1 / 4, _f(2, 3)), (0.0, 1.0, 1 / 2)),
    "SSPRK34SpiteriRuuth": (((1.0, 0.0), (0.0, 1.0), (_f(2, 3), _f(1, 3)), (0.0, 1.0)),
                            (1 / 2, 1 / 2, _f(1, 6), 1 / 2), (0.0, 1 / 2, 1.0, 1 / 2)),
}


class StrongStabilityPreservingRungeKutta:
    """``StrongStabilityPreservingRungeKutta(f, RKA, RKB, RKC, Q; dt, t0)``
    (StrongStabilityPreservingRungeKuttaMethod.jl:27-75); the stage loop is ``cmdg_ssprk_step``."""

    def __init__(self, dg, RKA, RKB, RKC, Q, dt=0.0, t0=0.0):
        import numpy as np
        self.dg, self.dt, self.t, self.steps = dg, dt, t0, 0
        self.RKA = np.ascontiguousarray(RKA, dtype=np.float64)
        self.RKB = np.ascontiguousarray(RKB, dtype=np.float64)
        self.RKC = np.ascontiguousarray(RKC, dtype=np.float64)
        self.Rstage = dg.create_state(Q.shape[1])
        self.Qstage = dg.create_state(Q.shape[1])

    def dostep(self, Q, nsteps=1, dt=None):
        import ctypes as C
        from . import _lib
        dt = self.dt if dt is None else dt
        p = lambda a: C.c_void_p(a.ctypes.data)
        self.dg._torch_ready()
        for i in range(int(nsteps)):     # each step starts at the running sum, as cmdg_lsrk_run's do
            _lib.check(self.dg.L.cmdg_ssprk_step(
                self.dg.handle, Q.data_ptr(), self.Rstage.data_ptr(), self.Qstage.data_ptr(),
                float(self.t), float(dt), len(self.RKB), p(self.RKA), p(self.RKB),
                p(self.RKC)), self.dg.handle)
            _advance(self, 1, dt)


def _ssp(name):
    def make(dg, Q, dt=0.0, t0=0.0):
        return StrongStabilityPreservingRungeKutta(dg, *SSPRK_COEFFICIENTS[name], Q, dt=dt, t0=t0)
    make.__name__ = name
    return make


SSPRK22Heuns, SSPRK22Ralstons = _ssp("SSPRK22Heuns"), _ssp("SSPRK22Ralstons")
SSPRK33ShuOsher, SSPRK34SpiteriRuuth = _ssp("SSPRK33ShuOsher"), _ssp("SSPRK34SpiteriRuuth")


# (RKA, RKB, RKC) of LowStorageRungeKutta3NMethod.jl:228-345: the classic fourth-order scheme and
# Heun's third-order scheme in Fyfe's 3N storage form
LS3N_COEFFICIENTS = {
    "LS3NRK44Classic": (((0.0, 0.0), (0.0, 1.0), (-1 / 2, 0.0), (2.0, -6.0)),
                        ((1 / 2, 0.0), (1 / 2, -1 / 2), (1.0, 0.0), (_f(1, 6), _f(1, 6))),
                        (0.0, 1 / 2, 1 / 2, 1.0)),
    "LS3NRK33Heuns": (((0.0, 0.0), (0.0, 1.0), (-1.0, _f(1, 3))),
                      ((_f(1, 3), 0.0), (_f(2, 3), -_f(1, 3)), (3 / 4, 1 / 4)),
                      (0.0, _f(1, 3), _f(2, 3))),
}


class LowStorageRungeKutta3N:
    """``LowStorageRungeKutta3N(f, RKA, RKB, RKC, RKW, Q; dt, t0)``
    (LowStorageRungeKutta3NMethod.jl:60-120); the stage loop is ``cmdg_ls3n_step``."""

    def __init__(self, dg, RKA, RKB, RKC, Q, dt=0.0, t0=0.0):
        import numpy as np
        self.dg, self.dt, self.t, self.steps = dg, dt, t0, 0
        self.RKA = np.ascontiguousarray(RKA, dtype=np.float64)
        self.RKB = np.ascontiguousarray(RKB, dtype=np.float64)
        self.RKC = np.ascontiguousarray(RKC, dtype=np.float64)
        self.dQ = dg.create_state(Q.shape[1])
        self.dR = dg.create_state(Q.shape[1])

    def dostep(self, Q, nsteps=1, dt=None):
        import ctypes as C
        from . import _lib
        dt = self.dt if dt is None else dt
        p = lambda a: C.c_void_p(a.ctypes.data)
        self.dg._torch_ready()
        for i in range(int(nsteps)):
            _lib.check(self.dg.L.cmdg_ls3n_step(
                self.dg.handle, Q.data_ptr(), self.dQ.data_ptr(), self.dR.data_ptr(),
                float(self.t), float(dt), len(self.RKC), p(self.RKA), p(self.RKB),
                p(self.RKC)), self.dg.handle)
            _advance(self, 1, dt)


def _ls3n(name):
    def make(dg, Q, dt=0.0, t0=0.0):
        return LowStorageRungeKutta3N(dg, *LS3N_COEFFICIENTS[name], Q, dt=dt, t0=t0)
    make.__name__ = name
    return make


LS3NRK44Classic, LS3NRK33Heuns = _ls3n("LS3NRK44Classic"), _ls3n("LS3NRK33Heuns")


def solve(Q, solver, timeend=None, numberofsteps=0, adjustfinalstep=True):
    """``solve!(Q, solver; timeend, adjustfinalstep, numberofsteps)`` without callbacks:
    whole steps are batched into one library call, the last (shortened) step is
    issued separately like ``general_dostep!`` does."""
    assert timeend is not None or numberofsteps > 0
    dt = solver.dt
    assert dt > 0
    step = 0
    while (timeend is None or solver.t < timeend):
        if timeend is not None and adjustfinalstep and solver.t + dt > timeend:
            solver.dostep(Q, 1, dt=timeend - solver.t)
            solver.t = timeend
        else:
            solver.dostep(Q, 1)
        step += 1
        if step == numberofsteps:
            break
    solver.dg.synchronize()
    return solver.t
