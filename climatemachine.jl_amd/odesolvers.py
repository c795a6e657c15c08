"""Host-side mirror of the reference's explicit low-storage Runge-Kutta driver.

Reference: ``LowStorageRungeKutta2N`` / ``dostep!`` / ``update!``
``src/Numerics/ODESolvers/LowStorageRungeKuttaMethod.jl:26-62,102-158``;
``LSRK54CarpenterKennedy`` ``:293-327`` (rational coefficients converted to Float64);
``solve!`` / ``general_dostep!`` ``ODESolvers.jl:49-158``.

The stage loop itself runs inside libcmdg (``cmdg_lsrk_run``): five fused
RHS+update passes per step, enqueued without host synchronisation.
"""
from fractions import Fraction

__all__ = ["LSRK54CarpenterKennedy", "solve", "LowStorageRungeKutta2N"]


def _f(num, den):
    return float(Fraction(num, den))


class LowStorageRungeKutta2N:
    def __init__(self, dg, RKA, RKB, RKC, Q, dt=0.0, t0=0.0):
        self.dg = dg
        self.RKA, self.RKB, self.RKC = tuple(RKA), tuple(RKB), tuple(RKC)
        self.dt, self.t = dt, t0
        self.steps = 0
        self.dQ = dg.create_state(Q.shape[1])          # zero initialised (:52-53)

    def dostep(self, Q, nsteps=1, dt=None):
        dt = self.dt if dt is None else dt
        self.dg.lsrk_run(Q, self.dQ, self.t, dt, nsteps, self.RKA, self.RKB, self.RKC)


def LSRK54CarpenterKennedy(dg, Q, dt=0.0, t0=0.0):
    RKA = (0.0, _f(-567301805773, 1357537059087), _f(-2404267990393, 2016746695238),
           _f(-3550918686646, 2091501179385), _f(-1275806237668, 842570457699))
    RKB = (_f(1432997174477, 9575080441755), _f(5161836677717, 13612068292357),
           _f(1720146321549, 2090206949498), _f(3134564353537, 4481467310338),
           _f(2277821191437, 14882151754819))
    RKC = (0.0, _f(1432997174477, 9575080441755), _f(2526269341429, 6820363962896),
           _f(2006345519317, 3224310063776), _f(2802321613138, 2924317926251))
    return LowStorageRungeKutta2N(dg, RKA, RKB, RKC, Q, dt=dt, t0=t0)


def solve(Q, solver, timeend=None, numberofsteps=0, adjustfinalstep=True):
    """``solve!(Q, solver; timeend, adjustfinalstep, numberofsteps)`` without callbacks:
    whole steps are batched into one library call, the last (shortened) step is
    issued separately like ``general_dostep!`` does."""
    assert timeend is not None or numberofsteps > 0
    t, dt = solver.t, solver.dt
    assert dt > 0
    step = 0
    while (timeend is None or t < timeend):
        if timeend is not None and adjustfinalstep and t + dt > timeend:
            solver.t = t
            solver.dostep(Q, 1, dt=timeend - t)
            t = timeend
            step += 1
        else:
            solver.t = t
            solver.dostep(Q, 1)
            t = t + dt
            step += 1
        if step == numberofsteps:
            break
    solver.t = t
    solver.steps = step
    solver.dg.synchronize()
    return t
