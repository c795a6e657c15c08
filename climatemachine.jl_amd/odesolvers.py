"""Host-side mirror of the reference's explicit low-storage Runge-Kutta driver.

Reference: ``LowStorageRungeKutta2N`` / ``dostep!`` / ``update!``
``src/Numerics/ODESolvers/LowStorageRungeKuttaMethod.jl:26-62,102-158``;
``LSRK54CarpenterKennedy`` ``:293-327`` (rational coefficients converted to Float64),
``LSRK144NiegemannDiehlBusch`` ``:349-410``;
``solve!`` / ``general_dostep!`` ``ODESolvers.jl:49-158``.

The stage loop itself runs inside libcmdg (``cmdg_lsrk_run``): five fused
RHS+update passes per step, enqueued without host synchronisation.
"""
from fractions import Fraction

__all__ = ["LSRK54CarpenterKennedy", "LSRK144NiegemannDiehlBusch", "solve",
           "LowStorageRungeKutta2N", "LSRK144_COEFFICIENTS"]


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
