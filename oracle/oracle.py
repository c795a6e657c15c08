"""ORACLE -- TEST INFRASTRUCTURE ONLY.

ctypes front-end of ``liboracle.so`` (the C restatement of the reference's DG
kernels) plus the reference's *orchestration* restated in Python:

* ``OracleDGModel.__call__``  follows ``(dg::DGModel)(tendency, Q, _, t, alpha, beta)``
  ``src/Numerics/DGMethods/DGModel.jl:85-427`` and the launchers
  ``SpaceDiscretization.jl:502-1368`` (horizontal kernel then vertical kernel,
  interior then exterior interface launches, halo begin/end in between);
* ``lsrk54_step`` / ``solve`` follow ``LowStorageRungeKuttaMethod.jl:102-158,293-327``
  and ``ODESolvers.jl:49-158``;
* ``euclidean_distance`` / ``norm`` follow ``MPIStateArrays.jl:583-644`` (mass weighted).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  Parity pinned by tests/test_oracle_golden.py against the
reference's stored L2 errors.
"""
import ctypes as C
import os
import subprocess
from fractions import Fraction

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EVERY, HORIZONTAL, VERTICAL = 0, 1, 2
MAXS = 32


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or any(
            os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


class _Physics(C.Structure):
    _fields_ = ([(n, C.c_int) for n in ("ns", "naux", "ngrad", "ngf", "ngl", "nhyp")]
                + [("hv_indexmap", C.c_int * MAXS), ("nf_first", C.c_int), ("p", C.c_void_p)]
                + [(n, C.c_void_p) for n in (
                    "flux_first_order", "flux_second_order", "source", "gradient_argument",
                    "gradient_flux", "post_gradient_laplacian", "wavespeed", "boundary_state",
                    "boundary_flux_second_order", "boundary_state_divergence",
                    "boundary_state_higher_order", "update_aux", "courant",
                    "update_penalty", "numerical_flux_law")])


class _Grid(C.Structure):
    _fields_ = [("dim", C.c_int), ("Nq", C.c_int * 3), ("Np", C.c_int), ("Nfp", C.c_int),
                ("nface", C.c_int), ("nvgeo", C.c_int), ("nelem", C.c_int64),
                ("nreal", C.c_int64), ("vgeo", C.c_void_p), ("sgeo", C.c_void_p),
                ("vmapM", C.c_void_p), ("vmapP", C.c_void_p), ("elemtobndy", C.c_void_p),
                ("D", C.c_void_p * 3)]


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_advdiff_new.restype = C.POINTER(_Physics)
        _LIB.orc_advdiff_new.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _LIB.orc_sw_new.restype = C.POINTER(_Physics)
        _LIB.orc_sw_new.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _LIB.orc_moist_new.restype = C.POINTER(_Physics)
        _LIB.orc_moist_new.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _LIB.orc_moist_saturation_adjustment.restype = C.c_double
        _LIB.orc_moist_saturation_adjustment.argtypes = [C.POINTER(_Physics), C.c_double, C.c_double,
                                                         C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB.orc_pgrad_new.restype = C.POINTER(_Physics)
        _LIB.orc_pgrad_new.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        if hasattr(_LIB, "orc_ocean_new"):
            _LIB.orc_ocean_new.restype = C.POINTER(_Physics)
            _LIB.orc_ocean_new.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        if hasattr(_LIB, "orc_atmos_new"):
            _LIB.orc_atmos_new.restype = C.POINTER(_Physics)
            _LIB.orc_atmos_new.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        for name in ("orc_ocean_se01_new", "orc_conti3d_se01_new", "orc_baro_se01_new"):
            getattr(_LIB, name).restype = C.POINTER(_Physics)
            getattr(_LIB, name).argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _LIB.orc_physics_free.argtypes = [C.POINTER(_Physics)]
        _LIB.orc_get_max_threads.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def get_max_threads():
    return int(lib().orc_get_max_threads())


def first_touch(a):
    """A copy of ``a`` (leading axis = elements) whose pages are first written by the OpenMP
    thread that will work on those elements (``orc_first_touch_copy``): bench.py's cpu_baseline."""
    a = np.ascontiguousarray(a)
    out = np.empty_like(a)
    if a.size:
        lib().orc_first_touch_copy(_p(out), _p(a), C.c_int64(a.shape[0]),
                                   C.c_int64(a.nbytes // a.shape[0]))
    return out


def stream_triad_gbs(n=1 << 27, repeats=3):
    """Achieved GB/s of ``a = b + s c`` over first-touched arrays with the threads in use."""
    import time
    b = first_touch(np.ones((n // 4096, 4096)))
    c = first_touch(np.ones((n // 4096, 4096)))
    a = first_touch(np.zeros((n // 4096, 4096)))
    best = 0.0
    for _ in range(repeats):
        t0 = time.perf_counter()
        lib().orc_stream_triad(_p(a), _p(b), _p(c), C.c_double(0.5), C.c_int64(a.size))
        best = max(best, 24.0 * a.size / (time.perf_counter() - t0) / 1e9)
    return best


class OracleGrid:
    """Keeps the numpy tables alive and exposes the C ``orc_grid``."""

    def __init__(self, grid):
        assert grid.dim == 3, "the oracle restates the 3-D kernels"
        self.grid = grid
        self.vgeo = np.ascontiguousarray(grid.vgeo, dtype=np.float64)
        self.sgeo = np.ascontiguousarray(grid.sgeo, dtype=np.float64)
        self.vmapM = np.ascontiguousarray(grid.vmapM, dtype=np.int64)
        self.vmapP = np.ascontiguousarray(grid.vmapP, dtype=np.int64)
        self.elemtobndy = np.ascontiguousarray(grid.elemtobndy, dtype=np.int64)
        # reference D is (Nq, Nq) column-major with D[i, n]; numpy C-order => transpose
        self.D = [np.ascontiguousarray(d.T, dtype=np.float64) for d in grid.D]
        g = _Grid()
        g.dim = 3
        g.Nq[:] = grid.Nq
        g.Np, g.Nfp, g.nface, g.nvgeo = grid.Np, max(grid.Nfp), 6, self.vgeo.shape[1]
        g.nelem, g.nreal = grid.nelem, grid.nreal
        g.vgeo, g.sgeo = _p(self.vgeo), _p(self.sgeo)
        g.vmapM, g.vmapP, g.elemtobndy = _p(self.vmapM), _p(self.vmapP), _p(self.elemtobndy)
        for i in range(3):
            g.D[i] = self.D[i].ctypes.data
        self.c = g
        self.interior = np.ascontiguousarray(grid.interiorelems, dtype=np.int64)
        self.exterior = np.ascontiguousarray(grid.exteriorelems, dtype=np.int64)
        self.activedofs = np.ascontiguousarray(grid.activedofs, dtype=np.uint8)

    def numa_distribute(self):
        """Re-home the per-element tables by first touch (bench.py's cpu_baseline)."""
        for name in ("vgeo", "sgeo", "vmapM", "vmapP", "elemtobndy"):
            setattr(self, name, first_touch(getattr(self, name)))
        g = self.c
        g.vgeo, g.sgeo = _p(self.vgeo), _p(self.sgeo)
        g.vmapM, g.vmapP, g.elemtobndy = _p(self.vmapM), _p(self.vmapP), _p(self.elemtobndy)


class OraclePhysics:
    def __init__(self, law, nf_first=0):
        ip, dp = law.descriptor()
        self._ip = np.ascontiguousarray(ip, dtype=np.int32)
        self._dp = np.ascontiguousarray(dp, dtype=np.float64)
        ctor = {1: "orc_advdiff_new", 2: "orc_atmos_new", 3: "orc_ocean_new",
                4: "orc_pgrad_new", 5: "orc_sw_new", 6: "orc_moist_new", 7: "orc_ocean_se01_new",
                8: "orc_conti3d_se01_new", 9: "orc_baro_se01_new"}[law.physics_id]
        self.c = getattr(lib(), ctor)(_p(self._ip), _p(self._dp), int(nf_first))
        ph = self.c.contents
        if nf_first >= 2 and not ph.numerical_flux_law:
            raise ValueError("Roe / HLLC numerical fluxes are methods of the dry atmosphere law only")
        self.ns, self.naux, self.ngrad = ph.ns, ph.naux, ph.ngrad
        self.ngf, self.ngl, self.nhyp = ph.ngf, ph.ngl, ph.nhyp
        assert (self.ns, self.naux, self.ngrad, self.ngf, self.ngl, self.nhyp) == (
            law.ns, law.naux, law.ngrad, law.ngradflux, law.ngradlap, law.nhyper)

    def __del__(self):
        try:
            lib().orc_physics_free(self.c)
        except Exception:
            pass


class NoExchange:
    """Single-rank halo: nothing to do."""

    def begin(self, arr, nvar):
        return None

    def end(self, arr, nvar, token):
        return None


class OracleDGModel:
    """``DGModel(balance_law, grid, nf_first, nf_second, nf_gradient; direction,
    diffusion_direction)`` (DGModel.jl:22-65) evaluated with the oracle kernels."""

    def __init__(self, law, grid, nf_first=0, direction=EVERY, diffusion_direction=None,
                 state_auxiliary=None, exchange=None):
        self.law = law
        self.grid = grid
        self.og = OracleGrid(grid)
        self.ph = OraclePhysics(law, nf_first)
        self.direction = direction
        self.diffusion_direction = direction if diffusion_direction is None else diffusion_direction
        ne, Np = grid.nelem, grid.Np
        self.state_auxiliary = (law.init_state_auxiliary(grid) if state_auxiliary is None
                                else state_auxiliary)
        if state_auxiliary is None and getattr(law, "discrete_hydrostatic_balance", False):
            # atmos_init_aux!(::HydrostaticState) step 2 (ref_state.jl:150-175)
            gradp = reference_pressure_gradient(grid, self.state_auxiliary[:, law.off_ref + 1, :])
            law.rebalance_reference_state(grid, self.state_auxiliary, gradp)
        self.state_gradient_flux = np.zeros((ne, max(self.ph.ngf, 0), Np))
        self.Qhypervisc_grad = np.zeros((ne, 3 * self.ph.ngl, Np))
        self.Qhypervisc_div = np.zeros((ne, self.ph.nhyp, Np))
        self.exchange = exchange or NoExchange()
        self.L = lib()
        # (filter, target) pairs: DGModel(...; gradient_filter, tendency_filter) DGModel.jl:44-45
        self.gradient_filter = None
        self.tendency_filter = None
        # law-specific update_auxiliary_state! / update_auxiliary_state_gradient! methods
        # (callables (dg, Q, t, "real" | "ghost")); None = the nodal default / `false`
        self.update_auxiliary_state_hook = None
        self.update_auxiliary_state_gradient_hook = None

    def numa_distribute(self):
        """Re-home every per-element array of the operator by first touch (cpu_baseline)."""
        self.og.numa_distribute()
        for name in ("state_auxiliary", "state_gradient_flux", "Qhypervisc_grad", "Qhypervisc_div"):
            setattr(self, name, first_touch(getattr(self, name)))

    # -- launchers (SpaceDiscretization.jl) ----------------------------------
    def _dirs(self, d):
        return (d in (EVERY, HORIZONTAL)), (d in (EVERY, VERTICAL))

    def _elems(self, surface):
        return self.og.interior if surface == "interior" else self.og.exterior

    def launch_volume_gradients(self, Q, t):
        h, v = self._dirs(self.diffusion_direction)
        a = (self.ph.c, C.byref(self.og.c))
        if h:
            self.L.orc_volume_gradients(*a, HORIZONTAL, _p(Q), _p(self.state_gradient_flux),
                                        _p(self.Qhypervisc_grad), _p(self.state_auxiliary),
                                        C.c_double(t), 0)
        if v:
            self.L.orc_volume_gradients(*a, VERTICAL, _p(Q), _p(self.state_gradient_flux),
                                        _p(self.Qhypervisc_grad), _p(self.state_auxiliary),
                                        C.c_double(t),
                                        int(self.diffusion_direction != VERTICAL))

    def launch_interface_gradients(self, Q, t, surface):
        h, v = self._dirs(self.diffusion_direction)
        el = self._elems(surface)
        for on, d in ((h, HORIZONTAL), (v, VERTICAL)):
            if on:
                self.L.orc_interface_gradients(
                    self.ph.c, C.byref(self.og.c), d, _p(Q), _p(self.state_gradient_flux),
                    _p(self.Qhypervisc_grad), _p(self.state_auxiliary), C.c_double(t),
                    _p(el), C.c_int64(len(el)))

    def launch_volume_divergence_of_gradients(self):
        h, v = self._dirs(self.diffusion_direction)
        if h:
            self.L.orc_volume_divergence_of_gradients(
                self.ph.c, C.byref(self.og.c), HORIZONTAL, _p(self.Qhypervisc_grad),
                _p(self.Qhypervisc_div), 0)
        if v:
            self.L.orc_volume_divergence_of_gradients(
                self.ph.c, C.byref(self.og.c), VERTICAL, _p(self.Qhypervisc_grad),
                _p(self.Qhypervisc_div), int(self.diffusion_direction != VERTICAL))

    def launch_interface_divergence_of_gradients(self, t, surface):
        h, v = self._dirs(self.diffusion_direction)
        el = self._elems(surface)
        for on, d in ((h, HORIZONTAL), (v, VERTICAL)):
            if on:
                self.L.orc_interface_divergence_of_gradients(
                    self.ph.c, C.byref(self.og.c), d, _p(self.Qhypervisc_grad),
                    _p(self.Qhypervisc_div), _p(self.state_auxiliary), C.c_double(t),
                    _p(el), C.c_int64(len(el)))

    def launch_volume_gradients_of_laplacians(self, Q, t):
        h, v = self._dirs(self.diffusion_direction)
        if h:
            self.L.orc_volume_gradients_of_laplacians(
                self.ph.c, C.byref(self.og.c), HORIZONTAL, _p(self.Qhypervisc_grad),
                _p(self.Qhypervisc_div), _p(Q), _p(self.state_auxiliary), C.c_double(t), 0)
        if v:
            self.L.orc_volume_gradients_of_laplacians(
                self.ph.c, C.byref(self.og.c), VERTICAL, _p(self.Qhypervisc_grad),
                _p(self.Qhypervisc_div), _p(Q), _p(self.state_auxiliary), C.c_double(t),
                int(self.diffusion_direction != VERTICAL))

    def launch_interface_gradients_of_laplacians(self, Q, t, surface):
        h, v = self._dirs(self.diffusion_direction)
        el = self._elems(surface)
        for on, d in ((h, HORIZONTAL), (v, VERTICAL)):
            if on:
                self.L.orc_interface_gradients_of_laplacians(
                    self.ph.c, C.byref(self.og.c), d, _p(self.Qhypervisc_grad),
                    _p(self.Qhypervisc_div), _p(Q), _p(self.state_auxiliary), C.c_double(t),
                    _p(el), C.c_int64(len(el)))

    def launch_volume_tendency(self, tendency, Q, t, alpha, beta):
        h, v = self._dirs(self.direction)
        a = (self.ph.c, C.byref(self.og.c), self.direction)
        b = (_p(tendency), _p(Q), _p(self.state_gradient_flux), _p(self.Qhypervisc_grad),
             _p(self.state_auxiliary), C.c_double(t), C.c_double(alpha))
        if h:
            self.L.orc_volume_tendency(*a, HORIZONTAL, *b, C.c_double(beta),
                                       int(self.direction == HORIZONTAL))
        if v:
            self.L.orc_volume_tendency(*a, VERTICAL, *b,
                                       C.c_double(1.0 if self.direction == EVERY else beta), 1)

    def launch_interface_tendency(self, tendency, Q, t, alpha, surface):
        h, v = self._dirs(self.direction)
        el = self._elems(surface)
        for on, d in ((h, HORIZONTAL), (v, VERTICAL)):
            if on:
                self.L.orc_interface_tendency(
                    self.ph.c, C.byref(self.og.c), d, _p(tendency), _p(Q),
                    _p(self.state_gradient_flux), _p(self.Qhypervisc_grad),
                    _p(self.state_auxiliary), C.c_double(t), _p(el), C.c_int64(len(el)),
                    C.c_double(alpha))

    def update_auxiliary_state(self, Q, t, which):
        if self.update_auxiliary_state_hook is not None:     # a law-specific method
            self.update_auxiliary_state_hook(self, Q, t, which)
            return
        g = self.grid
        e0, e1 = (0, g.nreal) if which == "real" else (g.nreal, g.nelem)
        self.L.orc_update_auxiliary_state(self.ph.c, C.byref(self.og.c), _p(Q),
                                          _p(self.state_auxiliary), C.c_double(t),
                                          C.c_int64(e0), C.c_int64(e1), _p(self.og.activedofs))

    # -- (dg::DGModel)(tendency, Q, _, t, alpha, beta)   DGModel.jl:85-427 ---------
    def __call__(self, tendency, Q, t, alpha=1.0, beta=0.0):
        ph, ex = self.ph, self.exchange
        topo = self.grid.topology
        communicate = not (topo.isstacked and self.direction == VERTICAL)
        self.update_auxiliary_state(Q, t, "real")
        tok_Q = tok_gf = tok_hg = tok_hd = None
        if communicate:
            tok_Q = ex.begin(Q, ph.ns)
        if ph.ngf > 0 or ph.nhyp > 0:
            self.launch_volume_gradients(Q, t)
            self.launch_interface_gradients(Q, t, "interior")
            if communicate:
                ex.end(Q, ph.ns, tok_Q)
                self.update_auxiliary_state(Q, t, "ghost")
            self.launch_interface_gradients(Q, t, "exterior")
            if self.gradient_filter is not None:      # DGModel.jl:185-193
                apply_filter(self.state_gradient_flux, self.gradient_filter[1], self.grid,
                             self.gradient_filter[0])
            if communicate:
                if ph.ngf > 0:
                    tok_gf = ex.begin(self.state_gradient_flux, ph.ngf)
                if ph.nhyp > 0:
                    tok_hg = ex.begin(self.Qhypervisc_grad, 3 * ph.ngl)
            if ph.ngf > 0 and self.update_auxiliary_state_gradient_hook is not None:
                self.update_auxiliary_state_gradient_hook(self, Q, t, "real")   # DGModel.jl:210-222
        if ph.nhyp > 0:
            self.launch_volume_divergence_of_gradients()
            self.launch_interface_divergence_of_gradients(t, "interior")
            if communicate:
                ex.end(self.Qhypervisc_grad, 3 * ph.ngl, tok_hg)
            self.launch_interface_divergence_of_gradients(t, "exterior")
            if communicate:
                tok_hd = ex.begin(self.Qhypervisc_div, ph.nhyp)
            self.launch_volume_gradients_of_laplacians(Q, t)
            self.launch_interface_gradients_of_laplacians(Q, t, "interior")
            if communicate:
                ex.end(self.Qhypervisc_div, ph.nhyp, tok_hd)
            self.launch_interface_gradients_of_laplacians(Q, t, "exterior")
            if communicate:
                tok_hg = ex.begin(self.Qhypervisc_grad, 3 * ph.ngl)
        self.launch_volume_tendency(tendency, Q, t, alpha, beta)
        self.launch_interface_tendency(tendency, Q, t, alpha, "interior")
        if communicate:
            if ph.ngf > 0 or ph.nhyp > 0:
                if ph.ngf > 0:
                    ex.end(self.state_gradient_flux, ph.ngf, tok_gf)
                    if self.update_auxiliary_state_gradient_hook is not None:   # DGModel.jl:355-361
                        self.update_auxiliary_state_gradient_hook(self, Q, t, "ghost")
                if ph.nhyp > 0:
                    ex.end(self.Qhypervisc_grad, 3 * ph.ngl, tok_hg)
            else:
                ex.end(Q, ph.ns, tok_Q)
                self.update_auxiliary_state(Q, t, "ghost")
        self.launch_interface_tendency(tendency, Q, t, alpha, "exterior")
        if self.tendency_filter is not None:          # DGModel.jl:417-425
            apply_filter(tendency, self.tendency_filter[1], self.grid, self.tendency_filter[0])


def reference_pressure_gradient(grid, p):
    """``grad reference_pressure`` (ref_state.jl:235-262): the tendency of the
    PressureGradientModel on ``grid`` with central fluxes, ``(nelem, 3, Np)``; the ghost
    elements carry the analytic pressure, so no exchange is needed."""
    from types import SimpleNamespace
    law = SimpleNamespace(physics_id=4, ns=3, naux=1, ngrad=0, ngradflux=0, ngradlap=0, nhyper=0,
                          descriptor=lambda: (np.zeros(16, dtype=np.int32), np.zeros(32)))
    aux = np.ascontiguousarray(p[:, None, :], dtype=np.float64)
    dg = OracleDGModel(law, grid, nf_first=1, state_auxiliary=aux)
    Q = np.zeros((grid.nelem, 3, grid.Np))
    T = np.zeros_like(Q)
    comm = dg.exchange
    dg.exchange = type("NoComm", (), {"begin": lambda s, a, n: None,
                                      "end": lambda s, a, n, t: None})()
    dg(T, Q, 0.0, 1.0, 0.0)
    dg.exchange = comm
    return T


# ---- LSRK54 Carpenter-Kennedy (LowStorageRungeKuttaMethod.jl:293-327) -------------
def _f(num, den):
    return float(Fraction(num, den))


RKA = (0.0, _f(-567301805773, 1357537059087), _f(-2404267990393, 2016746695238),
       _f(-3550918686646, 2091501179385), _f(-1275806237668, 842570457699))
RKB = (_f(1432997174477, 9575080441755), _f(5161836677717, 13612068292357),
       _f(1720146321549, 2090206949498), _f(3134564353537, 4481467310338),
       _f(2277821191437, 14882151754819))
RKC = (0.0, _f(1432997174477, 9575080441755), _f(2526269341429, 6820363962896),
       _f(2006345519317, 3224310063776), _f(2802321613138, 2924317926251))


def lsrk54_step(dg, Q, dQ, t, dt, step_filter=None):
    lsrk_step(dg, Q, dQ, t, dt, RKA, RKB, RKC, step_filter)


def lsrk_step(dg, Q, dQ, t, dt, rka, rkb, rkc, step_filter=None):
    """``dostep!`` (LowStorageRungeKuttaMethod.jl:102-144): ``rhs!(dQ, Q, p, t + c dt,
    increment = true)`` then ``update!`` on the real elements.  ``step_filter`` =
    ``(filter, target, direction)`` applied to Q after the step (the every-step callback
    of experiments/AtmosGCM/heldsuarez.jl:261-272)."""
    nreal = dg.grid.nreal
    n = nreal * Q.shape[1] * Q.shape[2]
    ns = len(rka)
    for s in range(ns):
        dg(dQ, Q, t + rkc[s] * dt, 1.0, 1.0)
        lib().orc_lsrk_update(_p(dQ), _p(Q), C.c_double(rka[(s + 1) % ns]),
                              C.c_double(rkb[s]), C.c_double(dt), C.c_int64(n))
    if step_filter is not None:
        f, tg, d = step_filter
        apply_filter(Q, tg, dg.grid, f, direction=d, state_auxiliary=dg.state_auxiliary)


def ssprk_step(dg, Q, Rstage, Qstage, t, dt, rka, rkb, rkc):
    """``dostep!`` of StrongStabilityPreservingRungeKuttaMethod.jl:117-165 with the ``update!``
    kernel of :167-190 on the real elements."""
    nr = dg.grid.nreal
    Qstage[:nr] = Q[:nr]
    for s in range(len(rkb)):
        dg(Rstage, Qstage, t + rkc[s] * dt, 1.0, 0.0)     # rhs!(...; increment = false)
        Qstage[:nr] = rka[s][0] * Q[:nr] + rka[s][1] * Qstage[:nr] + dt * rkb[s] * Rstage[:nr]
    Q[:nr] = Qstage[:nr]


def ls3n_step(dg, Q, dQ, dR, t, dt, rka, rkb, rkc):
    """``dostep!`` of LowStorageRungeKutta3NMethod.jl:153-199 with the ``update!`` kernel of
    :201-226 on the real elements."""
    nr, ns = dg.grid.nreal, len(rkc)
    dR[:nr] = -0.0
    for s in range(ns):
        dg(dQ, Q, t + rkc[s] * dt, 1.0, 1.0)             # rhs!(...; increment = true)
        sn = (s + 1) % ns
        Q[:nr] += rkb[s][0] * dt * dQ[:nr] + rkb[s][1] * dt * dR[:nr]
        dR[:nr] += rka[sn][1] * dQ[:nr]
        dQ[:nr] *= rka[sn][0]


def solve(dg, Q, dt, timeend, t0=0.0):
    """``solve!`` with ``adjustfinalstep = true`` (ODESolvers.jl:49-158)."""
    dQ = np.zeros_like(Q)
    t = t0
    nsteps = 0
    while t < timeend:
        step = dt
        final = False
        if t + step > timeend:
            step = timeend - t
            final = True
        lsrk54_step(dg, Q, dQ, t, step)
        t = timeend if final else t + step
        nsteps += 1
    return t, nsteps


def weighted_norm2_local(grid, A, B=None):
    """Local part of ``norm`` / ``euclidean_distance`` squared: mass-weighted sum over
    real elements (MPIStateArrays.jl:583-644, weights = vgeo[:, _M, :])."""
    nreal = grid.nreal
    M = grid.vgeo[:nreal, 9, :][:, None, :]
    d = A[:nreal] if B is None else A[:nreal] - B[:nreal]
    return float(np.sum(M * d * d))


# ---- Courant numbers and time-step selection -------------------------------------------
ADVECTIVE_COURANT, NONDIFFUSIVE_COURANT, DIFFUSIVE_COURANT = 0, 1, 2


def min_neighbor_distance(og, direction=EVERY):
    """Pointwise ``kernel_min_neighbor_distance!`` (Grids.jl:1228-1333): ``(nreal, Np)``."""
    out = np.zeros((og.grid.nreal, og.grid.Np))
    lib().orc_min_neighbor_distance(C.byref(og.c), int(direction), _p(out))
    return out


def courant(kind, dg, Q, dt, simtime=0.0, direction=EVERY):
    """``courant(local_courant, dg, m, Q, dt, simtime, direction)``
    (SpaceDiscretization.jl:307-365): rank-local maximum (the caller Allreduces)."""
    if dg.grid.nreal == 0:
        return -np.inf
    pw = min_neighbor_distance(dg.og, direction)
    gf = dg.state_gradient_flux
    lib().orc_local_courant(dg.ph.c, C.byref(dg.og.c), int(kind), _p(pw), _p(Q),
                            _p(dg.state_auxiliary), _p(gf), C.c_double(dt),
                            C.c_double(simtime), int(direction))
    return float(pw.max())


def calculate_dt(dg, Q, courant_number, t=0.0, direction=EVERY):
    """``calculate_dt(dg, model, Q, Courant_number, t, direction)`` (DGMethods.jl:79-83)."""
    return courant_number / courant(NONDIFFUSIVE_COURANT, dg, Q, 1.0, t, direction)


# ---- column (stack) integrals (integral_oracle.c) ---------------------------------------
class _IntegralLaw(C.Structure):
    _fields_ = [("nout", C.c_int), ("nrout", C.c_int), ("ns", C.c_int), ("naux", C.c_int),
                ("p", C.c_void_p), ("load", C.c_void_p), ("set", C.c_void_p),
                ("rload", C.c_void_p), ("rset", C.c_void_p)]


def integral_test_law():
    """``IntegralTestModel{3}`` of test/Numerics/DGMethods/integral_test.jl."""
    L = lib()
    L.orc_integral_test_law.restype = C.POINTER(_IntegralLaw)
    return L.orc_integral_test_law()


def integral_fields_law(src, scale, dst, rsrc, rdst, ns, naux):
    """Integrands ``scale_s * field_s`` with ``src = [(is_state, column), ...]``; forward
    integrals go to aux columns ``dst``, reverse integrals read ``rsrc`` and write ``rdst``."""
    L = lib()
    L.orc_integral_fields_law.restype = C.POINTER(_IntegralLaw)
    n = len(src)
    ia = lambda v: (C.c_int * n)(*[int(x) for x in v])
    return L.orc_integral_fields_law(n, ia([s[0] for s in src]), ia([s[1] for s in src]),
                                     (C.c_double * n)(*[float(x) for x in scale]), ia(dst),
                                     ia(rsrc), ia(rdst), int(ns), int(naux))


def indefinite_stack_integral(law, og, Q, aux, horzelems=None):
    """``indefinite_stack_integral!(dg, m, Q, state_auxiliary, t, elems)`` (DGModel.jl:445-487)."""
    g = og.grid
    nv = g.topology.stacksize
    h0, h1 = (0, g.nreal // nv) if horzelems is None else horzelems
    Imat = np.ascontiguousarray(np.asarray(g.Imat[-1], dtype=np.float64).T)     # column-major
    Qp = _p(Q) if Q is not None and Q.size else None
    lib().orc_indefinite_stack_integral(law, C.byref(og.c), int(nv), Qp, _p(aux), _p(Imat),
                                        15, C.c_int64(h0), C.c_int64(h1))


def reverse_indefinite_stack_integral(law, og, Q, aux, horzelems=None):
    """``reverse_indefinite_stack_integral!`` (DGModel.jl:489-529)."""
    g = og.grid
    nv = g.topology.stacksize
    h0, h1 = (0, g.nreal // nv) if horzelems is None else horzelems
    Qp = _p(Q) if Q is not None and Q.size else None
    lib().orc_reverse_indefinite_stack_integral(law, C.byref(og.c), int(nv), Qp, _p(aux),
                                                C.c_int64(h0), C.c_int64(h1))


# ---- HydrostaticBoussinesqModel: update_auxiliary_state! / ..._gradient! ------------------
def hydrostatic_boussinesq_hooks(dg, vert_filter, exp_filter):
    """Installs on ``dg`` (an OracleDGModel of the ocean law) the two law methods of
    src/Ocean/HydrostaticBoussinesq/hydrostatic_boussinesq_model.jl:
    ``update_auxiliary_state!`` (:654-680): vertical cutoff filter on u, vertical exponential
    filter on theta (real elements only); ``update_auxiliary_state_gradient!`` (:693-726):
    ``A.w = -D.div_h u``, upward integrals of (w, -alpha_T theta) into (w, pkin), downward
    integral of pkin, and the copy of w at the surface into wz0 of the whole column."""
    from types import SimpleNamespace
    law, grid = dg.law, dg.grid

    class _T:           # FilterIndices
        target_id = 0

        def __init__(self, idx):
            self.indices = tuple(idx)

        def aux_offsets(self):
            return (0, 0)

    ilaw = integral_fields_law([(0, 1), (1, 3)], [1.0, -law.alpha_T], [1, 2], [2, 2], [2, 2],
                               4, 8)
    # the reverse integral touches pkin only (DownwardIntegrals has one variable)
    rlaw = integral_fields_law([(0, 2)], [1.0], [2], [2], [2], 4, 8)
    dg._ocean_keep = SimpleNamespace(ilaw=ilaw, rlaw=rlaw)
    nv = grid.topology.stacksize

    vlaw = integral_fields_law([(0, 0), (0, 1)], [1.0, 1.0], [0, 1], [0, 1], [0, 1], 4, 2)
    dg._ocean_keep.vlaw = vlaw
    dg.integral_aux = np.zeros((grid.nelem, 2, grid.Np))      # VerticalIntegralModel: int_x[2]

    def integrate_velocity(X):
        """update_auxiliary_state!(integral_model, ...) (VerticalIntegralModel.jl:60-81):
        A.int_x = (x.u[1], x.u[2]) then the upward column integral; returns the value at the
        top of every stack, shape (nhorz, 2, Nqh)."""
        ia = dg.integral_aux
        ia[:, 0, :], ia[:, 1, :] = X[:, 0, :], X[:, 1, :]
        indefinite_stack_integral(vlaw, dg.og, X, ia)
        Nqh = grid.Nq[0] * grid.Nq[1]
        return ia.reshape(grid.nelem // nv, nv, 2, grid.Nq[2], Nqh)[:, -1, :, -1, :]

    dg.integrate_velocity = integrate_velocity

    def pre(dgm, Q, t, which):
        if which == "real":
            apply_filter(Q, _T((1, 2)), grid, vert_filter, direction=VERTICAL)
            apply_filter(Q, _T((4,)), grid, exp_filter, direction=VERTICAL)
        if getattr(law, "coupled", False) and which == "real":
            # compute_flow_deviation!(dg, ::HBModel, ::Coupled, Q, t)
            # (HydrostaticBoussinesqCoupling.jl:43-85): u_d = u - (vertical mean of u)
            top = integrate_velocity(Q)
            Nqh = grid.Nq[0] * grid.Nq[1]
            A5 = dgm.state_auxiliary.reshape(grid.nelem // nv, nv, 8, grid.Nq[2], Nqh)
            Q5 = Q.reshape(grid.nelem // nv, nv, 4, grid.Nq[2], Nqh)
            for c in (0, 1):
                A5[:, :, 4 + c] = Q5[:, :, c] - (top[:, c] / law.problem.H)[:, None, None, :]

    def post(dgm, Q, t, which):
        A, D = dgm.state_auxiliary, dgm.state_gradient_flux
        e0, e1 = (0, grid.nreal) if which == "real" else (grid.nreal, grid.nelem)
        if e1 <= e0:
            return
        A[e0:e1, 1, :] = -D[e0:e1, 0, :]
        h = (e0 // nv, e1 // nv)
        indefinite_stack_integral(ilaw, dgm.og, Q, A, horzelems=h)
        reverse_indefinite_stack_integral(rlaw, dgm.og, Q, A, horzelems=h)
        Nqh = grid.Nq[0] * grid.Nq[1]
        data = A.reshape(grid.nelem // nv, nv, A.shape[1], grid.Nq[2], Nqh)
        flat = data[h[0]:h[1], -1, 1, -1, :]                      # w at the top of the stack
        data[h[0]:h[1], :, 3, :, :] = flat[:, None, None, :]

    dg.update_auxiliary_state_hook = pre
    dg.update_auxiliary_state_gradient_hook = post


# ---- SplitExplicitSolver (src/Numerics/ODESolvers/SplitExplicitMethod.jl:70-177) ----------
class SplitExplicitOracle:
    """The split-explicit barotropic / baroclinic stepper restated: slow 3-D HBModel and fast
    2-D ShallowWaterModel (on the one-layer extrusion of the 2-D grid), both LSRK54, with the
    coupling functions of src/Ocean/SplitExplicit/Communication.jl.  ``dg3`` must carry
    ``hydrostatic_boussinesq_hooks`` (for ``integrate_velocity``)."""

    def __init__(self, dg3, dg2, Q3, Q2, dt_slow, dt_fast):
        self.dg3, self.dg2, self.dt, self.dt_fast = dg3, dg2, float(dt_slow), float(dt_fast)
        self.dQ3 = np.zeros_like(Q3)
        self.dQ2fast = np.full_like(Q3, -0.0)
        self.dQ2 = np.zeros_like(Q2)
        self.coupled = bool(dg3.law.coupled)
        g3, g2 = dg3.grid, dg2.grid
        self.nv = g3.topology.stacksize
        self.nh = g3.nelem // self.nv
        self.Nqh = g3.Nq[0] * g3.Nq[1]
        self.Nqk3, self.Nqk2 = g3.Nq[2], g2.Nq[2]
        self.H = dg3.law.problem.H

    # views: 3-D arrays as (nh, nv, nvar, Nqk, Nqh); extruded 2-D arrays as (nh, nvar, Nqk2, Nqh)
    def _v3(self, a):
        return a.reshape(self.nh, self.nv, a.shape[1], self.Nqk3, self.Nqh)

    def _v2(self, a):
        return a.reshape(self.nh, a.shape[1], self.Nqk2, self.Nqh)

    def dostep(self, Q3, Q2, time):
        dg3, dg2, H = self.dg3, self.dg2, self.H
        A3, A2 = dg3.state_auxiliary, dg2.state_auxiliary
        ns = len(RKA)
        for s in range(ns):
            ts = time + RKC[s] * self.dt
            if self.coupled:                       # initialize_states!: A.dG_u = -0
                A3[:, 6:8, :] = -0.0
            dg3(self.dQ2fast, Q3, ts, 1.0, 0.0)    # slow.rhs!(dQ2fast, ...; increment = false)
            if self.coupled:                       # tendency_from_slow_to_fast!
                top = dg3.integrate_velocity(self.dQ2fast)            # (nh, 2, Nqh)
                self._v2(A2)[:, 1:3] = top[:, :, None, :]             # G_U = int du
                self._v3(A3)[:, :, 6:8] -= (top / H)[:, None, :, None, :]
            dg3(self.dQ3, Q3, ts, 1.0, 1.0)        # slow.rhs!(dQslow, ...; increment = true)
            gamma = (1 - RKC[s]) if s == ns - 1 else (RKC[s + 1] - RKC[s])
            nsub = int(np.ceil(gamma * self.dt / self.dt_fast)) if self.dt_fast > 0 else 1
            fdt = gamma * self.dt / nsub
            for sub in range(nsub):
                ft = ts + sub * fdt
                lsrk_step(dg2, Q2, self.dQ2, ft, fdt, RKA, RKB, RKC)
            n = dg3.grid.nreal * Q3.shape[1] * Q3.shape[2]
            lib().orc_lsrk_update(_p(self.dQ3), _p(Q3), C.c_double(RKA[(s + 1) % ns]),
                                  C.c_double(RKB[s]), C.c_double(self.dt), C.c_int64(n))
            if self.coupled:                       # reconcile_from_fast_to_slow!
                top = dg3.integrate_velocity(Q3)
                U = self._v2(Q2)[:, 1:3, 0, :]                         # (nh, 2, Nqh)
                du = 1 / H * (U - top)
                self._v2(A2)[:, 3:5] = du[:, :, None, :]
                self._v3(Q3)[:, :, 0:2] += du[:, None, :, None, :]
                self._v3(Q3)[:, :, 2] = self._v2(Q2)[:, 0, 0, :][:, None, None, :]


# ---- src/Ocean/SplitExplicit01 --------------------------------------------------------------
def ocean01_hooks(dg, conti_dg, vert_filter, exp_filter):
    """``update_auxiliary_state!(dg, ::OceanModel, Q, t, elems)`` of
    src/Ocean/SplitExplicit01/OceanModel.jl:432-541 on ``dg`` (an OracleDGModel of the law):
    vertical cutoff filter on u, exponential filter on theta; ``conti3d_dg`` evaluated on Q and
    its theta tendency copied to A.w; upward integrals (w, -g alpha_T theta) -> (w, pkin),
    downward integral of pkin, w at the surface -> wz0; u_d = u - (1 / H) int u."""
    from types import SimpleNamespace
    law, grid = dg.law, dg.grid
    nv = grid.topology.stacksize
    Nqh = grid.Nq[0] * grid.Nq[1]

    class _T:
        target_id = 0

        def __init__(self, idx):
            self.indices = tuple(idx)

        def aux_offsets(self):
            return (0, 0)

    ilaw = integral_fields_law([(0, 0), (1, 3)], [1.0, -(law.grav * law.alpha_T)], [0, 1], [1, 1],
                               [1, 1], 4, 8)
    rlaw = integral_fields_law([(0, 1)], [1.0], [1], [1], [1], 4, 8)
    vlaw = integral_fields_law([(0, 0), (0, 1)], [1.0, 1.0], [0, 1], [0, 1], [0, 1], 4, 2)
    dg._ocean_keep = SimpleNamespace(ilaw=ilaw, rlaw=rlaw, vlaw=vlaw, conti=conti_dg)
    dg.integral_aux = np.zeros((grid.nelem, 2, grid.Np))
    dg.conti3d_Q = np.zeros((grid.nelem, 4, grid.Np))

    def integrate_velocity(X):
        """FlowIntegralModel / TendencyIntegralModel (VerticalIntegralModel.jl): the upward
        column integral of X's first two columns; value at the top of every stack (nh, 2, Nqh)."""
        ia = dg.integral_aux
        ia[:, 0, :], ia[:, 1, :] = X[:, 0, :], X[:, 1, :]
        indefinite_stack_integral(vlaw, dg.og, X, ia)
        return ia.reshape(grid.nelem // nv, nv, 2, grid.Nq[2], Nqh)[:, -1, :, -1, :]

    dg.integrate_velocity = integrate_velocity

    def pre(dgm, Q, t, which):
        if which != "real":
            return
        A = dgm.state_auxiliary
        apply_filter(Q, _T((1, 2)), grid, vert_filter, direction=VERTICAL)
        apply_filter(Q, _T((4,)), grid, exp_filter, direction=VERTICAL)
        conti_dg(dg.conti3d_Q, Q, t, 1.0, 0.0)            # increment = false
        A[:grid.nreal, 0, :] = dg.conti3d_Q[:grid.nreal, 3, :]
        indefinite_stack_integral(ilaw, dgm.og, Q, A)
        reverse_indefinite_stack_integral(rlaw, dgm.og, Q, A)
        data = A.reshape(grid.nelem // nv, nv, 8, grid.Nq[2], Nqh)
        data[:, :, 2, :, :] = data[:, -1, 0, -1, :][:, None, None, :]       # wz0
        top = integrate_velocity(Q)
        Q5 = Q.reshape(grid.nelem // nv, nv, 4, grid.Nq[2], Nqh)
        for c in (0, 1):                                                   # u_d
            data[:, :, 3 + c] = Q5[:, :, c] - (top[:, c] / law.problem.H)[:, None, None, :]

    dg.update_auxiliary_state_hook = pre


class SplitExplicit01Oracle:
    """``dostep!(Qvec, split::SplitExplicitLSRK2nSolver, param, time)``
    (src/Ocean/SplitExplicit01/SplitExplicitLSRK2nMethod.jl:81-190) with the exchange functions of
    src/Ocean/SplitExplicit01/Communication.jl; ``dg3`` carries ``ocean01_hooks``, ``dg2`` is
    the BarotropicModel on the one-layer extrusion of the 2-D grid, both LSRK54."""

    def __init__(self, dg3, dg2, Q3, Q2, dt_slow, dt_fast):
        self.dg3, self.dg2, self.dt, self.dt_fast = dg3, dg2, float(dt_slow), float(dt_fast)
        self.dQ3 = np.zeros_like(Q3)
        self.dQ2fast = np.full_like(Q3, -0.0)
        self.dQ2 = np.zeros_like(Q2)
        g3, g2 = dg3.grid, dg2.grid
        self.nv = g3.topology.stacksize
        self.nh = g3.nelem // self.nv
        self.Nqh = g3.Nq[0] * g3.Nq[1]
        self.Nqk3, self.Nqk2 = g3.Nq[2], g2.Nq[2]
        self.H = dg3.law.problem.H
        self.add = int(dg3.law.add_fast_substeps)

    def _v3(self, a):
        return a.reshape(self.nh, self.nv, a.shape[1], self.Nqk3, self.Nqh)

    def _v2(self, a):
        return a.reshape(self.nh, a.shape[1], self.Nqk2, self.Nqh)

    def dostep(self, Q3, Q2, time):
        dg3, dg2, H = self.dg3, self.dg2, self.H
        A3, A2 = dg3.state_auxiliary, dg2.state_auxiliary
        ns = len(RKA)
        # 2-D auxiliary columns: G_U 0:2, U_c 2:4, eta_c 4, U_s 5:7, eta_s 7, Delta_u 8:10,
        # eta_diag 10, Delta_eta 11; 2-D state U 0:2, eta 2; 3-D auxiliary dG_u 5:7
        for s in range(ns):
            first, last = s == 0, s == ns - 1
            ts = time + RKC[s] * self.dt
            fract_dt = ((1 - RKC[s]) if last else (RKC[s + 1] - RKC[s])) * self.dt
            # initialize_fast_state!
            if self.add == 0:
                steps = int(np.ceil(fract_dt / self.dt_fast)) if self.dt_fast > 0 else 1
                fs1 = fs2 = fs3 = steps
            else:
                steps = int(np.ceil(fract_dt / self.dt_fast / self.add)) if self.dt_fast > 0 else 1
                fs2, fs1, fs3 = self.add * steps, (self.add - 1) * steps, (self.add + 1) * steps
            fdt = fract_dt / fs2
            count = 0.0
            A2[:, 2:5, :] = -0.0
            if not first:
                Q2[:, 2, :] = A2[:, 7, :]
                Q2[:, 0:2, :] = A2[:, 5:7, :]
            A3[:, 5:7, :] = 0.0                                  # initialize_adjustment!
            dg3(self.dQ2fast, Q3, ts, 1.0, 0.0)                  # increment = false
            top = dg3.integrate_velocity(self.dQ2fast)           # tendency_from_slow_to_fast!
            self._v2(A2)[:, 0:2] = top[:, :, None, :]
            self._v3(A3)[:, :, 5:7] = (-top / H)[:, None, :, None, :]
            dg3(self.dQ3, Q3, ts, 1.0, 1.0)                      # increment = true
            n = dg3.grid.nreal * Q3.shape[1] * Q3.shape[2]
            lib().orc_lsrk_update(_p(self.dQ3), _p(Q3), C.c_double(RKA[(s + 1) % ns]),
                                  C.c_double(RKB[s]), C.c_double(self.dt), C.c_int64(n))
            for sub in range(1, fs3 + 1):
                lsrk_step(dg2, Q2, self.dQ2, ts + (sub - 1) * fdt, fdt, RKA, RKB, RKC)
                if sub >= fs1:                                   # cummulate_fast_solution!
                    A2[:, 2:4, :] += Q2[:, 0:2, :]
                    A2[:, 4, :] += Q2[:, 2, :]
                    count += 1.0
                if sub == fs2:
                    A2[:, 5:7, :] = Q2[:, 0:2, :]
                    A2[:, 7, :] = Q2[:, 2, :]
            # reconcile_from_fast_to_slow!
            A2[:, 2:5, :] *= 1 / count
            top = dg3.integrate_velocity(Q3)
            du = self._v2(A2)[:, 2:4, 0, :].copy()
            du -= top
            du /= H
            self._v2(A2)[:, 8:10] = du[:, :, None, :]
            self._v3(Q3)[:, :, 0:2] += du[:, None, :, None, :]
            if last:
                flat_eta = self._v3(Q3)[:, -1, 2, -1, :]
                self._v2(A2)[:, 10] = flat_eta[:, None, :]
                A2[:, 11, :] = A2[:, 4, :] - A2[:, 10, :]
                self._v3(Q3)[:, :, 2] = self._v2(A2)[:, 4, 0, :][:, None, None, :]
                Q2[:, 2, :] = A2[:, 7, :]
                Q2[:, 0:2, :] = A2[:, 5:7, :]


# ---- element filters (filter_oracle.c) --------------------------------------------------
class _FilterTarget(C.Structure):
    _fields_ = [("kind", C.c_int), ("nfs", C.c_int), ("idx", C.c_int * MAXS),
                ("aux_ref_rho", C.c_int), ("aux_ref_rhoe", C.c_int)]


def _c_target(target):
    """``target``: an object with ``target_id``, ``indices`` (1-based) and ``aux_offsets()``
    (the host classes of ``mesh/filters.py``)."""
    t = _FilterTarget()
    t.kind, t.nfs = target.target_id, len(target.indices)
    for i, v in enumerate(target.indices):
        t.idx[i] = v
    t.aux_ref_rho, t.aux_ref_rhoe = target.aux_offsets()
    return t


def apply_filter(Q, target, grid, filt, direction=EVERY, state_auxiliary=None):
    """``Filters.apply!(Q, target, grid, filter; direction, state_auxiliary)`` restated:
    the launch sequence of ``apply_async!`` (Filters.jl:440-505 spectral, :507-540 TMAR,
    :542-607 mass preserving) over the C kernels.  ``Q``: numpy ``(nelem, nstate, Np)``,
    filtered in place on the real elements; works for 2-D and 3-D grids."""
    L = lib()
    assert Q.flags.c_contiguous and Q.dtype == np.float64
    dim, nstate = grid.dim, Q.shape[1]
    Nq = (C.c_int * 3)(*(list(grid.Nq) + [1] * (3 - dim)))
    tg = _c_target(target)
    nreal = C.c_int64(grid.nreal)
    if state_auxiliary is None:
        aux, naux = None, 0
    else:
        aux = np.ascontiguousarray(state_auxiliary, dtype=np.float64)
        naux = aux.shape[1]
    auxp = _p(aux) if aux is not None else None
    vgeo = np.ascontiguousarray(grid.vgeo, dtype=np.float64)
    MCOL = 9     # _M, Grids.jl:129-146
    if filt.kind == 2:
        L.orc_apply_tmar_filter(dim, Nq, _p(Q), nstate, C.byref(tg), _p(vgeo), vgeo.shape[1],
                                MCOL, nreal)
        return
    calls = []
    if direction in (EVERY, HORIZONTAL):
        calls.append((HORIZONTAL, filt.filter_matrices[0]))
    if direction in (EVERY, VERTICAL):
        calls.append((VERTICAL, filt.filter_matrices[-1]))
    for d, F in calls:
        Fc = np.ascontiguousarray(np.asarray(F, dtype=np.float64).T)    # column-major
        if filt.kind == 0:
            L.orc_apply_filter(dim, Nq, d, _p(Q), nstate, auxp, naux, C.byref(tg), _p(Fc), nreal)
        else:
            L.orc_apply_mp_filter(dim, Nq, d, _p(Q), nstate, auxp, naux, C.byref(tg), _p(Fc),
                                  _p(vgeo), vgeo.shape[1], MCOL, nreal)
