"""Host-side mirror of the reference's ``DGModel`` operator API, backed by libcmdg.

Reference: ``DGModel(balance_law, grid, nf_first, nf_second, nf_gradient; direction,
diffusion_direction, state_auxiliary, ...)`` ``src/Numerics/DGMethods/DGModel.jl:22-65``;
the callable ``(dg)(tendency, Q, _, t, alpha, beta)`` ``:85-427``; ``init_ode_state``
``SpaceDiscretization.jl:79-148``; ``norm`` / ``euclidean_distance``
``src/Arrays/MPIStateArrays.jl:583-644``.

State arrays are torch float64 CUDA(=HIP) tensors of numpy-style shape
``(nelem, nstate, Np)`` -- the memory image of the reference's ``(Np, nstate, nelem)``.
torch is used for device memory only; every kernel is in libcmdg.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .balancelaws import EveryDirection, RusanovNumericalFlux

__all__ = ["DGModel", "connect_local", "group_rhs", "group_lsrk_run", "group_halo",
           "reference_pressure_gradient", "rccl_unique_id",
           "ADVECTIVE_COURANT", "NONDIFFUSIVE_COURANT", "DIFFUSIVE_COURANT"]

ADVECTIVE_COURANT, NONDIFFUSIVE_COURANT, DIFFUSIVE_COURANT = 0, 1, 2


def _dev(a, device, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(device)


class DGModel:
    def __init__(self, balance_law, grid, numerical_flux_first_order=RusanovNumericalFlux,
                 direction=EveryDirection, diffusion_direction=None, device="cuda:0",
                 state_auxiliary=None, keep_gradient_flux=False):
        if not torch.cuda.is_available():
            raise _lib.CmdgError("DGModel needs a HIP device; there is no CPU fallback")
        L = _lib.lib()
        self.L = L
        self.balance_law = balance_law
        self.grid = grid
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.direction = direction
        self.diffusion_direction = direction if diffusion_direction is None else diffusion_direction
        law, g = balance_law, grid
        if g.dim != 3 or g.N[0] != g.N[1]:
            raise _lib.CmdgError("libcmdg: 3-D grids with one horizontal polynomial order only")
        ip, dp = law.descriptor()
        counts = (C.c_int32 * 6)()
        ipa = (C.c_int32 * 16)(*[int(v) for v in ip])
        _lib.check(L.cmdg_physics_counts(law.physics_id, C.cast(ipa, C.c_void_p), C.cast(counts, C.c_void_p)))
        want = (law.ns, law.naux, law.ngrad, law.ngradflux, law.ngradlap, law.nhyper)
        if tuple(counts) != want:
            raise _lib.CmdgError("state counts differ: host %s device functor %s" % (want, tuple(counts)))
        dev = self.device
        # grid tables (reference layouts)
        self._vgeo = _dev(g.vgeo, dev)
        self._sgeo = _dev(g.sgeo, dev)
        self._vmapM = _dev(g.vmapM, dev)
        self._vmapP = _dev(g.vmapP, dev)
        self._elemtobndy = _dev(g.elemtobndy, dev)
        self._interior = _dev(np.asarray(g.interiorelems, dtype=np.int64), dev)
        self._exterior = _dev(np.asarray(g.exteriorelems, dtype=np.int64), dev)
        self._active = _dev(g.activedofs.astype(np.uint8), dev)
        self._D = np.ascontiguousarray(g.D[0].T, dtype=np.float64)   # column-major (Nq, Nq)
        self._Dv = np.ascontiguousarray(g.D[-1].T, dtype=np.float64)  # vertical (Nqv, Nqv)
        self._vmapsend = _dev(np.asarray(g.vmapsend, dtype=np.int64), dev)
        self._vmaprecv = _dev(np.asarray(g.vmaprecv, dtype=np.int64), dev)
        nn = len(g.nabrtorank)
        self._nabr = np.asarray(g.nabrtorank, dtype=np.int32)
        self._nsend = np.asarray(g.nabrtovmapsend, dtype=np.int64).reshape(-1)
        self._nrecv = np.asarray(g.nabrtovmaprecv, dtype=np.int64).reshape(-1)
        aux = law.init_state_auxiliary(g) if state_auxiliary is None else state_auxiliary
        if state_auxiliary is None and getattr(law, "discrete_hydrostatic_balance", False):
            # atmos_init_aux!(::HydrostaticState) step 2 (ref_state.jl:150-175) on the device
            gradp = reference_pressure_gradient(g, aux[:, law.off_ref + 1, :], device)
            law.rebalance_reference_state(g, aux, gradp)
        self.state_auxiliary = _dev(aux, dev) if isinstance(aux, np.ndarray) else aux
        ne, Np = g.nelem, g.Np
        self._gradient_flux = torch.zeros((ne, max(law.ngradflux, 1), Np), dtype=torch.float64, device=dev)
        self._hypervisc_grad = None         # reference layout, on demand (Qhypervisc_grad)
        self.Qhypervisc_div = torch.zeros((ne, max(law.nhyper, 1), Np), dtype=torch.float64, device=dev)
        d = _lib.CmdgDesc()
        d.dim = 3
        d.N[:] = list(g.N)
        d.nreal, d.nghost = g.nreal, g.nelem - g.nreal
        d.nvgeo = g.vgeo.shape[1]
        d.physics_id = law.physics_id
        d.iparam[:] = [int(v) for v in ip]
        d.dparam[:] = [float(v) for v in dp] + [0.0] * (64 - len(dp))
        d.nf_first = int(numerical_flux_first_order)
        d.direction, d.diffusion_direction = int(self.direction), int(self.diffusion_direction)
        d.stacked = int(bool(g.topology.isstacked))
        d.vgeo, d.sgeo = self._vgeo.data_ptr(), self._sgeo.data_ptr()
        d.vmapM, d.vmapP = self._vmapM.data_ptr(), self._vmapP.data_ptr()
        d.elemtobndy = self._elemtobndy.data_ptr()
        d.interiorelems, d.ninterior = self._interior.data_ptr(), self._interior.numel()
        d.exteriorelems, d.nexterior = self._exterior.data_ptr(), self._exterior.numel()
        d.activedofs = self._active.data_ptr()
        d.D = self._D.ctypes.data
        d.Dv = self._Dv.ctypes.data
        d.vmapsend, d.nvmapsend = self._vmapsend.data_ptr(), self._vmapsend.numel()
        d.vmaprecv, d.nvmaprecv = self._vmaprecv.data_ptr(), self._vmaprecv.numel()
        d.nnabr = nn
        d.nabrtorank = self._nabr.ctypes.data
        d.nabrtovmapsend = self._nsend.ctypes.data
        d.nabrtovmaprecv = self._nrecv.ctypes.data
        d.state_auxiliary = self.state_auxiliary.data_ptr()
        d.state_gradient_flux = self._gradient_flux.data_ptr()
        d.Qhypervisc_grad = 0               # the library's working copy is node-major (cmdg.h)
        d.Qhypervisc_div = self.Qhypervisc_div.data_ptr()
        torch.cuda.synchronize(dev)          # tables uploaded on torch's stream
        h = C.c_void_p()
        _lib.check(L.cmdg_create(C.byref(d), C.byref(h)))
        self.handle = h
        self._desc = d
        if keep_gradient_flux:
            self.set_option(_lib.OPT_KEEP_GRADFLUX, 1)
        if g.topology.isstacked and g.topology.stacksize:
            # length(topology.stacksize): lets the engine pick the launch order of tall stacks
            self.set_option(_lib.OPT_STACK_HEIGHT, int(g.topology.stacksize))

    def set_option(self, option, value):
        """``cmdg_set_option``: ``_lib.OPT_KEEP_GRADFLUX`` = refresh ``state_gradient_flux`` in
        every evaluation even when the law's fluxes never read it (zero viscosity);
        ``_lib.OPT_STACK_HEIGHT`` = elements per vertical stack (launch order only);
        ``_lib.OPT_REFERENCE_HALO`` = pack / unpack kernels around every ghost exchange and ghost
        elements refreshed, as the reference does (default: exterior launches write the send
        buffers, face kernels read the receive buffers)."""
        _lib.check(self.L.cmdg_set_option(self.handle, int(option), int(value)), self.handle)

    @property
    def state_gradient_flux(self):
        """``dg.state_gradient_flux`` as the reference holds it, ``(nelem, ngradflux, Np)``
        (``cmdg_export_gradient_flux``: a copy out of the node-major working array for the
        atmosphere laws, the working array itself otherwise)."""
        if self.balance_law.ngradflux > 0:
            self._torch_ready()
            _lib.check(self.L.cmdg_export_gradient_flux(self.handle, self._gradient_flux.data_ptr()), self.handle)
        return self._gradient_flux

    @property
    def Qhypervisc_grad(self):
        """``dg.states_higher_order[1]`` as the reference holds it, ``(nelem, 3 ngradlap, Np)``:
        ``cmdg_export_hypervisc_grad`` (the library's working copy is node-major; this is a
        copy made at the time of the call)."""
        g, law = self.grid, self.balance_law
        if self._hypervisc_grad is None:
            self._hypervisc_grad = torch.zeros((g.nelem, max(3 * law.ngradlap, 1), g.Np), dtype=torch.float64,
                                               device=self.device)
        if law.ngradlap > 0:
            self._torch_ready()      # (the fill of a fresh destination runs on torch's stream, the copy on ours)
            _lib.check(self.L.cmdg_export_hypervisc_grad(self.handle, self._hypervisc_grad.data_ptr()), self.handle)
        return self._hypervisc_grad

    def query(self, item):
        """``cmdg_query``: what the handle's kernels do (``_lib.CMDG_Q`` names, e.g.
        ``"GRADFLUX_LIVE"``, ``"DIRECT_SEND"``; ``("AUX_READ", p)`` for pass ``p``)."""
        out = C.c_int64()
        what = _lib.CMDG_Q[item[0]] + int(item[1]) if isinstance(item, tuple) else _lib.CMDG_Q[item]
        _lib.check(self.L.cmdg_query(self.handle, what, C.byref(out)), self.handle)
        return int(out.value)

    def close(self):
        if getattr(self, "handle", None):
            self.L.cmdg_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state helpers --------------------------------------------------------------
    def create_state(self, nstate=None):
        ns = self.balance_law.ns if nstate is None else nstate
        return torch.zeros((self.grid.nelem, ns, self.grid.Np), dtype=torch.float64,
                           device=self.device)

    def init_ode_state(self, t=0.0):
        """``init_ode_state(dg, t)``: one-time, on the host."""
        aux = self.state_auxiliary.cpu().numpy()
        Q = self.balance_law.init_state_prognostic(self.grid, aux, t)
        q = _dev(Q, self.device)
        torch.cuda.synchronize(self.device)
        return q

    # -- the operator ---------------------------------------------------------------
    def __call__(self, tendency, Q, t, alpha=1.0, beta=0.0, increment=None):
        """``dg(tendency, Q, nothing, t, alpha, beta)``; the 4-argument form with
        ``increment`` maps to ``(alpha, beta) = (1, increment)`` (SpaceDiscretization.jl:68-77)."""
        if increment is not None:
            alpha, beta = 1.0, float(bool(increment))
        self._torch_ready()
        _lib.check(self.L.cmdg_rhs(self.handle, tendency.data_ptr(), Q.data_ptr(), float(t),
                                   float(alpha), float(beta)), self.handle)

    def synchronize(self):
        _lib.check(self.L.cmdg_synchronize(self.handle), self.handle)

    def _torch_ready(self):
        """libcmdg runs on its own HIP streams: anything torch enqueued on its current
        stream (fills, copies of the arrays we are handed) must have finished first."""
        torch.cuda.current_stream(self.device).synchronize()

    def lsrk_run(self, Q, dQ, t, dt, nsteps, rka, rkb, rkc):
        a = (C.c_double * len(rka))(*rka)
        b = (C.c_double * len(rkb))(*rkb)
        c = (C.c_double * len(rkc))(*rkc)
        self._torch_ready()
        _lib.check(self.L.cmdg_lsrk_run(
            self.handle, Q.data_ptr(), dQ.data_ptr(), float(t), float(dt), int(nsteps),
            len(rka), C.cast(a, C.c_void_p), C.cast(b, C.c_void_p), C.cast(c, C.c_void_p)),
            self.handle)

    # -- transport --------------------------------------------------------------------
    # -- Courant numbers (SpaceDiscretization.jl:307-365, DGMethods.jl:79-83) ----------------
    def courant(self, local_courant, Q, dt, simtime=0.0, direction=EveryDirection):
        """``courant(local_courant, dg, m, Q, dt, simtime, direction)``: rank-local maximum;
        ``local_courant`` is ``ADVECTIVE_COURANT``, ``NONDIFFUSIVE_COURANT`` or
        ``DIFFUSIVE_COURANT``.  Multi-rank callers reduce with ``max``."""
        out = C.c_double()
        self._torch_ready()
        _lib.check(self.L.cmdg_courant(self.handle, int(local_courant), Q.data_ptr(), float(dt),
                                       float(simtime), int(direction), C.byref(out)), self.handle)
        return out.value

    def min_node_distance(self, direction=EveryDirection):
        """``min_node_distance(grid, direction)`` (Grids.jl:455-486), rank-local."""
        out = C.c_double()
        _lib.check(self.L.cmdg_min_node_distance(self.handle, int(direction), C.byref(out)),
                   self.handle)
        return out.value

    def calculate_dt(self, Q, courant_number, t=0.0, direction=EveryDirection):
        """``calculate_dt(dg, model, Q, Courant_number, t, direction)`` (DGMethods.jl:79-83)."""
        return courant_number / self.courant(NONDIFFUSIVE_COURANT, Q, 1.0, t, direction)

    # -- column integrals (DGModel.jl:445-529) -----------------------------------------------
    def _stack_desc(self, src, scale, dst, rsrc, rdst):
        d = _lib.CmdgStackIntegralDesc()
        d.nout = len(dst) if dst else len(rdst)
        for s in range(d.nout):
            if src:
                d.src_is_state[s], d.src_col[s] = int(src[s][0]), int(src[s][1])
                d.scale[s] = float(scale[s]) if scale else 1.0
                d.dst_col[s] = int(dst[s])
            if rsrc:
                d.rsrc_col[s], d.rdst_col[s] = int(rsrc[s]), int(rdst[s])
        return d

    def indefinite_stack_integral(self, Q, aux, src, dst, scale=None):
        """``indefinite_stack_integral!(dg, m, Q, state_auxiliary, t)``: the upward integral
        of ``scale_s * field_s`` (``src = [(is_state, column), ...]``, 0-based) along every
        stack of elements goes to auxiliary column ``dst[s]``."""
        g = self.grid
        Imat = np.ascontiguousarray(np.asarray(g.Imat[-1], dtype=np.float64).T)
        d = self._stack_desc(src, scale, dst, None, None)
        self._torch_ready()
        _lib.check(self.L.cmdg_indefinite_stack_integral(
            self.handle, Q.data_ptr() if Q is not None else None,
            Q.shape[1] if Q is not None else 0, aux.data_ptr(), aux.shape[1],
            int(g.topology.stacksize), Imat.ctypes.data, C.byref(d)), self.handle)

    def reverse_indefinite_stack_integral(self, aux, rsrc, rdst):
        """``reverse_indefinite_stack_integral!``: auxiliary column ``rdst[s]`` receives
        (value of column ``rsrc[s]`` at the top of the stack) - (its value at the node)."""
        d = self._stack_desc(None, None, None, rsrc, rdst)
        self._torch_ready()
        _lib.check(self.L.cmdg_reverse_indefinite_stack_integral(
            self.handle, aux.data_ptr(), aux.shape[1], int(self.grid.topology.stacksize),
            C.byref(d)), self.handle)

    def set_rhs_hooks(self, pre_filters=(), gradflux_to_aux=(), integral=None,
                      reverse_integral=None, surface_to_column=(), flow_deviation=None,
                      pre_rhs=None, ops_before_gradients=False):
        """The composition a law's ``update_auxiliary_state!`` /
        ``update_auxiliary_state_gradient!`` overrides stand for (see ``cmdg_rhs_hooks`` in
        include/cmdg.h); ``set_rhs_hooks()`` with no arguments clears them."""
        if not (pre_filters or gradflux_to_aux or integral or reverse_integral
                or surface_to_column or flow_deviation or pre_rhs):
            self._hooks = None
            _lib.check(self.L.cmdg_set_rhs_hooks(self.handle, None), self.handle)
            return
        hk = _lib.CmdgRhsHooks()
        hk.npre = len(pre_filters)
        for i, f in enumerate(pre_filters):
            hk.pre_filter[i] = f.handle
        hk.ncopy = len(gradflux_to_aux)
        for i, (g, a, sc) in enumerate(gradflux_to_aux):
            hk.copy_gf_col[i], hk.copy_aux_col[i], hk.copy_scale[i] = int(g), int(a), float(sc)
        if integral:
            n = len(integral["dst"])
            hk.has_integral = 1
            hk.integral.nout = n
            for s in range(n):
                hk.integral.src_is_state[s], hk.integral.src_col[s] = integral["src"][s]
                hk.integral.scale[s] = float(integral.get("scale", [1.0] * n)[s])
                hk.integral.dst_col[s] = int(integral["dst"][s])
        if reverse_integral:
            n = len(reverse_integral["rdst"])
            hk.has_reverse_integral = 1
            hk.reverse_integral.nout = n
            for s in range(n):
                hk.reverse_integral.rsrc_col[s] = int(reverse_integral["rsrc"][s])
                hk.reverse_integral.rdst_col[s] = int(reverse_integral["rdst"][s])
        hk.nsurf = len(surface_to_column)
        for i, (a, b) in enumerate(surface_to_column):
            hk.surf_src_col[i], hk.surf_dst_col[i] = int(a), int(b)
        if flow_deviation:      # (state column of u, auxiliary column of u_d, depth H)
            hk.has_flow_deviation = 1
            hk.flow_u_col, hk.flow_ud_col = int(flow_deviation[0]), int(flow_deviation[1])
            hk.flow_H = float(flow_deviation[2])
        if pre_rhs:             # (nested DGModel, its tendency column, auxiliary column)
            hk.pre_rhs_handle = pre_rhs[0].handle
            hk.pre_rhs_src_col, hk.pre_rhs_dst_aux_col = int(pre_rhs[1]), int(pre_rhs[2])
        hk.ops_before_gradients = int(bool(ops_before_gradients))
        hk.nvertelem = int(self.grid.topology.stacksize or 0)
        Imat = np.ascontiguousarray(np.asarray(self.grid.Imat[-1], dtype=np.float64).T)
        hk.Imat = Imat.ctypes.data
        self._hooks = (hk, Imat, list(pre_filters), pre_rhs)      # keep alive
        _lib.check(self.L.cmdg_set_rhs_hooks(self.handle, C.byref(hk)), self.handle)

    def set_filters(self, gradient_filter=None, tendency_filter=None, step_filter=None):
        """``DGModel(...; gradient_filter, tendency_filter)`` (DGModel.jl:44-45, applied at
        :185-193 and :417-425) and the every-step user filter callback of
        ``experiments/AtmosGCM/heldsuarez.jl:261-272``.  Arguments are
        ``mesh.filters.DeviceFilter`` objects (or None) bound to this model."""
        fs = (gradient_filter, tendency_filter, step_filter)
        self._filters = fs          # keep them alive
        hs = [f.handle if f is not None else None for f in fs]
        _lib.check(self.L.cmdg_set_filters(self.handle, *hs), self.handle)

    def halo_begin(self, array):
        """``begin_ghost_exchange!(array)`` (MPIStateArrays.jl:411-442) of a state-like array."""
        self._torch_ready()
        _lib.check(self.L.cmdg_halo_begin(self.handle, array.data_ptr(), array.shape[1]), self.handle)

    def halo_end(self, array):
        """``end_ghost_exchange!(array)`` (MPIStateArrays.jl:451-483)."""
        _lib.check(self.L.cmdg_halo_end(self.handle, array.data_ptr(), array.shape[1]), self.handle)

    def comm_init_rccl(self, unique_id, rank, nranks):
        """``unique_id``: the 128 bytes of ``rccl_unique_id()`` made on rank 0."""
        _lib.check(self.L.cmdg_comm_init_rccl(self.handle, bytes(unique_id), int(rank),
                                              int(nranks)), self.handle)

    def comm_selftest(self, count=4096):
        _lib.check(self.L.cmdg_comm_selftest(self.handle, int(count)), self.handle)

    # -- reductions (local part) ------------------------------------------------------
    def norm2_local(self, A, weighted=True):
        out = C.c_double()
        self._torch_ready()
        _lib.check(self.L.cmdg_norm2_local(self.handle, A.data_ptr(), A.shape[1], int(weighted),
                                           C.cast(C.byref(out), C.c_void_p)), self.handle)
        return out.value

    def distance2_local(self, A, B):
        out = C.c_double()
        self._torch_ready()
        _lib.check(self.L.cmdg_distance2_local(self.handle, A.data_ptr(), B.data_ptr(),
                                               A.shape[1], C.cast(C.byref(out), C.c_void_p)),
                   self.handle)
        return out.value

    def norm(self, A, weighted=True):
        return math.sqrt(self.norm2_local(A, weighted))

    def euclidean_distance(self, A, B):
        return math.sqrt(self.distance2_local(A, B))

    # -- measurement ------------------------------------------------------------------
    def profile_enable(self, on=True):
        _lib.check(self.L.cmdg_profile_enable(self.handle, int(on)), self.handle)

    def profile_reset(self):
        _lib.check(self.L.cmdg_profile_reset(self.handle), self.handle)

    def profile_get(self, kernel):
        ms, n = C.c_double(), C.c_int64()
        _lib.check(self.L.cmdg_profile_get(self.handle, _lib.CMDG_K[kernel],
                                           C.cast(C.byref(ms), C.c_void_p),
                                           C.cast(C.byref(n), C.c_void_p)), self.handle)
        return ms.value, n.value


def reference_pressure_gradient(grid, p, device="cuda:0"):
    """``grad reference_pressure`` (src/Atmos/Model/ref_state.jl:235-262): one evaluation of
    the PressureGradientModel operator with central fluxes; ``(nelem, 3, Np)`` numpy array.
    Ghost elements hold the analytic pressure already, so this evaluation needs no exchange:
    the operator is built on a view of the grid without neighbours."""
    import copy
    from .balancelaws import PressureGradientModel
    g = copy.copy(grid)
    g.nabrtorank = []
    g.nabrtovmapsend = np.zeros((2, 0), dtype=np.int64)
    g.nabrtovmaprecv = np.zeros((2, 0), dtype=np.int64)
    dg = DGModel(PressureGradientModel(p), g, numerical_flux_first_order=1, device=device)
    Q = dg.create_state(3)
    T = dg.create_state(3)
    dg(T, Q, 0.0, 1.0, 0.0)
    out = T.cpu().numpy()
    dg.close()
    return out


def group_halo(dgs, arrays):
    """One ghost exchange of a state-like array per connected handle (``cmdg_group_halo``)."""
    for d in dgs:
        d._torch_ready()
    _lib.check(dgs[0].L.cmdg_group_halo(_harr(dgs), len(dgs), _parr(arrays), arrays[0].shape[1]),
               dgs[0].handle)


def rccl_unique_id():
    buf = (C.c_char * 128)()
    _lib.check(_lib.lib().cmdg_comm_unique_id(C.cast(buf, C.c_void_p)))
    return bytes(buf.raw)


def _harr(dgs):
    return (C.c_void_p * len(dgs))(*[d.handle for d in dgs])


def _parr(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def connect_local(dgs):
    """Connect the per-rank models of one process (rank r = dgs[r]) through device
    copies: the single-GPU rehearsal of the multi-GPU halo path."""
    L = _lib.lib()
    _lib.check(L.cmdg_comm_connect_local(C.cast(_harr(dgs), C.c_void_p), len(dgs)), dgs[0].handle)


def group_rhs(dgs, tendencies, Qs, t, alpha=1.0, beta=0.0):
    L = _lib.lib()
    dgs[0]._torch_ready()
    _lib.check(L.cmdg_group_rhs(C.cast(_harr(dgs), C.c_void_p), len(dgs),
                                C.cast(_parr(tendencies), C.c_void_p),
                                C.cast(_parr(Qs), C.c_void_p), float(t), float(alpha),
                                float(beta)), dgs[0].handle)
    for d in dgs:
        d.synchronize()


def group_lsrk_run(dgs, Qs, dQs, t, dt, nsteps, rka, rkb, rkc):
    L = _lib.lib()
    a = (C.c_double * len(rka))(*rka)
    b = (C.c_double * len(rkb))(*rkb)
    c = (C.c_double * len(rkc))(*rkc)
    dgs[0]._torch_ready()
    _lib.check(L.cmdg_group_lsrk_run(
        C.cast(_harr(dgs), C.c_void_p), len(dgs), C.cast(_parr(Qs), C.c_void_p),
        C.cast(_parr(dQs), C.c_void_p), float(t), float(dt), int(nsteps), len(rka),
        C.cast(a, C.c_void_p), C.cast(b, C.c_void_p), C.cast(c, C.c_void_p)), dgs[0].handle)
