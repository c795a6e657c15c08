"""Ghost exchange without pack / unpack launches (HaloDev, csrc/cmdg_common.h).

By default the exterior launch of every pass writes the nodes of vmapsend straight into the send
buffer and the face kernels read ghost neighbours from the receive buffers; with
``OPT_REFERENCE_HALO`` the handle runs begin/end_ghost_exchange! as the reference does
(kernel_fillsendbuf! / kernel_transferrecvbuf! around every exchange, MPIStateArrays.jl:411-483).
Both must give the same bits, and the default must really launch nothing: the HIP-event
counters of the pack / unpack kernels say how many ran.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_group(cm, torch, make, size, nsteps, dt, reference, nan_ghosts=True):
    """``nsteps`` fused LSRK54 steps of a ``size``-rank partition on one GPU (local transport).
    Returns the per-rank real-element states, the handles' answers to DIRECT_SEND / DIRECT_RECV
    and the pack / unpack launch counts of rank 0."""
    dgs, Qs, grids = [], [], []
    for r in range(size):
        law, grid, dg = make(r, size)
        if reference:
            dg.set_option(cm._lib.OPT_REFERENCE_HALO, 1)
        q = dg.init_ode_state(0.0)
        if nan_ghosts:
            q[grid.nreal:] = float("nan")      # whatever is read of a ghost comes from the exchange
        dgs.append(dg), Qs.append(q), grids.append(grid)
    dQs = [x.create_state() for x in dgs]
    torch.cuda.synchronize()
    cm.dgmodel.connect_local(dgs)
    modes = (dgs[0].query("DIRECT_SEND"), dgs[0].query("DIRECT_RECV"))
    s = cm.odesolvers.LSRK54CarpenterKennedy(dgs[0], Qs[0], dt=dt)
    dgs[0].profile_reset()
    dgs[0].profile_enable(True)
    cm.dgmodel.group_lsrk_run(dgs, Qs, dQs, 0.0, dt, nsteps, s.RKA, s.RKB, s.RKC)
    for x in dgs:
        x.synchronize()
    dgs[0].profile_enable(False)
    counts = {k: dgs[0].profile_get(k)[1] for k in ("PACK", "UNPACK", "TRANSPORT")}
    out = [q[:g.nreal].cpu().numpy().copy() for q, g in zip(Qs, grids)]
    ghosts = [q[g.nreal:].cpu().numpy().copy() for q, g in zip(Qs, grids)]
    for x in dgs:
        x.close()
    return out, ghosts, modes, counts


def test_held_suarez_direct_exchange_equals_reference_exchange(cm, torch):
    """Held-Suarez on the cubed sphere, 3 ranks (orientation-3 faces across ranks), 4 exchanges per
    stage (Q, grad G, Laplacian, nu grad^3): the law's nodal refresh is fused into the gradient pass,
    so both halves run direct -- one pack launch in the whole run (the caller's Q at the first
    stage), no unpack launch, ghost elements of Q never written."""
    from helpers import held_suarez_setup

    def make(r, size):
        law, grid, d, dd = held_suarez_setup(3, 2, rank=r, size=size)
        return law, grid, cm.dgmodel.DGModel(law, grid, direction=d, diffusion_direction=dd)

    nsteps = 2
    direct, gh_d, modes_d, n_d = _run_group(cm, torch, make, 3, nsteps, 2.0, reference=False)
    ref, gh_r, modes_r, n_r = _run_group(cm, torch, make, 3, nsteps, 2.0, reference=True)
    assert modes_d == (1, 1) and modes_r == (0, 0)
    for a, b in zip(direct, ref):
        assert np.isfinite(a).all()
        assert np.array_equal(a, b)                       # same kernels, same payloads: same bits
    nex = 4 * 5 * nsteps
    assert n_r["PACK"] == nex and n_r["UNPACK"] == nex, n_r
    assert n_d["PACK"] == 1 and n_d["UNPACK"] == 0, n_d
    assert n_d["TRANSPORT"] == n_r["TRANSPORT"]
    assert all(np.isnan(g).all() for g in gh_d)           # nothing unpacked
    assert all(np.isfinite(g).any() for g in gh_r)


def test_advection_diffusion_direct_exchange(cm, torch):
    """Config 1's law on 2 ranks: two exchanges per stage (Q and the gradient flux)."""
    from helpers import pseudo1d_setup

    def make(r, size):
        law, grid, dt = pseudo1d_setup(direction=0, rank=r, size=size)
        return law, grid, cm.dgmodel.DGModel(law, grid, direction=0)

    dt = pseudo1d_setup(direction=0)[2]
    direct, _, modes_d, n_d = _run_group(cm, torch, make, 2, 3, dt, reference=False)
    ref, _, modes_r, n_r = _run_group(cm, torch, make, 2, 3, dt, reference=True)
    assert modes_d == (1, 1) and modes_r == (0, 0)
    for a, b in zip(direct, ref):
        assert np.array_equal(a, b)
    assert n_r["UNPACK"] == n_r["PACK"] == 2 * 5 * 3
    assert n_d["PACK"] == 1 and n_d["UNPACK"] == 0, n_d


def test_direct_send_with_unpacked_receive(cm, torch):
    """The moist LES law (BOMEX pieces): its nodal update_auxiliary_state! is a kernel of its own
    and runs on the ghost elements too, so what is received is unpacked into them (one switch
    for all arrays of a handle); the send side is direct."""
    from helpers import bomex_setup

    def make(r, size):
        law, grid = bomex_setup(nx=4, ny=4, nz=4, rank=r, size=size)[:2]
        return law, grid, cm.dgmodel.DGModel(law, grid)

    direct, _, modes_d, n_d = _run_group(cm, torch, make, 2, 2, 0.01, reference=False, nan_ghosts=False)
    ref, _, modes_r, n_r = _run_group(cm, torch, make, 2, 2, 0.01, reference=True, nan_ghosts=False)
    assert modes_d == (1, 0) and modes_r == (0, 0)
    for a, b in zip(direct, ref):
        assert np.isfinite(a).all()
        assert np.array_equal(a, b)
    assert n_d["UNPACK"] == n_r["UNPACK"] == n_r["PACK"] == 2 * 5 * 2
    assert n_d["PACK"] == 1, n_d


def test_two_pipelines_equal_one_stream(cm, torch):
    """CMDG_OPT_HALO_PIPELINE: exterior launches and exchanges on the halo stream, interior
    launches on the compute stream (default) == the reference's order on one stream."""
    from helpers import held_suarez_setup
    out = []
    for pipeline in (1, 0):
        def make(r, size):
            law, grid, d, dd = held_suarez_setup(3, 2, rank=r, size=size)
            dg = cm.dgmodel.DGModel(law, grid, direction=d, diffusion_direction=dd)
            dg.set_option(cm._lib.OPT_HALO_PIPELINE, pipeline)
            assert dg.query("HALO_PIPELINE") == pipeline
            return law, grid, dg
        out.append(_run_group(cm, torch, make, 4, 2, 2.0, reference=False)[0])
    for a, b in zip(*out):
        assert np.isfinite(a).all()
        assert np.array_equal(a, b)


def test_smagorinsky_bubble_direct_exchange(cm, torch):
    """Rising bubble (SmagorinskyLilly: the gradient-flux state is live and exchanged, its plus
    side read by the tendency pass), 2 ranks."""
    from helpers import rising_bubble_setup

    def make(r, size):
        law, grid = rising_bubble_setup(nx=4, ny=4, nz=3, rank=r, size=size)
        return law, grid, cm.dgmodel.DGModel(law, grid)

    dt = 0.05
    direct, _, modes_d, n_d = _run_group(cm, torch, make, 2, 2, dt, reference=False)
    ref, _, modes_r, n_r = _run_group(cm, torch, make, 2, 2, dt, reference=True)
    assert modes_d == (1, 1)
    for a, b in zip(direct, ref):
        assert np.isfinite(a).all()
        assert np.array_equal(a, b)
    assert n_d["PACK"] == 1 and n_d["UNPACK"] == 0, n_d


def test_direct_exchange_through_rccl(cm, torch):
    """The same through the RCCL transport: rank 0 of a 2-rank periodic brick with itself as every
    neighbour (one GPU), hyperdiffusion law (Q, grad G, Laplacian and nu grad^3 exchanges), three
    LSRK steps: direct == reference bit for bit, and cmdg_lsrk_run carries the freshness of the
    send buffer of Q from one step to the next."""
    from helpers import periodic_hyperdiffusion_setup
    from test_gpu_halo import _self_neighbour_grid
    grid = _self_neighbour_grid(cm, 0, 2)
    grid.nabrtorank = [0] * len(grid.nabrtorank)
    out, counts = [], []
    for reference in (False, True):
        law = periodic_hyperdiffusion_setup()[0]
        dg = cm.dgmodel.DGModel(law, grid, numerical_flux_first_order=1, direction=0)
        dg.comm_init_rccl(cm.dgmodel.rccl_unique_id(), 0, 1)
        if reference:
            dg.set_option(cm._lib.OPT_REFERENCE_HALO, 1)
        Q = dg.init_ode_state(0.0)
        solver = cm.odesolvers.LSRK54CarpenterKennedy(dg, Q, dt=1e-4)
        dg.profile_reset()
        dg.profile_enable(True)
        solver.dostep(Q, nsteps=3)
        dg.synchronize()
        dg.profile_enable(False)
        counts.append({k: dg.profile_get(k)[1] for k in ("PACK", "UNPACK", "TRANSPORT")})
        out.append(Q[:grid.nreal].cpu().numpy().copy())
        dg.close()
    assert np.abs(out[0]).max() > 0 and np.isfinite(out[0]).all()
    assert np.array_equal(out[0], out[1])
    assert counts[0]["PACK"] == 1, counts
    assert counts[1]["PACK"] == counts[1]["UNPACK"] > 1


def test_switching_between_one_stream_and_two_pipelines(cm, torch):
    """A gradient filter puts a partitioned handle on the reference's one-stream order (its exterior
    gradient launch leaves the gradient flux unfiltered in the send buffer, so that one is packed);
    taking the filter away switches to the two pipelines.  Steps with, without and again with the
    filter, two ranks: direct exchange == reference exchange bit for bit across the switches."""
    from helpers import pseudo1d_setup
    F = cm.mesh.filters
    out = []
    for reference in (False, True):
        dgs, Qs, grids, filts = [], [], [], []
        for r in range(2):
            law, grid, dt = pseudo1d_setup(direction=0, rank=r, size=2)
            dg = cm.dgmodel.DGModel(law, grid, direction=0)
            if reference:
                dg.set_option(cm._lib.OPT_REFERENCE_HALO, 1)
            q = dg.init_ode_state(0.0)
            q[grid.nreal:] = float("nan")
            filts.append(F.make_device_filter(dg, F.CutoffFilter(grid, 3),
                                              F.FilterIndices(range(1, law.ngradflux + 1)),
                                              nstate=law.ngradflux))
            dgs.append(dg), Qs.append(q), grids.append(grid)
        dQs = [x.create_state() for x in dgs]
        torch.cuda.synchronize()
        cm.dgmodel.connect_local(dgs)
        s = cm.odesolvers.LSRK54CarpenterKennedy(dgs[0], Qs[0], dt=dt)
        t = 0.0
        for filtered in (True, False, True, False):
            for dg, f in zip(dgs, filts):
                dg.set_filters(gradient_filter=f if filtered else None)
            assert dgs[0].query("HALO_PIPELINE") == (0 if filtered or reference else 1)
            cm.dgmodel.group_lsrk_run(dgs, Qs, dQs, t, dt, 2, s.RKA, s.RKB, s.RKC)
            t += 2 * dt
        for x in dgs:
            x.synchronize()
        out.append([q[:g.nreal].cpu().numpy().copy() for q, g in zip(Qs, grids)])
        for dg, f in zip(dgs, filts):
            dg.set_filters()
            f.close()
            dg.close()
    for a, b in zip(*out):
        assert np.isfinite(a).all()
        assert np.array_equal(a, b)


def test_step_graph_equals_eager_steps(cm, torch):
    """CMDG_OPT_STEP_GRAPH: cmdg_lsrk_run replays one captured step (stage times from device
    memory, advanced as updatetime! does) for every step but the first of a run.  Same bits as
    eager steps; two runs reuse the graph; the time really advances (the advection-diffusion
    law's boundary data depend on it).  A handle that exchanges through RCCL is recorded with the
    halo stream as the origin of the capture (csrc/cmdg.hip, graph_eligible): same bits again."""
    from helpers import pseudo1d_setup
    from test_gpu_halo import _self_neighbour_grid
    out, counts = [], []
    for graph in (0, 1):
        law, grid, dt = pseudo1d_setup(direction=0)
        dg = cm.dgmodel.DGModel(law, grid, direction=0)
        dg.set_option(cm._lib.OPT_STEP_GRAPH, graph)
        Q = dg.init_ode_state(0.0)
        solver = cm.odesolvers.LSRK54CarpenterKennedy(dg, Q, dt=dt)
        solver.dostep(Q, nsteps=4)
        solver.dostep(Q, nsteps=3)
        dg.synchronize()
        counts.append(dg.query("GRAPH_STEPS"))
        out.append(Q[:grid.nreal].cpu().numpy().copy())
        dg.close()
    assert counts == [0, 3 + 2], counts
    assert np.isfinite(out[0]).all() and np.abs(out[0]).max() > 0
    assert np.array_equal(out[0], out[1])
    # with neighbours (RCCL, the rank as its own neighbour): the groups are recorded as well
    out, counts = [], []
    for graph in (0, 1):
        grid = _self_neighbour_grid(cm, 0, 2)
        grid.nabrtorank = [0] * len(grid.nabrtorank)
        dg = cm.dgmodel.DGModel(pseudo1d_setup()[0], grid, direction=0)
        dg.comm_init_rccl(cm.dgmodel.rccl_unique_id(), 0, 1)
        dg.set_option(cm._lib.OPT_STEP_GRAPH, graph)
        assert dg.query("HALO_PIPELINE") == 1
        Q = dg.init_ode_state(0.0)
        solver = cm.odesolvers.LSRK54CarpenterKennedy(dg, Q, dt=1e-4)
        solver.dostep(Q, nsteps=4)
        solver.dostep(Q, nsteps=3)
        dg.synchronize()
        counts.append(dg.query("GRAPH_STEPS"))
        out.append(Q[:grid.nreal].cpu().numpy().copy())
        dg.close()
    assert counts == [0, 3 + 2], counts
    assert np.isfinite(out[0]).all() and np.abs(out[0]).max() > 0
    assert np.array_equal(out[0], out[1])


def test_step_graph_is_recorded_again_after_an_option_changes_the_launches(cm, torch):
    """A recorded step holds the kernels of the options it was recorded under.  Held-Suarez has no
    viscosity, so by default ``state_gradient_flux`` is not formed; CMDG_OPT_KEEP_GRADFLUX switches
    to the kernels that refresh it.  Setting it between two graph runs must drop the graph: the
    second run's gradient flux equals the eager handle's, bit for bit (a stale graph would leave
    it untouched)."""
    from helpers import held_suarez_setup
    res = []
    for graph in (0, 1):
        law, grid, direction, diffusion_direction = held_suarez_setup()
        dg = cm.dgmodel.DGModel(law, grid, direction=direction, diffusion_direction=diffusion_direction)
        dg.set_option(cm._lib.OPT_STEP_GRAPH, graph)
        Q = dg.init_ode_state(0.0)
        solver = cm.odesolvers.LSRK54CarpenterKennedy(dg, Q, dt=1.0)
        solver.dostep(Q, nsteps=3)
        dg.synchronize()
        assert float(dg.state_gradient_flux.abs().max()) == 0.0     # not formed
        n0 = dg.query("GRAPH_STEPS")
        dg.set_option(cm._lib.OPT_KEEP_GRADFLUX, 1)
        solver.dostep(Q, nsteps=3)
        dg.synchronize()
        res.append((Q[:grid.nreal].cpu().numpy().copy(),
                    dg.state_gradient_flux[:grid.nreal].cpu().numpy().copy(), n0, dg.query("GRAPH_STEPS")))
        dg.close()
    (Qe, ge, _, ne), (Qg, gg, n0, n1) = res
    assert ne == 0 and n0 == 2 and n1 == 4
    assert np.abs(ge).max() > 0 and np.array_equal(ge, gg) and np.array_equal(Qe, Qg)


def test_async_run_equals_the_callers_own_run(cm, torch):
    """CMDG_OPT_ASYNC_RUN: cmdg_lsrk_run hands the run to the handle's own thread and returns;
    the next entry point waits for it.  Same launches in the same order: same bits, with and
    without the recorded step, with and without neighbours (RCCL, the rank as its own peer);
    two runs queued back to back execute in order; an option set in between waits for them."""
    from helpers import pseudo1d_setup
    from test_gpu_halo import _self_neighbour_grid
    for neighbours in (False, True):
        out = []
        for asyn, graph in ((0, 0), (1, 0), (1, 1)):
            if neighbours:
                grid = _self_neighbour_grid(cm, 0, 2)
                grid.nabrtorank = [0] * len(grid.nabrtorank)
                dg = cm.dgmodel.DGModel(pseudo1d_setup()[0], grid, direction=0)
                dg.comm_init_rccl(cm.dgmodel.rccl_unique_id(), 0, 1)
                dt = 1e-4
            else:
                law, grid, dt = pseudo1d_setup(direction=0)
                dg = cm.dgmodel.DGModel(law, grid, direction=0)
            dg.set_option(cm._lib.OPT_STEP_GRAPH, graph)
            dg.set_option(cm._lib.OPT_ASYNC_RUN, asyn)
            Q = dg.init_ode_state(0.0)
            solver = cm.odesolvers.LSRK54CarpenterKennedy(dg, Q, dt=dt)
            solver.dostep(Q, nsteps=4)           # (returns at once: enqueued by the caller or handed over)
            solver.dostep(Q, nsteps=3)
            dg.set_option(cm._lib.OPT_KEEP_GRADFLUX, 0)      # an entry point in between: waits
            dg.synchronize()
            if graph:
                assert dg.query("GRAPH_STEPS") == 5
            out.append(Q[:grid.nreal].cpu().numpy().copy())
            dg.set_option(cm._lib.OPT_ASYNC_RUN, 0)
            dg.close()
        assert np.isfinite(out[0]).all() and np.abs(out[0]).max() > 0
        assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2])
