#!/usr/bin/env python3
"""DG-RHS DOF-update throughput of the MI355X-native hot path.

``python bench.py --gpus N --steps K --warmup W``.  For N > 1 the command starts its own N ranks
(one child process per GPU, before this process touches the GPU) unless it finds itself already
launched as a rank (``RANK`` in the environment, e.g. under ``python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N``).

One "step" is one LSRK54 time step = 5 fused (gradient pass + tendency/update pass)
evaluations of the DG right-hand side on the rank's elements.  ``value`` is the whole-job
DOF-updates/s: nodes x prognostic states x 5 stages x K steps / wall time (max over
ranks), with every input resident in HBM before the timed region.  Rank 0 prints ONE JSON
line (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

# The CPU baseline's OpenMP threads stay where their first touch put their pages.  libgomp reads
# these when it is loaded (with torch or with liboracle.so, whichever comes first), so they are set
# before either; nothing on the GPU path depends on them.
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")
# (with a binding policy libgomp pins the initial thread to the first place as soon as it is
# loaded: the hardware threads this process may use are read now, not later)
_AFFINITY_AT_START = len(os.sched_getaffinity(0))

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


_RESULT_OUT = None


def claim_stdout():
    """stdout carries the ONE JSON line of the contract, but RCCL (and any other native library)
    writes banners and warnings to file descriptor 1 as well.  A rank therefore keeps a private
    handle on the real stdout for its result and points descriptor 1 at stderr for everything
    else, native code included."""
    global _RESULT_OUT
    if _RESULT_OUT is None:
        sys.stdout.flush()
        _RESULT_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(obj):
    out = _RESULT_OUT or sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


def build_workload(cm, name, rank, size, ne, args, nhorz=None, nvert=None):
    """Returns (law, grid, direction, dt, description)."""
    M, BL = cm.mesh, cm.balancelaws
    if name == "advdiff-brick":
        # config 1 physics (pseudo1D_advection_diffusion.jl:293-368) on Ne^3 elements;
        # weak scaling: Ne x Ne columns per rank -> (Ne * sqrt-ish) handled by growing x1
        nx = ne * size
        rng = [np.linspace(-1, 1, nx + 1), np.linspace(-1, 1, ne + 1), np.linspace(-1, 1, ne + 1)]
        topl = M.StackedBrickTopology(rng, boundary=((1, 2),) * 3, periodicity=(False,) * 3,
                                      connectivity="full", rank=rank, size=size)
        grid = M.DiscontinuousSpectralElementGrid(topl, 4)
        n = np.ones(3) / np.sqrt(3)
        law = BL.AdvectionDiffusion(3, BL.Pseudo1D(n, 1.0, 1 / 100, -1 / 2, 1 / 10),
                                    (BL.InhomogeneousBC(0), BL.InhomogeneousBC(1)))
        dt = (1.0 / 4) / (max(nx, ne) * 16)
        desc = {"workload": "advection-diffusion Pseudo1D (BASELINE configs[0] physics), N=4, "
                            "%dx%dx%d brick elements, LSRK54, fp64" % (nx, ne, ne),
                "elements": nx * ne * ne, "nodes_per_element": 125, "states": law.ns,
                "parallelism": "element partition (Hilbert), %d rank(s)" % size}
        return law, grid, (0, 0), dt, desc
    if name == "heldsuarez":
        # BASELINE.json configs[2]: Held-Suarez dry GCM on the stacked cubed sphere, N = 4,
        # 8 levels, radii [a, a + 30 km] (experiments/AtmosGCM/heldsuarez.jl:174-240,
        # src/Driver/driver_configs.jl:344-470).  One GPU runs the configuration the metric is
        # quoted on, 6 x 30 x 30 x 8 = 43 200 elements.  --scaling weak (default) keeps that
        # work per GPU: n_horz = 30 / 42 / 60 / 85 at 1 / 2 / 4 / 8 GPUs; --scaling strong
        # keeps the 43 200-element sphere and splits it; --scaling weak-small is the round-1
        # family of ~5 400 elements per GPU (n_horz = 11 / 15 / 21 / 30).
        A = cm.atmos
        ps = A.PlanetParameters()
        n_horz = nhorz or args.nhorz or hs_nhorz(args.scaling, size)
        n_vert = nvert or args.nvert
        Rrange = np.linspace(ps.planet_radius, ps.planet_radius + 30e3, n_vert + 1)
        topl = M.StackedCubedSphereTopology(n_horz, Rrange, boundary=(1, 2), rank=rank, size=size,
                                            connectivity=getattr(args, "connectivity", "full"))
        grid = M.DiscontinuousSpectralElementGrid(topl, 4,
                                                  meshwarp=M.equiangular_cubed_sphere_warp)
        law = A.DryAtmosModel(A.HeldSuarezSetup(ps), orientation=A.ORIENT_SPHERICAL,
                              ref_state=A.DecayingTemperatureProfile(ps, 290.0, 220.0, 8e3),
                              viscosity=0.0, dynamic_viscosity=False,
                              hyperdiffusion_timescale=8 * 3600.0,
                              sources=A.SRC_GRAVITY | A.SRC_CORIOLIS | A.SRC_HELD_SUAREZ,
                              boundary_conditions=(A.BC_ATMOS_DEFAULT, A.BC_ATMOS_DEFAULT),
                              param_set=ps)
        nel = 6 * n_horz * n_horz * n_vert
        desc = {"workload": "Held-Suarez dry GCM (BASELINE configs[2]), stacked cubed sphere "
                            "6x%dx%dx%d elements, N=4, LSRK54 explicit, hyperdiffusion "
                            "(DryBiharmonic, horizontal), Rusanov, fp64" % (n_horz, n_horz, n_vert),
                "elements": nel, "nodes_per_element": 125, "states": law.ns,
                "parallelism": "element partition (Hilbert, whole columns), %d rank(s), "
                               "RCCL p2p halo" % size}
        return law, grid, (0, 1), 0.15, desc
    if name == "risingbubble":
        # BASELINE.json configs[1]: dry rising thermal bubble, N = 4, 20 x 20 x 20 = 8 000
        # elements of 500 m (experiments/TestCase/risingbubble.jl; the script's own mesh is
        # 20 x 1 x 20), SmagorinskyLilly, HydrostaticState(DryAdiabaticProfile), LSRK54 here
        # so that a "step" is the same five stages as the headline workload.
        A = cm.atmos
        ps = A.PlanetParameters()
        nx = ny = nz = args.ne if args.ne != 32 else 20
        ny *= size
        rng = [np.linspace(0.0, 500.0 * n, n + 1) for n in (nx, ny, nz)]
        topl = M.StackedBrickTopology(rng, periodicity=(True, True, False),
                                      boundary=((0, 0), (0, 0), (1, 2)), rank=rank, size=size)
        grid = M.DiscontinuousSpectralElementGrid(topl, 4)
        setup = A.RisingBubbleSetup(ps, xc=250.0 * nx, zc=100.0 * nz, rc=100.0 * nx)
        law = A.DryAtmosModel(setup, orientation=A.ORIENT_FLAT,
                              ref_state=A.DryAdiabaticProfile(ps, 300.0, 0.0),
                              smagorinsky=ps.C_smag, sources=A.SRC_GRAVITY,
                              boundary_conditions=(A.BC_ATMOS_DEFAULT, A.BC_ATMOS_DEFAULT),
                              param_set=ps)
        desc = {"workload": "dry rising bubble (BASELINE configs[1]), stacked brick %dx%dx%d "
                            "elements, N=4, SmagorinskyLilly, LSRK54 explicit, Rusanov, fp64"
                            % (nx, ny, nz),
                "elements": nx * ny * nz, "nodes_per_element": 125, "states": law.ns,
                "parallelism": "element partition (Hilbert, whole columns), %d rank(s)" % size}
        return law, grid, (0, 0), 0.1, desc
    if name == "bomex":
        # BASELINE.json configs[3]: BOMEX moist LES at N = 6, about 65 k elements on 8 GPUs
        # (experiments/AtmosLES/bomex_les.jl + bomex_model.jl: 6.4 km x 6.4 km x 3 km, periodic
        # in x and y, EquilMoist, SmagorinskyLilly(0.23), BOMEX sources and surface fluxes);
        # per GPU: ne x ne x 2 ne elements of (200 m, 200 m, 3000 / (2 ne) m), weak scaling in y.
        MO = cm.moist
        ne = args.bomex_ne
        nx, ny, nz = ne, ne * size, 2 * ne
        rng = [np.linspace(0.0, 200.0 * nx, nx + 1), np.linspace(0.0, 200.0 * ny, ny + 1),
               np.linspace(0.0, 3000.0, nz + 1)]
        topl = M.StackedBrickTopology(rng, periodicity=(True, True, False),
                                      boundary=((0, 0), (0, 0), (1, 2)), rank=rank, size=size)
        grid = M.DiscontinuousSpectralElementGrid(topl, getattr(args, "bomex_order", 6))
        law = MO.bomex_model(3000.0)
        desc = {"workload": "BOMEX moist LES (BASELINE configs[3]), stacked brick %dx%dx%d elements, "
                            "N=%d, EquilMoist (saturation adjustment in every flux evaluation, as "
                            "in the reference snapshot), SmagorinskyLilly, BOMEX sources and "
                            "surface fluxes, LSRK54 explicit, Rusanov, fp64"
                            % (nx, ny, nz, getattr(args, "bomex_order", 6)),
                "elements": nx * ny * nz,
                "nodes_per_element": (getattr(args, "bomex_order", 6) + 1) ** 3, "states": law.ns,
                "parallelism": "element partition (Hilbert, whole columns), %d rank(s)" % size}
        return law, grid, (0, 0), 0.004, desc
    raise SystemExit("unknown workload %s" % name)


def hs_nhorz(scaling, size):
    if scaling in ("strong", "both"):
        return 30
    if scaling == "weak-small":
        return {1: 11, 2: 15, 3: 18, 4: 21, 5: 24, 6: 26, 7: 28, 8: 30}.get(
            size, int(round(30 * (size / 8) ** 0.5)))
    return {1: 30, 2: 42, 3: 52, 4: 60, 5: 67, 6: 73, 7: 79, 8: 85}.get(
        size, int(round(30 * size ** 0.5)))


def algorithmic_bytes_per_node(law, kernel, Nq=5):
    """every distinct array element a pass needs moves once;
    face tables add F = (5*8 + 2*8) * 6 * Nfp / Np = 336 / Nq B per node (67 at N = 4)."""
    b, F = 8, int(round(336 / Nq))
    ns, naux, ngf, ngl, nhyp = law.ns, law.naux, law.ngradflux, law.ngradlap, law.nhyper
    if kernel == "GRADIENTS":
        return b * (ns + naux + 9 + ngf + 3 * ngl) + F
    if kernel == "DIVGRAD":
        return b * (3 * ngl + 11 + ngl) + F
    if kernel == "GRADLAP":
        return b * (ngl + ns + naux + 9 + nhyp) + F
    if kernel == "TENDENCY":   # + fused LSRK: dQ read/write, Q write
        return b * (ns + naux + ngf + nhyp + 11 + 2 * ns + ns) + F
    raise KeyError(kernel)


def kernel_info(dg, law, grid, direction):
    """What the handle's instantiation reads and writes (cmdg_query), for needed_bytes_per_node."""
    Nq, Nqv = grid.N[0] + 1, grid.N[-1] + 1
    return {"ns": law.ns, "naux": law.naux, "ngf": law.ngradflux, "ngl": law.ngradlap,
            "nhyp": law.nhyper, "Nq": Nq, "Nqv": Nqv,
            "direction": int(direction[0]), "diffusion_direction": int(direction[1]),
            "gf_live": bool(dg.query("GRADFLUX_LIVE")),
            "law_gf": bool(dg.query("LAW_NEEDS_GRADFLUX")),
            "nder": dg.query("NDERIVED"),
            "q_read": [dg.query(("STATE_READ", p)) for p in range(4)],
            "aux_read": [dg.query(("AUX_READ", p)) for p in range(4)],
            "nupd_fused": dg.query("NUPDATED_AUX") if dg.query("FUSED_UPDATE_AUX") else 0}


def needed_bytes_per_node(info, kernel):
    """HBM bytes per node one launch of the SHIPPED instantiation needs, every distinct array
    element counted once (a face neighbour's value is some work-group's own volume node):
    what the kernel reads and writes in csrc/kernels.h, not SURVEY's generic formula --
    a law whose second-order flux never reads state_gradient_flux (zero viscosity) has those
    columns neither formed nor read; of Q and the auxiliary state a pass reads the columns its
    pointwise functions use (the law declares them: P::state_read / P::aux_read, e.g. 6 of the
    dry atmosphere's 17 auxiliary columns in the tendency pass; every column for a law that does
    not say); a pass differentiating in one direction reads only that direction's metric rows;
    the face tables are the digested ones (faceP int32 + faceG 4 doubles = 36 B per face node
    instead of 56)."""
    b = 8
    Nq, Nqv = info["Nq"], info["Nqv"]
    Np, nft = Nq * Nq * Nqv, 4 * Nq * Nqv + 2 * Nq * Nq
    F = 36.0 * nft / Np

    def metric_rows(d):          # rows of d xi / d x the pass reads: 3 per differentiated axis
        return {0: 9, 1: 6, 2: 3}[d]
    ns, ngf, ngl, nhyp = info["ns"], info["ngf"], info["ngl"], info["nhyp"]
    qr = info.get("q_read") or [ns] * 4
    ar = info.get("aux_read") or [info["naux"]] * 4
    dd, dm = info["diffusion_direction"], info["direction"]
    if kernel == "GRADIENTS":    # reads Q, aux, metric, MI (faces); writes the fused aux refresh,
        #                          the gradient flux if anybody reads it, the hyperdiffusion gradients
        return b * (qr[0] + ar[0] + metric_rows(dd) + 1 + info["nupd_fused"]
                    + (ngf if info["gf_live"] else 0) + 3 * ngl) + F
    if kernel == "DIVGRAD":      # reads the gradients, M, MI, metric; writes ngl Laplacians
        return b * (3 * ngl + 2 + metric_rows(dd) + ngl) + F
    if kernel == "GRADLAP":      # reads the Laplacians, Q, aux, metric, MI; writes nhyp columns
        return b * (ngl + qr[2] + ar[2] + metric_rows(dd) + 1 + nhyp) + F
    if kernel == "TENDENCY":     # + fused LSRK: dQ read and written, Q written
        return b * (qr[3] + ar[3] + (ngf if info["law_gf"] else 0) + nhyp + 2 + metric_rows(dm)
                    + info["nder"] + 3 * ns) + F
    raise KeyError(kernel)


def _oracle_steps(O, law, grid, direction, dt, budget_s, max_steps, warm_rhs=True):
    dg = O.OracleDGModel(law, grid, nf_first=0, direction=direction[0],
                         diffusion_direction=direction[1])
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    dQ = np.zeros_like(Q)
    dg.numa_distribute()                         # first touch by the thread that owns the elements
    Q, dQ = O.first_touch(Q), O.first_touch(dQ)
    if warm_rhs:                                 # page faults + thread pool: one RHS evaluation
        dg(O.first_touch(np.zeros_like(Q)), O.first_touch(Q), 0.0, 1.0, 0.0)
    n, t0 = 0, time.time()
    while True:
        O.lsrk54_step(dg, Q, dQ, n * dt, dt)
        n += 1
        el = time.time() - t0
        if el > budget_s or n >= max_steps:
            break
    return grid.nreal * grid.Np * law.ns * 5 * n / el, n, el


def host_cpu_info():
    """What this process may use of the host: hardware threads in its affinity mask, the cgroup
    CPU quota (a container may see 128 threads and be allowed 16 CPUs' worth of time), sockets
    and physical cores behind the mask."""
    info = {"affinity_threads": _AFFINITY_AT_START, "cgroup_cpu_quota": None,
            "sockets": None, "physical_cores": None}
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            info["cgroup_cpu_quota"] = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                info["cgroup_cpu_quota"] = q / per
        except Exception:
            pass
    try:
        cores, cpu = set(), {}
        for line in open("/proc/cpuinfo"):
            if ":" not in line:
                if "processor" in cpu:
                    cores.add((cpu.get("physical id", "0"), cpu.get("core id", cpu["processor"])))
                cpu = {}
                continue
            k, v = line.split(":", 1)
            cpu[k.strip()] = v.strip()
        info["sockets"] = len({c[0] for c in cores}) or None
        info["physical_cores"] = len(cores) or None
    except Exception:
        pass
    return info


def baseline_threads(info, omp_max):
    """Threads the all-core figure runs on: one per hardware thread this process may keep busy."""
    n = min(omp_max, info["affinity_threads"])
    if info["cgroup_cpu_quota"]:
        n = min(n, max(1, int(info["cgroup_cpu_quota"] + 0.5)))
    return max(1, n)


def cpu_baseline(cm, law, grid, direction, dt, budget_s, args):
    """The oracle (CPU restatement of the reference kernels in the reference's unfused launch
    order: horizontal then vertical volume kernel, per-direction interface launches, separate
    auxiliary pass, separate update!) timed on this host's cores: all of them on the workload
    itself, and one core on a smaller sphere of the same workload (a full-size step would take
    minutes on one core)."""
    from oracle import oracle as O
    O.build()
    host = host_cpu_info()
    cores = baseline_threads(host, O.get_max_threads())
    O.set_num_threads(cores)
    what = "the same workload"
    if args.workload == "bomex" and args.bomex_ne > 16:
        # bounded sample: one oracle step of the 65 536-element box takes minutes on the host
        import copy
        a2 = copy.copy(args)
        a2.bomex_ne = 16
        law, grid, direction, dt, _ = build_workload(cm, "bomex", 0, 1, 0, a2)
        what = "the same law and order on 16x16x32 elements"
    v, n, el = _oracle_steps(O, law, grid, direction, dt, budget_s, 50)
    out = {"value": v, "unit": "DOF-updates/s", "cores": cores, "kind": "port",
           "sample": "%d LSRK54 step(s) of %s (%d elements) in %.1f s, "
                     "OpenMP over elements, after one untimed RHS evaluation" % (n, what, grid.nreal, el),
           # how the threads were placed, and the memory-bandwidth ceiling they reach together
           "threads": cores, "host": host,
           "omp": {k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES")},
           "first_touch": "every per-element array re-homed by an OpenMP static loop over elements",
           "stream_triad_GBs": round(O.stream_triad_gbs(), 1)}
    if args.workload == "heldsuarez":
        law1, grid1, dir1, dt1, _ = build_workload(cm, "heldsuarez", 0, 1, 0, args, nhorz=4)
        O.set_num_threads(1)
        try:
            v1, n1, el1 = _oracle_steps(O, law1, grid1, dir1, dt1, budget_s, 50)
        finally:
            O.set_num_threads(cores)
        out["one_core"] = {"value": v1, "unit": "DOF-updates/s", "cores": 1,
                           "sample": "%d LSRK54 step(s) of the same workload on a 6x4x4x%d sphere "
                                     "(%d elements) in %.1f s, one thread" % (
                                         n1, args.nvert, grid1.nreal, el1)}
    return out


def parity_check(cm, args, dev):
    """GPU vs oracle on the same workload at a size the oracle finishes in seconds, outside the
    timed region: one RHS evaluation (tendency, per prognostic state, L-inf relative) and two
    LSRK54 steps (state).  A failure aborts the benchmark: a fast wrong answer is no answer."""
    import torch
    from oracle import oracle as O
    O.build()
    if args.workload != "heldsuarez":
        return None
    law, grid, direction, dt, _ = build_workload(cm, args.workload, 0, 1, 4, args, nhorz=3, nvert=2)
    dg = cm.dgmodel.DGModel(law, grid, direction=direction[0], diffusion_direction=direction[1],
                            device=dev)
    odg = O.OracleDGModel(law, grid, nf_first=0, direction=direction[0],
                          diffusion_direction=direction[1])
    Q0 = law.init_state_prognostic(grid, odg.state_auxiliary, 0.0)
    rng = np.random.default_rng(20250117)
    Q0[:, 1:4] += 0.5 * rng.standard_normal(Q0[:, 1:4].shape)
    Q0[:, 4] *= 1 + 1e-3 * rng.standard_normal(Q0[:, 4].shape)
    To = np.zeros_like(Q0)
    odg(To, Q0.copy(), 0.0, 1.0, 0.0)
    Q = torch.from_numpy(Q0.copy()).to(dev)
    Tg = dg.create_state()
    torch.cuda.synchronize()
    dg(Tg, Q, 0.0, 1.0, 0.0)
    Tg = Tg.cpu().numpy()
    nr = grid.nreal
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    tend = [rel(Tg[:nr, s], To[:nr, s]) for s in range(law.ns)]
    solver = cm.odesolvers.LSRK54CarpenterKennedy(dg, Q, dt=dt)
    solver.dostep(Q, nsteps=2)
    dg.synchronize()
    Qo, dQo = Q0.copy(), np.zeros_like(Q0)
    for s in range(2):
        O.lsrk54_step(odg, Qo, dQo, s * dt, dt)
    Qg = Q.cpu().numpy()
    state = [rel(Qg[:nr, s], Qo[:nr, s]) for s in range(law.ns)]
    dg.close()
    tol = 1e-12
    out = {"against": "oracle (CPU restatement), same inputs, 6x3x3x2 sphere",
           "tendency_rel_linf": max(tend), "state_rel_linf_2_steps": max(state), "tolerance": tol,
           "ok": bool(max(tend) < tol and max(state) < tol)}
    if not out["ok"]:
        raise SystemExit("bench.py: GPU != oracle: %s" % json.dumps(out))
    return out


def timed_run(cm, dg, law, grid, dt, steps, warmup, sync_all, distributed, dev):
    """W warm-up steps, then K timed steps bracketed by barrier + synchronize."""
    import torch
    import torch.distributed as dist
    Q = dg.init_ode_state(0.0)
    solver = cm.odesolvers.LSRK54CarpenterKennedy(dg, Q, dt=dt)
    solver.dostep(Q, nsteps=warmup)
    sync_all()
    t0 = time.perf_counter()
    solver.dostep(Q, nsteps=steps)
    dg.synchronize()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if distributed:
        dist.barrier()
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    return Q, solver, el


PMC_FILES = {("heldsuarez", 43200): "r04_heldsuarez_n30_pmc_hbm_per_launch.json",
             ("risingbubble", 8000): "r04_risingbubble_8000_pmc_hbm_per_launch.json",
             ("bomex", 8192): "r04_bomex_n6_8192_pmc_hbm_per_launch.json"}


def kernel_source_digest():
    """Digest of the kernel sources: a committed PMC profile is quoted only for the kernels it was
    taken on (scripts/pmc_any.sh stores the digest next to the counters)."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "climatemachine.jl_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        # what the pass kernels are compiled from: the kernels, the laws, the flags
        if name in ("kernels.h", "cmdg_common.h", "Makefile") or name.startswith("physics_"):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(workload, nreal, dom):
    """HBM-side traffic of the dominant kernel from the committed rocprofv3 --pmc passes of this
    same command (scripts/pmc_any.sh; FETCH_SIZE and WRITE_SIZE in separate passes, KB units,
    FETCH_SIZE doubled as the gfx950 guide prescribes and as profiles/r01_pmc_calibration_n30.json
    confirms for this library's 8-byte-per-lane loads).  Returned only when the profile was taken
    on the kernel sources of this tree."""
    name = PMC_FILES.get((workload, nreal))
    if not name:
        return None, None
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, None
    rec = json.load(open(path))
    if rec.get("kernel_source_digest") != kernel_source_digest():
        return None, "profiles/%s is stale (taken on other kernel sources)" % name
    pm = rec.get("k_%s" % dom.lower())
    if not (pm and "FETCH_SIZE" in pm and "WRITE_SIZE" in pm):
        return None, None
    return 1024.0 * (2.0 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"]), "profiles/" + name


def measure(cm, args, workload_args, rank, world, distributed, dev, with_halo):
    """Builds one workload on this rank, runs the timed region and the event pass, and (rank 0)
    returns the fields of the JSON line for it."""
    import torch
    import torch.distributed as dist
    t0 = time.time()
    law, grid, direction, dt, desc = build_workload(cm, args.workload, rank, world, args.ne,
                                                    workload_args)
    log("[rank %d] mesh+grid: %d real + %d ghost elements in %.1f s" % (
        rank, grid.nreal, grid.nelem - grid.nreal, time.time() - t0))
    dg = cm.dgmodel.DGModel(law, grid, direction=direction[0],
                            diffusion_direction=direction[1], device=dev)
    if distributed:
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            uid = torch.frombuffer(bytearray(cm.dgmodel.rccl_unique_id()), dtype=torch.uint8).clone()
        uid = uid.to(dev)
        dist.broadcast(uid, 0)
        dg.comm_init_rccl(uid.cpu().numpy().tobytes(), rank, world)
        dg.comm_selftest()
    # opt-in: one recorded step replayed (CMDG_OPT_STEP_GRAPH) / enqueued by the handle's own thread
    # (CMDG_OPT_ASYNC_RUN).  The default stays eager for handles that exchange: the capture with
    # RCCL groups in it has only ever run with a rank as its own peer.
    if args.step_graph:
        dg.set_option(cm._lib.OPT_STEP_GRAPH, 1)
    if args.async_run:
        dg.set_option(cm._lib.OPT_ASYNC_RUN, 1)
    step_filter = None
    if args.filter:
        assert args.workload in ("heldsuarez", "bomex")
        F = cm.mesh.filters
        if args.workload == "bomex":
            # the experiment's every-step callback (bomex_les.jl:97-117): TMAR filter of q_tot
            step_filter = F.make_device_filter(dg, F.TMARFilter(), F.FilterIndices(6))
            desc = dict(desc, step_filter="TMARFilter on moisture.rho q_tot")
        else:
            step_filter = F.make_device_filter(dg, F.ExponentialFilter(grid, 0, 20),
                                               F.AtmosFilterPerturbations(law))
            desc = dict(desc, step_filter="ExponentialFilter(grid, 0, 20) on AtmosFilterPerturbations")
        dg.set_filters(step_filter=step_filter)

    def sync_all():
        dg.synchronize()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()

    # ---- timed region: K steps, nothing but the library's own launches in flight ----------
    Q, solver, el = timed_run(cm, dg, law, grid, dt, args.steps, args.warmup, sync_all,
                              distributed, dev)
    if distributed:
        nn = torch.tensor([grid.nreal], device=dev, dtype=torch.int64)
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
        total_elems = int(nn.item())
    else:
        total_elems = grid.nreal
    # ---- same K steps again with a HIP event pair around every kernel launch (recorded on
    # the launch stream): per-kernel durations for the roofline.  The events themselves cost
    # ~10 % of a step at this size, hence the separate pass.
    dg.profile_reset()
    if not args.no_events:
        dg.profile_enable(True)
    sync_all()
    t1 = time.perf_counter()
    solver.dostep(Q, nsteps=args.steps)
    dg.synchronize()
    torch.cuda.synchronize()
    el_events = time.perf_counter() - t1
    dg.profile_enable(False)
    finite = bool(torch.isfinite(Q[:grid.nreal]).all().item())
    out = None
    if rank == 0:
        dofs = total_elems * grid.Np * law.ns * 5 * args.steps
        kern = {}
        ext = {}
        for k in ("GRADIENTS", "DIVGRAD", "GRADLAP", "TENDENCY", "PACK", "UNPACK", "FILTER"):
            ms, n = dg.profile_get(k)
            if k + "_EXT" in cm._lib.CMDG_K:          # interior + exterior launches of the pass
                ms_e, n_e = dg.profile_get(k + "_EXT")
                if n_e:
                    ext[k] = {"interior_avg_ms": ms / max(n, 1), "exterior_avg_ms": ms_e / n_e}
                ms, n = ms + ms_e, n + n_e
            if n:
                kern[k] = (ms / n, n)
        out = {"value": dofs / el, "ms_per_step": 1e3 * el / args.steps, "config": desc,
               "node_updates_per_s": dofs / el / law.ns, "state_finite": finite,
               "elements_total": total_elems,
               "kernels_ms": {k: {"avg_ms": v[0], "launches": v[1]} for k, v in kern.items()}}
        # partitioned runs: rank 0's ghost exchange as the event pass saw it -- the RCCL group of
        # each exchange on the halo stream, and the time the compute stream had nothing left to
        # do but wait for the exterior pipeline / an exchange (zero when hidden)
        if with_halo:
            halo = {"rank": 0, "real_elements": int(grid.nreal),
                    "ghost_elements": int(grid.nelem - grid.nreal),
                    "interior_elements": int(len(grid.interiorelems)),
                    "exterior_elements": int(len(grid.exteriorelems)),
                    "neighbours": [int(r) for r in grid.nabrtorank],
                    "send_nodes": int(len(grid.vmapsend)),
                    "exchange": {k.lower(): dg.query(k) for k in
                                 ("DIRECT_SEND", "DIRECT_RECV", "HALO_PIPELINE")},
                    "step_graph": bool(args.step_graph), "async_run": bool(args.async_run),
                    "graph_steps_replayed": int(dg.query("GRAPH_STEPS"))}
            for key, name in (("TRANSPORT", "rccl_group"), ("HALO_EXPOSED", "exposed")):
                ms, n = dg.profile_get(key)
                if n:
                    halo[name + "_avg_us"] = 1e3 * ms / n
                    halo[name + "_ms_per_step"] = ms / args.steps
                    halo["exchanges_per_step"] = n / args.steps
            halo["passes"] = ext
            out["halo"] = halo
        passes = [k for k in kern if k in ("GRADIENTS", "DIVGRAD", "GRADLAP", "TENDENCY")]
        if passes:
            info = kernel_info(dg, law, grid, direction)
            dom = max(passes, key=lambda k: kern[k][0] * kern[k][1])
            avg_ms, nl = kern[dom]
            elems_per_launch = grid.nreal * 5 * args.steps / nl   # interior/exterior launches split
            nodes = grid.Np * elems_per_launch
            needed = needed_bytes_per_node(info, dom)
            achieved = needed * nodes / (avg_ms * 1e-3) / 1e9
            traffic, traffic_src = (None, None)
            if world == 1 and not args.filter:
                traffic, traffic_src = committed_traffic(args.workload, grid.nreal, dom)
            survey = algorithmic_bytes_per_node(law, dom, grid.N[0] + 1)
            out["roofline"] = {
                "bound": "hbm", "kernel": "k_%s" % dom.lower(), "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_over_needed": (traffic / (needed * nodes)) if traffic else None,
                "traffic_source": ("committed rocprofv3 --pmc profile of this command on these "
                                   "kernel sources, not measured in this run: " + traffic_src)
                if traffic else traffic_src,
                "algorithmic_bytes_per_launch": needed * nodes,
                "algorithmic_bytes_per_node": needed,
                "bytes_model": "what the shipped instantiation reads and writes, each distinct "
                               "array element once (bench.needed_bytes_per_node)",
                "survey_formula_bytes_per_node": survey,
                "frac_on_survey_formula": survey * nodes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "avg_launch_ms": avg_ms,
                "per_kernel": {k: {"needed_bytes_per_node": needed_bytes_per_node(info, k),
                                   "frac": needed_bytes_per_node(info, k) * grid.Np * grid.nreal * 5
                                   * args.steps / kern[k][1] / (kern[k][0] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                               for k in passes},
                "timing": "HIP events on the launch stream, second pass of the same "
                          "%d steps (%.3f ms/step with events)" % (
                              args.steps, 1e3 * el_events / args.steps)}
            # flops of the (N+1)-point derivative contractions alone (2 (N+1) per
            # node, differentiated scalar and direction): gradient arguments, the two hyperdiffusion
            # passes, the tendency.  Secondary figure: the path is HBM bound (< 1 flop/B), the
            # contraction runs on fp64 VALU out of LDS, not on MFMA.
            cf = 6 * (grid.N[0] + 1) * (law.ngrad + 2 * law.ngradlap + law.ns)
            out["contraction"] = {
                "flops_per_node_update": cf,
                "achieved_TFLOPs": cf * (dofs / law.ns) / el / 1e12,
                "unit": "fp64 VALU out of LDS (v_mul_f64 + v_add_f64, no contraction of a*b+c)",
                # the shipped kernels issue no MFMA; measured on this chip (profiles/r02_*):
                "mfma_util": 0.0,
                "measured_fp64_peaks_TFLOPs": {"v_mfma_f64_16x16x4_f64": 47.7, "v_mfma_f64_4x4x4_4b_f64": 72.7,
                                               "v_fma_f64": 64.8, "v_mul_f64+v_add_f64": 34.0},
                "mfma_variant": "k_gradients with the contraction on v_mfma_f64_16x16x4_f64: +70 % time, "
                                "5.1 % matrix-pipe busy (SQ_VALU_MFMA_BUSY_CYCLES); contraction removed "
                                "altogether: -2.1 % per step",
                "evidence": ["profiles/r02_fp64_peak.jsonl", "profiles/r02_ab_mfma_contraction.txt",
                             "profiles/r02_mfma_variant_pmc_per_launch.json"]}
        if "FILTER" in kern:
            # Q read + written (5 fields each) and the two reference-state columns
            fb = 8 * (2 * law.ns + 2) * grid.Np * grid.nreal
            out["filter_kernel"] = {"avg_launch_ms": kern["FILTER"][0],
                                    "algorithmic_bytes_per_node": 8 * (2 * law.ns + 2),
                                    "achieved_GBs": fb / (kern["FILTER"][0] * 1e-3) / 1e9}
        out["_cpu_inputs"] = (law, grid, direction, dt)
    if step_filter is not None:
        dg.set_filters()
        step_filter.close()
    dg.close()
    return out


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """``--gpus N`` without a launcher: start one child process per GPU with the environment
    torch.distributed.run would give it.  This process has not touched the GPU (no torch import,
    no HIP call) and never does; rank 0's stdout -- the one JSON line -- is relayed, everything
    else goes to stderr; the exit code is non-zero if any rank's is."""
    import subprocess
    port = int(os.environ.get("MASTER_PORT") or free_port())
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      stderr=sys.stderr))
    deadline = time.time() + 3600
    try:
        line = procs[0].communicate(timeout=3600)[0].decode()
    except subprocess.TimeoutExpired:
        line = ""
    codes = []
    for p_ in procs:
        try:
            codes.append(p_.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:
            codes.append(-9)
    if any(codes):
        for p_ in procs:          # exactly the children started above
            if p_.poll() is None:
                p_.kill()
        log("bench.py: rank exit codes %s" % codes)
        sys.stdout.write(line)
        sys.stdout.flush()
        raise SystemExit(next(c for c in codes if c) or 1)
    sys.stdout.write(line)
    sys.stdout.flush()


def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="heldsuarez",
                    choices=["heldsuarez", "advdiff-brick", "risingbubble", "bomex",
                             "ocean-split-explicit"])
    ap.add_argument("--ne", type=int, default=32, help="advdiff-brick: elements per side per rank")
    ap.add_argument("--bomex-ne", type=int, default=16,
                    help="bomex: ne x ne x 2 ne elements per rank (32: the 65 536 elements of configs[3])")
    ap.add_argument("--bomex-order", type=int, default=6,
                    help="bomex: polynomial order (6: configs[3]; 4 for kernel-shape comparisons)")
    ap.add_argument("--ocean-nx", type=int, default=48, help="ocean-split-explicit: horizontal elements per side")
    ap.add_argument("--ocean-nz", type=int, default=16, help="ocean-split-explicit: vertical elements")
    ap.add_argument("--nhorz", type=int, default=0, help="heldsuarez: elements per cube edge")
    ap.add_argument("--scaling", default="both", choices=["both", "weak", "strong", "weak-small"],
                    help="heldsuarez on N > 1 GPUs: strong = the 6x30x30x8 sphere of BASELINE "
                         "configs[2] split over the GPUs, weak = 43 200 elements per GPU (n_horz "
                         "30/42/60/85 at 1/2/4/8 GPUs), weak-small = ~5 400 elements per GPU (n_horz "
                         "11/15/21/30); both (default) = strong as the headline and weak under "
                         "'secondary' of the same line.  On one GPU all but weak-small are the "
                         "same 43 200-element sphere")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run GPU vs oracle check")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary measurement (one GPU: n_horz = 11; N GPUs: the weak family)")
    ap.add_argument("--no-n1-reference", action="store_true",
                    help="N > 1: skip rank 0's one-GPU run of the 43 200-element sphere that "
                         "efficiency_vs_n1 is computed against")
    ap.add_argument("--nvert", type=int, default=8, help="heldsuarez: vertical elements")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU baseline")
    ap.add_argument("--filter", action="store_true",
                    help="heldsuarez: exponential filter of the perturbations after every step "
                         "(experiments/AtmosGCM/heldsuarez.jl:261-272); off for the headline "
                         "metric, which is the RHS + LSRK path alone")
    ap.add_argument("--step-graph", action="store_true",
                    help="record one LSRK step into a HIP graph and replay it (CMDG_OPT_STEP_GRAPH)")
    ap.add_argument("--async-run", action="store_true",
                    help="the handle's own thread enqueues the run (CMDG_OPT_ASYNC_RUN)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="time without per-kernel HIP events")
    ap.add_argument("--dry-launch", action="store_true",
                    help="start the ranks as usual, but each only prints its launch environment "
                         "(no torch, no GPU): checks the launcher itself")
    return ap.parse_args(argv)


def with_scaling(args, scaling):
    import copy
    a = copy.copy(args)
    a.scaling = scaling
    return a


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        return launch_ranks(args, argv)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_launch:
        rec = {"rank": rank, "world": world, "local_rank": local,
               "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")),
               "gpus": args.gpus, "torch_imported": "torch" in sys.modules}
        if rank == 0:
            emit({"dry_launch": rec})
        else:
            log("[dry-launch] %s" % json.dumps(rec))
        if os.environ.get("BENCH_DRY_FAIL_RANK") == str(rank):
            raise SystemExit(3)
        return
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    claim_stdout()
    if args.workload == "ocean-split-explicit":
        return main_ocean(args, rank, world, local)

    import torch
    import torch.distributed as dist
    from cmdg_loader import cm

    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    # launched as a rank (even a single one): take the distributed path, so that a 1-rank
    # launch rehearses process-group + RCCL set-up on a single-GPU box
    distributed = launched
    gloo = None
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on stdout at NCCL_DEBUG=VERSION; stdout carries the one
        # JSON line of the contract
        if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "WARN"
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device(dev))
        gloo = dist.new_group(backend="gloo")    # long waits without a spinning RCCL kernel

    parity = None
    if rank == 0 and world == 1 and not args.no_parity and not args.no_cpu:
        parity = parity_check(cm, args, dev)
        log("[parity] %s" % json.dumps(parity))

    hs = args.workload == "heldsuarez"
    # BENCH_REHEARSE_NRANK=1: take the N > 1 path (both families, the one-GPU reference, the CPU
    # barrier) with whatever world size this is -- a one-rank rehearsal on a single-GPU box
    multi = world > 1 or bool(os.environ.get("BENCH_REHEARSE_NRANK"))
    both = hs and multi and args.scaling == "both" and not args.nhorz
    head_scaling = "strong" if (hs and args.scaling in ("both", "strong")) else (
        args.scaling if hs else "weak")
    main_args = with_scaling(args, "strong" if head_scaling == "strong" else args.scaling)
    res = measure(cm, args, main_args, rank, world, distributed, dev,
                  with_halo=multi or bool(os.environ.get("BENCH_HALO_BLOCK")))
    sec = None
    if both and not args.no_secondary:
        sec = measure(cm, args, with_scaling(args, "weak"), rank, world, distributed, dev, with_halo=True)
    n1 = None
    if multi and distributed and hs and not args.no_n1_reference and not args.nhorz:
        # the one-GPU value both families are scaled against, measured by rank 0 alone on its GPU
        # while the others wait (CPU barrier): the 43 200-element sphere of BASELINE configs[2]
        if rank == 0:
            n1 = measure(cm, args, with_scaling(args, "strong"), 0, 1, False, dev, with_halo=False)
        dist.barrier(group=gloo)

    if rank == 0:
        law, grid, direction, dt = res.pop("_cpu_inputs")
        out = {"metric": "DG RHS DOF-updates/sec", "value": res.pop("value"), "unit": "DOF-updates/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": res.pop("ms_per_step"), "higher_is_better": True,
               "scaling": "strong" if head_scaling == "strong" else "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
        out.update(res)
        if parity is not None:
            out["parity"] = parity
        if n1 is not None:
            n1.pop("_cpu_inputs")
            out["n1_reference"] = {"value": n1["value"], "ms_per_step": n1["ms_per_step"],
                                   "elements": n1["elements_total"],
                                   "note": "rank 0 alone, same steps / warm-up, no neighbours"}
            out["efficiency_vs_n1"] = out["value"] / (world * n1["value"])
        if sec is not None:
            sec.pop("_cpu_inputs")
            s2 = {"scaling": "weak", "value": sec["value"], "ms_per_step": sec["ms_per_step"],
                  "workload": sec["config"]["workload"], "elements": sec["elements_total"],
                  "halo": sec.get("halo"), "kernels_ms": sec["kernels_ms"],
                  "roofline": sec.get("roofline")}
            if n1 is not None:
                s2["efficiency_vs_n1"] = sec["value"] / (world * n1["value"])
            out["secondary"] = s2
        elif (world == 1 and hs and not args.nhorz and not args.filter
                and args.scaling != "weak-small" and not args.no_secondary):
            # the round-1 headline size (SURVEY section 8(d)'s weak-scaling base of ~5 400 elements
            # per GPU), so that numbers of that family stay comparable
            a2 = with_scaling(args, "weak-small")
            r2 = measure(cm, args, a2, 0, 1, False, dev, with_halo=False)
            out["secondary"] = {"workload": r2["config"]["workload"], "elements": r2["elements_total"],
                                "value": r2["value"], "ms_per_step": r2["ms_per_step"]}
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(cm, law, grid, direction, dt, args.cpu_budget, args)
        emit(out)
        if not out["state_finite"]:
            raise SystemExit("bench.py: the state is not finite after the timed steps")
    if distributed:
        dist.barrier(group=gloo)
        dist.destroy_process_group()


def ocean_setup(cm, Nx, Nz, rank=0, size=1, connectivity=None):
    """BASELINE configs[4]: hydrostatic Boussinesq ocean box with the split-explicit stepper
    (test/Ocean/SplitExplicit/hydrostatic_spindown.jl:3-140, SplitExplicitSolver variant:
    SimpleBox 1e6 x 1e6 x 400 m, 3-D HBModel + 2-D ShallowWaterModel on the one-layer extrusion
    of the horizontal grid, periodic in x and y) at Nx x Nx x Nz elements per rank, N = 4; more
    ranks extend the box in y (same element size), both grids partitioned alike (whole columns)."""
    O, M = cm.ocean, cm.mesh
    Lx, Ly, H = 1e6, 1e6 * size, 400.0
    problem = O.SimpleBox(Lx, Ly, H, rotation=O.FIXED)
    law3 = O.HydrostaticBoussinesqModel(problem, c_h=1.0, alpha_T=0.0, kappa_h=0.0, kappa_z=0.0,
                                        coupled=True)
    law2 = O.ShallowWaterModel(problem, law3.nu_h, advection=False, coupled=True, c=1.0)
    x, y = np.linspace(0.0, Lx, Nx + 1), np.linspace(0.0, Ly, Nx * size + 1)
    topl = M.StackedBrickTopology([x, y, np.linspace(-H, 0.0, Nz + 1)],
                                  periodicity=(True, True, False),
                                  boundary=((0, 0), (0, 0), (1, 2)), rank=rank, size=size,
                                  **({"connectivity": connectivity} if connectivity else {}))
    grid3 = M.DiscontinuousSpectralElementGrid(topl, 4)
    # (fields are constant along the extrusion: two nodes carry them, 50 nodes per element)
    grid2 = O.extruded_barotropic_grid(x, y, 4, N_extrusion=1, rank=rank, size=size,
                                       **({"connectivity": connectivity} if connectivity else {}))
    # the reference's 5 x 5 x 8 runs use dt_slow = 5400 s over dt_fast = 300 s; both scale with
    # the horizontal element size so that the finer box stays inside the barotropic CFL limit
    return law3, grid3, law2, grid2, 5400.0 * 5.0 / Nx, 300.0 * 5.0 / Nx


def main_ocean(args, rank, world, local):
    """``--workload ocean-split-explicit``: one "step" is one slow LSRK54 step of the 3-D model
    (5 stages, each with its barotropic sub-steps, the two slow right-hand sides and the
    exchanges between the models).  ``value`` counts the 3-D model's DOF updates."""
    import torch
    import torch.distributed as dist
    from cmdg_loader import cm
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "WARN"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    O = cm.ocean
    Nx, Nz = args.ocean_nx, args.ocean_nz
    t0 = time.time()
    law3, g3, law2, g2, dt_slow, dt_fast = ocean_setup(cm, Nx, Nz, rank, world)
    log("[ocean rank %d] grids: %d + %d ghost 3-D elements, %.1f s" % (
        rank, g3.nreal, g3.nelem - g3.nreal, time.time() - t0))
    dg3 = cm.dgmodel.DGModel(law3, g3, device=dev)
    keep = O.install_hydrostatic_boussinesq_hooks(dg3)
    dg2 = cm.dgmodel.DGModel(law2, g2, device=dev,
                             numerical_flux_first_order=cm.balancelaws.CentralNumericalFluxFirstOrder)
    if distributed:        # one RCCL communicator per model: their exchanges interleave freely
        for d in (dg3, dg2):
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                uid = torch.frombuffer(bytearray(cm.dgmodel.rccl_unique_id()), dtype=torch.uint8).clone()
            uid = uid.to(dev)
            dist.broadcast(uid, 0)
            d.comm_init_rccl(uid.cpu().numpy().tobytes(), rank, world)
            d.comm_selftest()

    def sync_all():
        dg3.synchronize(), dg2.synchronize()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
    Q3, Q2 = dg3.init_ode_state(0.0), dg2.init_ode_state(0.0)
    se = O.SplitExplicitSolver(dg3, dg2, Q3, Q2, dt_slow, dt_fast)
    se.dostep(Q3, Q2, args.warmup)
    sync_all()
    t0 = time.perf_counter()
    se.dostep(Q3, Q2, args.steps)
    dg3.synchronize(), dg2.synchronize()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    total3 = g3.nreal
    if distributed:
        dist.barrier()
        tt = torch.tensor([el, float(g3.nreal)], device=dev, dtype=torch.float64)
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        el, total3 = float(mx[0].item()), int(round(tt[1].item()))
    for d in (dg3, dg2):
        d.profile_reset()
        if not args.no_events:
            d.profile_enable(True)
    t1 = time.perf_counter()
    se.dostep(Q3, Q2, args.steps)
    dg3.synchronize(), dg2.synchronize()
    torch.cuda.synchronize()
    el_ev = time.perf_counter() - t1
    for d in (dg3, dg2):
        d.profile_enable(False)
    RKC = se.RKC
    nsub = sum(int(np.ceil(((1 - RKC[s]) if s == 4 else (RKC[s + 1] - RKC[s])) * dt_slow / dt_fast))
               for s in range(5))
    finite = bool(torch.isfinite(Q3[:g3.nreal]).all().item() and torch.isfinite(Q2[:g2.nreal]).all().item())
    dofs = total3 * g3.Np * law3.ns * 5 * args.steps
    if distributed and rank != 0:
        del keep
        dg3.close(), dg2.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    kernels = {}
    for name, d in (("slow", dg3), ("fast", dg2)):
        for kn in ("GRADIENTS", "TENDENCY", "FILTER", "STACK_INTEGRAL"):
            ms, n = d.profile_get(kn)
            if n:
                kernels["%s_%s" % (name, kn.lower())] = {
                    "avg_ms": ms / n, "launches_per_step": n / args.steps, "ms_per_step": ms / args.steps}
    out = {"metric": "DG RHS DOF-updates/sec", "value": dofs / el, "unit": "DOF-updates/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "hydrostatic Boussinesq ocean box (BASELINE configs[4]), split-explicit "
                                  "barotropic / baroclinic stepper, %dx%dx%d elements, N=4 (%d 3-D "
                                  "elements, %d columns), dt_slow=%g s, dt_fast<=%g s (%d barotropic "
                                  "LSRK54 steps per slow step), Coupled, fp64"
                                  % (Nx, Nx * world, Nz, total3, total3 // Nz, dt_slow, dt_fast, nsub),
                      "elements": int(total3), "nodes_per_element": int(g3.Np), "states": law3.ns,
                      "parallelism": "element partition (Hilbert, whole columns, both models alike), "
                                     "%d rank(s), RCCL p2p halo" % world},
           "node_updates_per_s": dofs / el / law3.ns, "state_finite": finite, "kernels_ms": kernels}
    if "slow_tendency" in kernels:
        info = kernel_info(dg3, law3, g3, (0, 0))
        avg_ms = kernels["slow_tendency"]["avg_ms"]
        needed = needed_bytes_per_node(info, "TENDENCY") - 8 * 1.5 * law3.ns
        # (the slow model's update! is a launch of its own: the pass writes dQ, and reads it in the
        # increment = true call of the stage, not in the increment = false one)
        nodes = g3.nreal * g3.Np
        achieved = needed * nodes / (avg_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "k_tendency (3-D HBModel)", "achieved": achieved,
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": None, "algorithmic_bytes_per_launch": needed * nodes,
                           "algorithmic_bytes_per_node": needed, "avg_launch_ms": avg_ms,
                           "timing": "HIP events on the launch stream, second pass of the same %d "
                                     "steps (%.3f ms/step with events)" % (args.steps, 1e3 * el_ev / args.steps)}
    if not args.no_cpu and world == 1:
        from oracle import oracle as OR
        OR.build()
        host = host_cpu_info()
        OR.set_num_threads(baseline_threads(host, OR.get_max_threads()))
        F = cm.mesh.filters
        o3 = OR.OracleDGModel(law3, g3)
        OR.hydrostatic_boussinesq_hooks(o3, F.CutoffFilter(g3, 3), F.ExponentialFilter(g3, 1, 8))
        o2 = OR.OracleDGModel(law2, g2, nf_first=1)
        q3 = law3.init_state_prognostic(g3, o3.state_auxiliary, 0.0)
        q2 = law2.init_state_prognostic(g2, o2.state_auxiliary, 0.0)
        so = OR.SplitExplicitOracle(o3, o2, q3, q2, dt_slow, dt_fast)
        t0 = time.perf_counter()
        so.dostep(q3, q2, 0.0)
        c = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": g3.nreal * g3.Np * law3.ns * 5 / c, "unit": "DOF-updates/s",
                               "cores": OR.get_max_threads(), "kind": "port", "host": host,
                               "sample": "1 slow step of the same workload in %.1f s, OpenMP over elements" % c}
    emit(out)
    del keep
    dg3.close(), dg2.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if not finite:
        raise SystemExit("bench.py: the state is not finite after the timed steps")


if __name__ == "__main__":
    main()
