#!/usr/bin/env python3
"""DG-RHS DOF-update throughput of the MI355X-native hot path.

``python bench.py --gpus N --steps K --warmup W``; for N > 1 launch with
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N``.

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

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_workload(cm, name, rank, size, ne, args):
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
        # src/Driver/driver_configs.jl:344-470).  Weak scaling keeps ~5 400 elements per GPU:
        # n_horz = 11 / 15 / 21 / 30 at 1 / 2 / 4 / 8 GPUs (6 x 30 x 30 x 8 at 8 GPUs).
        A = cm.atmos
        ps = A.PlanetParameters()
        n_horz = args.nhorz or {1: 11, 2: 15, 3: 18, 4: 21, 5: 24, 6: 26, 7: 28, 8: 30}.get(
            size, int(round(30 * (size / 8) ** 0.5)))
        n_vert = args.nvert
        Rrange = np.linspace(ps.planet_radius, ps.planet_radius + 30e3, n_vert + 1)
        topl = M.StackedCubedSphereTopology(n_horz, Rrange, boundary=(1, 2), rank=rank, size=size)
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
        grid = M.DiscontinuousSpectralElementGrid(topl, 6)
        law = MO.bomex_model(3000.0)
        desc = {"workload": "BOMEX moist LES (BASELINE configs[3]), stacked brick %dx%dx%d elements, "
                            "N=6, EquilMoist (saturation adjustment in every flux evaluation, as "
                            "in the reference snapshot), SmagorinskyLilly, BOMEX sources and "
                            "surface fluxes, LSRK54 explicit, Rusanov, fp64" % (nx, ny, nz),
                "elements": nx * ny * nz, "nodes_per_element": 343, "states": law.ns,
                "parallelism": "element partition (Hilbert, whole columns), %d rank(s)" % size}
        return law, grid, (0, 0), 0.004, desc
    raise SystemExit("unknown workload %s" % name)


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


def cpu_baseline(law, grid, direction, dt, budget_s):
    """The oracle (CPU restatement of the reference kernels in the reference's unfused
    launch order) timed on this host's cores on the same workload."""
    from oracle import oracle as O
    O.build()
    cores = O.get_max_threads()
    dg = O.OracleDGModel(law, grid, nf_first=0, direction=direction[0],
                         diffusion_direction=direction[1])
    Q = law.init_state_prognostic(grid, dg.state_auxiliary, 0.0)
    dQ = np.zeros_like(Q)
    O.lsrk54_step(dg, Q, dQ, 0.0, dt)           # warm-up (page faults, thread pool)
    n, t0 = 0, time.time()
    while True:
        O.lsrk54_step(dg, Q, dQ, n * dt, dt)
        n += 1
        el = time.time() - t0
        if el > budget_s or n >= 50:
            break
    dofs = grid.nreal * grid.Np * law.ns * 5 * n
    return {"value": dofs / el, "unit": "DOF-updates/s", "cores": cores, "kind": "port",
            "sample": "%d LSRK54 step(s) of the same workload (%d elements) in %.1f s, "
                      "OpenMP over elements" % (n, grid.nreal, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="heldsuarez", choices=["heldsuarez", "advdiff-brick", "risingbubble", "bomex"])
    ap.add_argument("--ne", type=int, default=32, help="advdiff-brick: elements per side per rank")
    ap.add_argument("--bomex-ne", type=int, default=16,
                    help="bomex: ne x ne x 2 ne elements per rank (32: the 65 536 elements of configs[3])")
    ap.add_argument("--nhorz", type=int, default=0, help="heldsuarez: elements per cube edge")
    ap.add_argument("--nvert", type=int, default=8, help="heldsuarez: vertical elements")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU baseline")
    ap.add_argument("--filter", action="store_true",
                    help="heldsuarez: exponential filter of the perturbations after every step "
                         "(experiments/AtmosGCM/heldsuarez.jl:261-272); off for the headline "
                         "metric, which is the RHS + LSRK path alone")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="time without per-kernel HIP events")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from cmdg_loader import cm

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with that many ranks" % args.gpus)
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    # launched by torch.distributed.run (even with one rank): take the distributed path, so
    # that a 1-rank launch rehearses process-group + RCCL set-up on a single-GPU box
    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner on stdout at NCCL_DEBUG=VERSION; stdout carries the one
        # JSON line of the contract
        if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "WARN"
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device(dev))

    t0 = time.time()
    law, grid, direction, dt, desc = build_workload(cm, args.workload, rank, world, args.ne, args)
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
    Q = dg.init_ode_state(0.0)
    solver = cm.odesolvers.LSRK54CarpenterKennedy(dg, Q, dt=dt)
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

    solver.dostep(Q, nsteps=args.warmup)
    sync_all()
    # ---- timed region: K steps, nothing but the library's own launches in flight ----------
    t0 = time.perf_counter()
    solver.t = args.warmup * dt
    solver.dostep(Q, nsteps=args.steps)
    dg.synchronize()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if distributed:
        dist.barrier()
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
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

    if rank == 0:
        dofs = total_elems * grid.Np * law.ns * 5 * args.steps
        kern = {}
        for k in ("GRADIENTS", "DIVGRAD", "GRADLAP", "TENDENCY", "PACK", "UNPACK", "FILTER"):
            ms, n = dg.profile_get(k)
            if n:
                kern[k] = (ms / n, n)
        if not kern:
            print(json.dumps({"ms_per_step": 1e3 * el / args.steps, "value": dofs / el}), flush=True)
            return
        dom = max((k for k in kern if k in ("GRADIENTS", "DIVGRAD", "GRADLAP", "TENDENCY")),
                  key=lambda k: kern[k][0] * kern[k][1])
        avg_ms, nl = kern[dom]
        elems_per_launch = grid.nreal * 5 * args.steps / nl   # interior/exterior launches split
        bytes_per_launch = algorithmic_bytes_per_node(law, dom, grid.N[0] + 1) * grid.Np * elems_per_launch
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        # HBM-side traffic of the dominant kernel from the committed rocprofv3 --pmc passes of
        # this same command (scripts/pmc_any.sh; FETCH_SIZE and WRITE_SIZE in separate passes, KB
        # units, FETCH_SIZE doubled as the gfx950 guide prescribes and as
        # profiles/r01_pmc_calibration_n30.json confirms for this library's 8-byte-per-lane
        # loads); only quoted for the configuration it was measured on
        traffic, traffic_src = None, None
        pmc_files = {("heldsuarez", 5808): "r01_heldsuarez_n11_pmc_hbm_per_launch.json",
                     ("risingbubble", 8000): "r01_risingbubble_8000_pmc_hbm_per_launch.json",
                     ("bomex", 8192): "r01_bomex_n6_8192_pmc_hbm_per_launch.json"}
        pmc_name = pmc_files.get((args.workload, grid.nreal))
        if pmc_name and world == 1 and not args.filter:
            pmc_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", pmc_name)
            if os.path.exists(pmc_file):
                pm = json.load(open(pmc_file)).get("k_%s" % dom.lower())
                if pm and "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
                    traffic = 1024.0 * (2.0 * pm["FETCH_SIZE"] + pm["WRITE_SIZE"])
                    traffic_src = "profiles/" + pmc_name
        out = {
            "metric": "DG RHS DOF-updates/sec", "value": dofs / el, "unit": "DOF-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": desc,
            "node_updates_per_s": dofs / el / law.ns,
            "state_finite": finite,
            "kernels_ms": {k: {"avg_ms": v[0], "launches": v[1]} for k, v in kern.items()},
            "roofline": {"bound": "hbm", "kernel": "k_%s" % dom.lower(), "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "algorithmic_bytes_per_node": algorithmic_bytes_per_node(law, dom, grid.N[0] + 1),
                         "avg_launch_ms": avg_ms,
                         "timing": "HIP events on the launch stream, second pass of the same "
                                   "%d steps (%.3f ms/step with events)" % (
                                       args.steps, 1e3 * el_events / args.steps)},
        }
        # flops of the (N+1)-point derivative contractions alone (2 (N+1) per
        # node, differentiated scalar and direction): gradient arguments, the two hyperdiffusion
        # passes, the tendency.  Secondary figure: the path is HBM bound (< 1 flop/B), the
        # contraction runs on fp64 VALU out of LDS, not on MFMA.
        cf = 6 * (grid.N[0] + 1) * (law.ngrad + 2 * law.ngradlap + law.ns)
        out["contraction"] = {"flops_per_node_update": cf,
                              "achieved_TFLOPs": cf * (dofs / law.ns) / el / 1e12,
                              "unit": "fp64 VALU (no MFMA)"}
        if "FILTER" in kern:
            # Q read + written (5 fields each) and the two reference-state columns
            fb = 8 * (2 * law.ns + 2) * grid.Np * grid.nreal
            out["filter_kernel"] = {"avg_launch_ms": kern["FILTER"][0],
                                    "algorithmic_bytes_per_node": 8 * (2 * law.ns + 2),
                                    "achieved_GBs": fb / (kern["FILTER"][0] * 1e-3) / 1e9}
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(law, grid, direction, dt, args.cpu_budget)
        print(json.dumps(out), flush=True)
    if step_filter is not None:
        dg.set_filters()
        step_filter.close()
    dg.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
