#!/usr/bin/env python3
"""How much of the ghost exchange is NOT hidden behind interior work, measured on one GPU.

One rank of the 8-rank Held-Suarez partition (its real elements, ghosts, interior / exterior
lists and per-neighbour ranges exactly as on an 8-GPU node) runs with the RCCL transport and every
neighbour mapped to the rank itself: message sizes, the number of send / receive pairs per group,
pack / unpack kernels, the second stream and the event choreography are the real ones; what is
not real is the wire (an RCCL self-copy instead of xGMI), so the transport time below is a lower
bound.  HIP events: CMDG_K_TRANSPORT brackets each RCCL group on the halo stream,
CMDG_K_HALO_EXPOSED is the time the compute stream had nothing left to do but wait for an
exchange (zero when the exchange finished behind the interior kernels).

    python scripts/measure_halo_exposure.py [--scaling weak|strong] [--rank R] [--steps K]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong", "weak-small"])
    ap.add_argument("--size", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0, help="first rank tried")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--repeat", type=int, default=3,
                    help="timed runs of --steps steps; the fastest is reported (all are listed)")
    ap.add_argument("--nvert", type=int, default=8)
    ap.add_argument("--nhorz", type=int, default=0)
    ap.add_argument("--reference-halo", action="store_true",
                    help="pack / unpack kernels around every exchange (CMDG_OPT_REFERENCE_HALO)")
    ap.add_argument("--step-graph", action="store_true",
                    help="record one step into a HIP graph and replay it (CMDG_OPT_STEP_GRAPH)")
    ap.add_argument("--async-run", action="store_true",
                    help="the handle's own thread enqueues the run (CMDG_OPT_ASYNC_RUN)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="interior and exterior launches on one stream (CMDG_OPT_HALO_PIPELINE = 0)")
    args = ap.parse_args()
    # face-connected ghosts: what a neighbour sends is what it receives (the vertex-connected
    # default of the stacked topologies lists a few hundred extra nodes on one side only)
    args.connectivity = "face"
    import torch
    from cmdg_loader import cm
    # a rank whose per-neighbour send and receive ranges have equal lengths (around a cube corner
    # of the sphere three elements meet and a pair of ranks may differ by a face)
    for rank in [args.rank] + [r for r in range(args.size) if r != args.rank]:
        law, grid, direction, dt, desc = bench.build_workload(cm, "heldsuarez", rank, args.size, 0, args)
        nn = len(grid.nabrtorank)
        send = np.asarray(grid.nabrtovmapsend).reshape(nn, 2)
        recv = np.asarray(grid.nabrtovmaprecv).reshape(nn, 2)
        if all(send[n][1] - send[n][0] == recv[n][1] - recv[n][0] for n in range(nn)):
            args.rank = rank
            break
    else:
        raise SystemExit("send / receive ranges differ per neighbour on every rank")
    real_nbrs = list(grid.nabrtorank)
    grid.nabrtorank = [0] * nn
    dg = cm.dgmodel.DGModel(law, grid, direction=direction[0], diffusion_direction=direction[1])
    dg.comm_init_rccl(cm.dgmodel.rccl_unique_id(), 0, 1)
    if args.reference_halo:
        dg.set_option(cm._lib.OPT_REFERENCE_HALO, 1)
    if args.no_pipeline:
        dg.set_option(cm._lib.OPT_HALO_PIPELINE, 0)
    if args.step_graph:
        dg.set_option(cm._lib.OPT_STEP_GRAPH, 1)
    if args.async_run:
        dg.set_option(cm._lib.OPT_ASYNC_RUN, 1)
    modes = {k: dg.query(k) for k in ("DIRECT_SEND", "DIRECT_RECV", "HALO_PIPELINE")}
    Q = dg.init_ode_state(0.0)
    solver = cm.odesolvers.LSRK54CarpenterKennedy(dg, Q, dt=dt)
    solver.dostep(Q, nsteps=20)
    dg.synchronize()
    runs, enq = [], []
    h0 = (dg.query("HOST_POST_NS"), dg.query("HOST_POST_COUNT"))
    for _ in range(args.repeat):
        Q.copy_(dg.init_ode_state(0.0))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.dostep(Q, nsteps=args.steps)
        enq.append(time.perf_counter() - t0)      # the host is done enqueueing (the run is asynchronous)
        dg.synchronize()
        runs.append(time.perf_counter() - t0)
    el = min(runs)
    h1 = (dg.query("HOST_POST_NS"), dg.query("HOST_POST_COUNT"))
    graph_steps = dg.query("GRAPH_STEPS")
    Q.copy_(dg.init_ode_state(0.0))
    torch.cuda.synchronize()
    dg.profile_reset()
    dg.profile_enable(True)
    solver.dostep(Q, nsteps=args.steps)
    dg.synchronize()
    dg.profile_enable(False)
    out = {"workload": desc["workload"], "rank": args.rank, "of": args.size,
           "real_elements": int(grid.nreal), "ghost_elements": int(grid.nelem - grid.nreal),
           "interior_elements": int(len(grid.interiorelems)), "exterior_elements": int(len(grid.exteriorelems)),
           "neighbours": real_nbrs, "send_nodes_per_neighbour": [int(s[1] - s[0] + 1) for s in send],
           "bytes_per_exchange_per_state_column": int(8 * len(grid.vmapsend)),
           "exchange": modes,
           "ms_per_step": 1e3 * el / args.steps, "steps": args.steps,
           "ms_per_step_runs": [1e3 * r / args.steps for r in runs],
           "host_enqueue_ms_per_step": 1e3 * min(enq) / args.steps,
           "step_graph": bool(args.step_graph), "graph_steps_replayed": int(graph_steps),
           "async_run": bool(args.async_run),
           "host_enqueue_is": "the caller's thread inside cmdg_lsrk_run" + (
               " (the run itself is enqueued by the handle's own thread)" if args.async_run else ""),
           "host_rccl_post_us_per_exchange": 1e-3 * (h1[0] - h0[0]) / max(h1[1] - h0[1], 1), "kernels": {}}
    for k in ("GRADIENTS", "GRADIENTS_EXT", "DIVGRAD", "DIVGRAD_EXT", "GRADLAP", "GRADLAP_EXT", "TENDENCY",
              "TENDENCY_EXT", "PACK", "TRANSPORT", "UNPACK", "HALO_EXPOSED"):
        ms, n = dg.profile_get(k)
        if n:
            out["kernels"][k] = {"avg_us": 1e3 * ms / n, "launches": n, "total_ms_per_step": ms / args.steps}
    ex = out["kernels"].get("HALO_EXPOSED")
    if ex:
        out["exposed_ms_per_step"] = ex["total_ms_per_step"]
        out["exposed_fraction_of_step"] = ex["total_ms_per_step"] / out["ms_per_step"]
        out["exchanges_per_stage"] = ex["launches"] / (5 * args.steps)
    if args.step_graph:   # the same 100 steps from the same state and time, replayed and eager: same bits?
        res = []
        for graph in (1, 0):
            dg.set_option(cm._lib.OPT_STEP_GRAPH, graph)
            Q.copy_(dg.init_ode_state(0.0))
            torch.cuda.synchronize()
            solver.t = 0.0
            g0 = dg.query("GRAPH_STEPS")
            # (few steps: with every neighbour mapped to the rank itself the flow is not physical
            # and does not stay finite for long)
            solver.dostep(Q, nsteps=2)          # one eager step, one replayed
            dg.synchronize()
            res.append((Q[:grid.nreal].cpu().numpy().copy(), dg.query("GRAPH_STEPS") - g0))
        # (bit equality, NaN positions included: the rehearsal's flow need not stay finite)
        out["graph_equals_eager"] = bool(np.array_equal(res[0][0], res[1][0], equal_nan=True))
        out["graph_check_nonfinite_values"] = int((~np.isfinite(res[0][0])).sum())
        out["graph_check_steps_replayed"] = [int(res[0][1]), int(res[1][1])]
    print(json.dumps(out))
    dg.close()


if __name__ == "__main__":
    main()
