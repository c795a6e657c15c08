/* libcmdg -- MI355X-native DG right-hand side + low-storage Runge-Kutta stepping for
 * ClimateMachine-style balance laws.  C ABI: plain pointers and sizes, no C++ or
 * torch types.  One handle per (rank, GPU); a handle is driven by one host thread.
 *
 * The reference (ClimateMachine.jl, Julia) has no FFI seam on this path: the seam
 * is the Julia callable `(dg::DGModel)(tendency, Q, _, t, alpha, beta)`.  Each
 * entry point below names the reference interface it replaces (paths relative to
 * the reference root).  INTEGRATION.md shows the Julia `ccall` stubs.
 *
 * All arrays use the reference's column-major layouts and 1-based integer tables:
 *   state arrays (Np, nstate, nelem) Float64, nelem = nreal + nghost, real first
 *   vgeo (Np, nvgeo, nelem)   sgeo (5, Nfp, nface, nelem)     [Grids.jl:76-146]
 *   vmapM/vmapP (Nfp, nface, nelem) Int64                     [Grids.jl:559-637]
 *   elemtobndy (nface, nelem) Int64                           [Topologies.jl]
 * "device" pointers must stay valid from cmdg_create to cmdg_destroy.
 * Every function returns 0 (CMDG_OK) or a negative cmdg_status; none throws.
 *
 * Environment (read when a handle is created; none is needed in normal use):
 *   CMDG_RCCL_LIB          path of the RCCL library to use (default: the one already in the process,
 *                          e.g. torch's, else librccl.so.1)
 *   CMDG_ROCTX=1           load the roctx library so that the phases of an evaluation carry ranges
 *                          (cmdg:halo:pack / transport / end, cmdg:<pass>[:exterior]); they are
 *                          emitted anyway when rocprofv3 --marker-trace has loaded it
 *   CMDG_REFERENCE_HALO=1, CMDG_HALO_PIPELINE=0   initial values of the two exchange options below
 *   CMDG_HALO_PRIORITY=1   create the halo stream with the highest stream priority
 */
#ifndef CMDG_H
#define CMDG_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cmdg_context *cmdg_handle;

typedef enum cmdg_status {
    CMDG_OK = 0,
    CMDG_ERR_INVALID = -1,     /* bad argument / unsupported configuration */
    CMDG_ERR_HIP = -2,         /* a HIP runtime call failed */
    CMDG_ERR_NO_DEVICE = -3,   /* no gfx950 device visible */
    CMDG_ERR_COMM = -4,        /* RCCL / transport failure */
    CMDG_ERR_UNSUPPORTED = -5  /* physics / polynomial order not compiled in */
} cmdg_status;

/* directions (src/Numerics/Mesh/Grids.jl Direction types) */
enum { CMDG_EVERY_DIRECTION = 0, CMDG_HORIZONTAL_DIRECTION = 1, CMDG_VERTICAL_DIRECTION = 2 };
/* first-order numerical flux (NumericalFluxes.jl:219,298); second order and gradient
 * fluxes are the central ones (:668, :65), as in every configuration in scope */
enum {
    CMDG_RUSANOV = 0, CMDG_CENTRAL_FIRST_ORDER = 1,
    /* methods the dry AtmosModel defines for itself (src/Atmos/Model/AtmosModel.jl:1006,
     * :1154, :1515); CMDG_PHYSICS_DRY_ATMOS without orientation / reference state only */
    CMDG_ROE = 2, CMDG_HLLC = 3, CMDG_LMARS = 4,
    /* RoeNumericalFluxMoist(; LM, HH, LV, LVPP) of the moist AtmosModel (AtmosModel.jl:1276-1513):
     * CMDG_PHYSICS_MOIST_ATMOS with constant viscosity, N = 4 */
    CMDG_ROE_MOIST = 5, CMDG_ROE_MOIST_LM = 6, CMDG_ROE_MOIST_HH = 7, CMDG_ROE_MOIST_LV = 8,
    CMDG_ROE_MOIST_LVPP = 9
};
/* balance laws carried as device functors (pointwise Julia physics cannot cross a C ABI) */
enum {
    CMDG_PHYSICS_ADVECTION_DIFFUSION = 1, CMDG_PHYSICS_DRY_ATMOS = 2,
    CMDG_PHYSICS_HYDROSTATIC_BOUSSINESQ = 3,
    CMDG_PHYSICS_PRESSURE_GRADIENT = 4, /* PressureGradientModel, ref_state.jl:196-233 */
    CMDG_PHYSICS_SHALLOW_WATER = 5,     /* ShallowWaterModel (barotropic half of the split-explicit
                                           ocean), on a one-layer extrusion of the 2-D grid */
    CMDG_PHYSICS_MOIST_ATMOS = 6,       /* AtmosModel LES configuration with EquilMoist
                                           (src/Atmos/Model/moisture.jl:70-115) */
    /* the older split-explicit ocean, src/Ocean/SplitExplicit01: OceanModel, the
     * Continuity3dModel its update_auxiliary_state! evaluates, and the BarotropicModel (on the
     * one-layer extrusion of the 2-D grid) */
    CMDG_PHYSICS_OCEAN_SE01 = 7, CMDG_PHYSICS_CONTINUITY3D_SE01 = 8,
    CMDG_PHYSICS_BAROTROPIC_SE01 = 9
};

/* Construction record: the fields of `DGModel(balance_law, grid, nf1, nf2, nfgrad;
 * state_auxiliary, state_gradient_flux, states_higher_order, direction,
 * diffusion_direction)` (src/Numerics/DGMethods/DGModel.jl:22-65) and of the
 * `DiscontinuousSpectralElementGrid` it reads (src/Numerics/Mesh/Grids.jl:187-264). */
typedef struct cmdg_desc {
    int32_t dim;                 /* 3 */
    int32_t N[3];                /* polynomial orders (horizontal, horizontal, vertical);
                                    N[0] == N[1]; N[2] may differ (polynomialorder = (N_h, N_v),
                                    Grids.jl:187-264) for the combinations compiled in */
    int64_t nreal, nghost;       /* topology.realelems / ghostelems counts */
    int32_t nvgeo;               /* columns of vgeo (25: GeometricFactors.jl:60-67) */
    int32_t physics_id;          /* CMDG_PHYSICS_* */
    int32_t iparam[16];          /* law parameters, see csrc/physics_*.h */
    double dparam[64];
    int32_t nf_first;            /* CMDG_RUSANOV | CMDG_CENTRAL_FIRST_ORDER | CMDG_ROE | CMDG_HLLC | CMDG_LMARS */
    int32_t direction;           /* dg.direction */
    int32_t diffusion_direction; /* dg.diffusion_direction */
    int32_t stacked;             /* isstacked(grid.topology): with a vertical-only
                                    direction no halo is exchanged (DGModel.jl:104-108) */
    /* grid tables, device pointers */
    const double *vgeo, *sgeo;
    const int64_t *vmapM, *vmapP, *elemtobndy;
    const int64_t *interiorelems;
    int64_t ninterior;           /* grid.interiorelems (1-based) */
    const int64_t *exteriorelems;
    int64_t nexterior;           /* grid.exteriorelems (1-based) */
    const uint8_t *activedofs;   /* (Np*nelem) grid.activedofs, may be NULL if unused */
    const double *D;             /* HOST pointer, (Nq, Nq) column-major, grid.D[1] */
    /* halo tables: grid.vmapsend / vmaprecv (device) and per-neighbour ranges (host) */
    const int64_t *vmapsend;
    int64_t nvmapsend;
    const int64_t *vmaprecv;
    int64_t nvmaprecv;
    int32_t nnabr;
    const int32_t *nabrtorank;       /* HOST, nnabr */
    const int64_t *nabrtovmapsend;   /* HOST, 2*nnabr: first,last (1-based inclusive) */
    const int64_t *nabrtovmaprecv;   /* HOST, 2*nnabr */
    /* arrays owned by the `dg` in the reference; device pointers, caller-owned.
     * state_auxiliary is required; the other three may be NULL (library allocates). */
    double *state_auxiliary;       /* (Np, naux, nelem) */
    double *state_gradient_flux;   /* (Np, ngradflux, nelem); physics_id 2: filled by
                                      cmdg_export_gradient_flux only (see there) */
    double *Qhypervisc_grad;       /* (Np, 3*ngradlap, nelem)  create_states.jl:22-26; filled by
                                      cmdg_export_hypervisc_grad only (see there) */
    double *Qhypervisc_div;        /* (Np, nhyper, nelem) */
    const double *Dv;              /* HOST pointer, (Nqv, Nqv) column-major, grid.D[end]; may be
                                      NULL when N[2] == N[0] */
} cmdg_desc;

/* ---- queries ------------------------------------------------------------------ */
const char *cmdg_version(void);
const char *cmdg_status_string(int status);
/* number_states(balance_law, st) for st = Prognostic, Auxiliary, Gradient, GradientFlux,
 * GradientLaplacian, Hyperdiffusive (src/BalanceLaws/state_types.jl:40-108) -> out[6] */
int cmdg_physics_counts(int32_t physics_id, const int32_t *iparam, int32_t out[6]);

/* ---- lifetime ----------------------------------------------------------------- */
/* replaces DGModel(...) construction, DGModel.jl:22-65.
 * Preconditions on the grid tables, checked on the device at create (CMDG_ERR_INVALID with the
 * message in cmdg_last_error(NULL) otherwise) -- every grid DiscontinuousSpectralElementGrid
 * builds satisfies them:
 *   - vmapM is the canonical face numbering of Grids.jl:586-594 (vmapM[n, f, e] = the volume node
 *     of face node n of face f of element e): the kernels compute it instead of loading it;
 *   - sgeo's vMI equals vgeo's MI at the face node, bit for bit (Grids.jl:1097-1101);
 *   - plus-side node ids fit 32 bits (nelem * Np < 2^31).
 * The handle keeps a digest of the face tables of its own next to the caller's arrays: 36 B per
 * face node of the real elements (int32 plus-side node + normal + sM); handles with neighbours
 * add 8 B per node of vmapsend and 4 B per ghost node. */
int cmdg_create(const cmdg_desc *desc, cmdg_handle *out);
int cmdg_destroy(cmdg_handle h);

/* Balance laws / template combinations outside the compiled set.  In the reference a law's
 * pointwise functions are compiled into the kernels when the model first runs
 * (src/BalanceLaws/interface.jl:37-464, KernelAbstractions); behind a C ABI the equivalent is a
 * plug-in: a shared object built from this library's own kernel headers (csrc/engine.h + a
 * physics_*.h device functor, one translation unit instantiating make_engine<Law, Nq>) that exports
 *   cmdg::EngineBase *cmdg_plugin_make_engine(const cmdg_desc *, char *err, int errlen)
 * and returns NULL for descriptors it does not serve, and
 *   unsigned long cmdg_plugin_abi(void)        -- engine_abi_stamp() of the headers it was built with;
 * a plug-in built against other headers than this library is refused at load.  cmdg_create asks the loaded plug-ins when no
 * compiled-in engine takes a descriptor (a law the library knows in a combination or at an order
 * it was not built for, or a physics_id of the plug-in's own).  cmdg_load_plugin loads one
 * (idempotent; CMDG_ERR_INVALID with the reason in cmdg_last_error(NULL)); the environment variable
 * CMDG_PLUGINS (colon-separated paths) is read at the first create that needs it.
 * climatemachine.jl_amd/plugins.py writes and builds plug-ins with hipcc. */
int cmdg_load_plugin(const char *path);
/* message of the last failure on this handle (never NULL) */
const char *cmdg_last_error(cmdg_handle h);

/* ---- the hot path ------------------------------------------------------------- */
/* replaces (dg::DGModel)(tendency, Q, _, t, alpha, beta), DGModel.jl:85-427:
 *   tendency .= alpha * dQdt(Q, t) + beta * tendency   on real elements;
 * ghost face nodes of Q are refreshed as a side effect; returns when all device work is
 * complete (DGModel.jl:426). */
int cmdg_rhs(cmdg_handle h, double *tendency, double *Q, double t, double alpha, double beta);
/* same, enqueued on the handle's stream without the final wait */
int cmdg_rhs_async(cmdg_handle h, double *tendency, double *Q, double t, double alpha,
                   double beta);
/* replaces dostep!(Q, lsrk::LowStorageRungeKutta2N, p, time) with its update! kernel,
 * LowStorageRungeKuttaMethod.jl:102-158: for s in 1:nstages
 *   dQ = rhs(Q, t + rkc[s] dt) + dQ;  Q += rkb[s] dt dQ;  dQ *= rka[s % nstages + 1]
 * (the update is fused into the last kernel of each stage).  Asynchronous: call
 * cmdg_synchronize before reading Q on another stream or on the host. */
int cmdg_lsrk_step(cmdg_handle h, double *Q, double *dQ, double t, double dt, int32_t nstages,
                   const double *rka, const double *rkb, const double *rkc);
/* nsteps back-to-back steps of size dt starting at t (solve! loop body,
 * ODESolvers.jl:110-158 without callbacks); step i starts at the running sum t + dt + ... + dt,
 * as updatetime! accumulates it (ODESolvers.jl:96-98) */
int cmdg_lsrk_run(cmdg_handle h, double *Q, double *dQ, double t, double dt, int64_t nsteps,
                  int32_t nstages, const double *rka, const double *rkb, const double *rkc);
int cmdg_synchronize(cmdg_handle h);

/* Options of a handle.
 * CMDG_OPT_KEEP_GRADFLUX (default 0): a law whose second-order flux does not read the
 *   gradient-flux state (the dry atmosphere with zero viscosity: tau = -2 nu S = 0) has the
 *   nine columns of dg.state_gradient_flux neither formed, stored nor exchanged -- the tendency
 *   is bit-identical without them.  1 restores the reference's behaviour of refreshing
 *   state_gradient_flux in every evaluation (DGModel.jl:126-206), for callers that read it
 *   (diagnostics, the diffusive Courant number).
 * CMDG_OPT_STACK_HEIGHT (default 0 = unknown): number of elements of a vertical stack of a stacked
 *   topology (length(topology.stacksize); elements of a stack are contiguous, e = ev + (eh - 1) nv).
 *   Results do not depend on it.  With tall stacks (more than 16 elements) the engine walks its
 *   element lists in tiles of 32 columns x 4 levels instead of column by column, so that the
 *   elements in flight on one XCD are horizontal neighbours whose face gathers meet in its L2.
 * CMDG_OPT_REFERENCE_HALO (default 0): handles with neighbours.  By default an evaluation launches
 *   no pack or unpack kernel: the exterior launch of every pass writes the nodes of vmapsend of
 *   what it produces straight into the send buffer, and -- unless the law's nodal
 *   update_auxiliary_state! or the hooks read the ghost ELEMENTS -- the face kernels read the plus
 *   side of ghost neighbours from the receive buffers, so the ghost elements of Q, of
 *   state_gradient_flux and of the hyperdiffusion arrays are NOT refreshed.  Results are
 *   bit-identical.  1 restores begin/end_ghost_exchange! as the reference runs them
 *   (kernel_fillsendbuf! / kernel_transferrecvbuf! around every exchange, MPIStateArrays.jl:411-483),
 *   for callers that read ghost elements after an evaluation (diagnostics, output, a Courant
 *   number over ghosts).  cmdg_halo_begin/end always do.  cmdg_query(CMDG_Q_DIRECT_RECV) says
 *   which mode an evaluation runs in: it turns itself off when hooks or a nodal
 *   update_auxiliary_state! kernel read ghost elements.  Wire format: the reference's (nstate,
 *   nvmap) state-fastest buffers, with one exception in the default mode -- of Qhypervisc_div's
 *   nhyp columns only the ngl the Laplacian pass writes (and the next pass reads) travel.
 * CMDG_OPT_HALO_PIPELINE (default 1): handles whose exchanges run direct both ways launch the
 *   exterior element list of every pass on the halo stream, between the exchange it waits for
 *   and the exchange it feeds (a chain without event hops), and the interior list on the compute
 *   stream one pass behind or ahead (the passes of a step form two loosely coupled pipelines).
 *   0: the reference's order on one compute stream (interior launch, wait, exterior launch).
 *   Results do not depend on it.
 * CMDG_OPT_STEP_GRAPH (default 0; environment CMDG_STEP_GRAPH=1): cmdg_lsrk_run records one step
 *   into a HIP graph and replays it for every step but the first of a run; the evaluation times
 *   live in device memory and advance as updatetime! does.  Results are bit-identical.
 *   Handles that exchange through RCCL are recorded too when their exchanges run direct and
 *   pipelined (CMDG_Q_HALO_PIPELINE): one hipGraphLaunch then stands for the 40 kernel launches,
 *   100 event operations and 20 RCCL groups of a Held-Suarez step.  What crashed in round 3 is
 *   settled (scripts/probe/rccl_capture_probe.py, one fresh process per hypothesis, both stacks):
 *   on HIP 7.0.2 / RCCL 2.26.6 hipStreamEndCapture segfaults if and only if an ncclSend / ncclRecv
 *   group was recorded on a stream that JOINED the capture through an event -- Global or Relaxed
 *   mode alike; on the origin stream every mode, several operations per group and several groups
 *   per capture are fine; ROCm 7.2 / RCCL 2.27.7 accepts both forms (the probes use a rank as its
 *   own peer: whether the crash needs the self-send cannot be told on one GPU, and does not matter
 *   for the remedy).  The
 *   library therefore begins such a capture on the halo stream (the groups' stream) and forks the
 *   compute stream.  Runs a capture cannot hold stay eager: profiling, filters, hooks, a nodal
 *   update_auxiliary_state! kernel of its own, the local transport, exchanges that are packed.
 * CMDG_OPT_TENDENCY_PAIRS (default 0; environment CMDG_TENDENCY_PAIRS=1): the tendency pass takes
 *   horizontally adjacent elements two to a work-group and reads the xi1 face they share out of
 *   LDS instead of gathering it (laws with one polynomial order and no node cache; the pairs are
 *   found from vmap+ at create: faces that meet node for node).  Results are bit-identical.
 * CMDG_OPT_TENDENCY_FOUR_WAVES (environment CMDG_TENDENCY_FOUR_WAVES): elements above N = 4 -- the
 *   tendency pass on 256-thread work-groups, two nodes per thread, three work-groups per CU
 *   (k_tendency_big) instead of two elements per eleven-wave work-group.  Bit-identical; a
 *   recorded experiment like CMDG_OPT_TENDENCY_PAIRS (slower; effective only in a library built
 *   with -DCMDG_TEND_FOUR_WAVES=1 / -DCMDG_TEND_PAIRS=1, a no-op otherwise).
 * CMDG_OPT_ASYNC_RUN (default 0): cmdg_lsrk_run hands the run to a thread the handle owns and
 *   returns at once (the tableau is copied, Q and dQ must stay valid): the calling thread is not the
 *   one that spends ~1 ms per step of a partitioned run inside hipGraphLaunch or posting RCCL
 *   groups.  Runs execute in order; every other entry point of the handle first waits until that
 *   thread is idle, so the handle is still driven by one thread at a time; a deferred run that
 *   failed is reported by the next cmdg_synchronize.  Same launches in the same order: results
 *   are bit-identical.
 * CMDG_OPT_STREAM_PRIORITY (default 0): 1 puts both streams of the handle at the device's highest
 *   stream priority, -1 at the lowest.  Meant for a handle whose launches are small and form a long dependent chain
 *   next to another handle's bandwidth-bound launches (the barotropic model of the split-explicit
 *   ocean, whose sub-steps decide the length of a slow stage).  The handle must be idle; results
 *   do not depend on it. */
enum {
    CMDG_OPT_KEEP_GRADFLUX = 1, CMDG_OPT_STACK_HEIGHT = 2, CMDG_OPT_REFERENCE_HALO = 3,
    CMDG_OPT_HALO_PIPELINE = 4, CMDG_OPT_STEP_GRAPH = 5, CMDG_OPT_STREAM_PRIORITY = 6,
    CMDG_OPT_TENDENCY_PAIRS = 7, CMDG_OPT_ASYNC_RUN = 8,
    CMDG_OPT_TENDENCY_FOUR_WAVES = 9
};
int cmdg_set_option(cmdg_handle h, int32_t option, int32_t value);

/* What the handle's kernels actually do, for byte accounting and tests (no reference counterpart):
 *   GRADFLUX_LIVE         is state_gradient_flux formed by an evaluation (see CMDG_OPT_KEEP_GRADFLUX)
 *   LAW_NEEDS_GRADFLUX    does the tendency pass read it
 *   NDERIVED              columns of handle-owned time-invariant per-node fields the source reads
 *   NUPDATED_AUX          auxiliary columns the law's nodal update_auxiliary_state! rewrites (0: none)
 *   FUSED_UPDATE_AUX      is that refresh fused into the gradient pass
 *   DIRECT_SEND / _RECV   is the ghost exchange running without pack / unpack launches
 *   TENDENCY_ELEMS_PER_GROUP  elements per work-group of the tendency pass
 *   STATE_READ + p, AUX_READ + p   columns of Q / of state_auxiliary the volume code of pass p
 *                         reads (p = 0 gradients, 1 Laplacian, 2 gradient of Laplacian, 3 tendency);
 *                         every column for a law that does not declare less */
enum {
    CMDG_Q_GRADFLUX_LIVE = 1, CMDG_Q_LAW_NEEDS_GRADFLUX = 2, CMDG_Q_NDERIVED = 3,
    CMDG_Q_NUPDATED_AUX = 4, CMDG_Q_FUSED_UPDATE_AUX = 5, CMDG_Q_DIRECT_SEND = 6,
    CMDG_Q_DIRECT_RECV = 7, CMDG_Q_TENDENCY_ELEMS_PER_GROUP = 8,
    CMDG_Q_HALO_PIPELINE = 9, /* are the two pipelines of CMDG_OPT_HALO_PIPELINE in use */
    CMDG_Q_HOST_POST_NS = 10,   /* host nanoseconds spent posting exchanges (RCCL group calls) ... */
    CMDG_Q_HOST_POST_COUNT = 11, /* ... and how many were posted, since the handle was created */
    CMDG_Q_GRAPH_STEPS = 12,     /* steps cmdg_lsrk_run replayed from a captured graph */
    CMDG_Q_TENDENCY_PAIRS = 13,  /* work-groups of the tendency pass that share a face (-1: option off) */
    CMDG_Q_STATE_READ = 16, CMDG_Q_AUX_READ = 20
};
int cmdg_query(cmdg_handle h, int32_t what, int64_t *out);

/* dg.states_higher_order[1] (Qhypervisc_grad, create_states.jl:22-26) in the reference layout
 * (Np, 3*ngradlap, nelem).  The library keeps its working copy node-major -- (3*ngradlap, Np,
 * nelem): the 3*ngradlap values of a node are contiguous, which is what a plus-side face gather of
 * the Laplacian and tendency passes reads -- so cmdg_desc.Qhypervisc_grad is
 * NOT written by an evaluation; this call writes the array as the reference would hold it after
 * the last evaluation (`dst`, or cmdg_desc.Qhypervisc_grad if dst is NULL) and returns when the
 * copy is complete.  The copy runs on the handle's own (non-blocking) stream: work the caller has
 * pending on `dst` on another stream -- the fill of a freshly allocated array -- must have finished
 * before the call.  Ghost elements as described for CMDG_OPT_REFERENCE_HALO.  For diagnostics
 * and tests: nothing on the hot path reads the reference layout. */
int cmdg_export_hypervisc_grad(cmdg_handle h, double *dst);
/* The same for dg.state_gradient_flux, (Np, ngradflux, nelem).  For the dry atmosphere (physics_id 2)
 * the library's working copy is node-major as well and cmdg_desc.state_gradient_flux is
 * written by this call only; for every other law the array of cmdg_desc IS the working copy (hooks
 * and filters address it) and the call is a no-op (dst NULL or that array) or a copy. */
int cmdg_export_gradient_flux(cmdg_handle h, double *dst);

/* ---- halo (MPIStateArrays.jl:411-514, 837-871) -------------------------------- */
/* begin_ghost_exchange!: pack face nodes of `array` (Np, nstate, nelem) and post the
 * sends/receives; end_ghost_exchange!: wait and unpack into the ghost elements. */
int cmdg_halo_begin(cmdg_handle h, double *array, int32_t nstate);
int cmdg_halo_end(cmdg_handle h, double *array, int32_t nstate);
/* kernel_fillsendbuf! / kernel_transferrecvbuf! (MPIStateArrays.jl:837-871) on their own, for
 * arrays of any (Np, nstate): sendbuf (nstate, nvmap) state-fastest <- buf (Np, nstate, nelem)
 * at the 1-based linear node ids vmapsend[i] = n + Np (e - 1), and the inverse into the ghost
 * elements.  Device pointers; runs on the null stream and returns when done.  (The handle-based
 * exchange above launches the same kernels; these entries let the reference's known-answer
 * test, test/Arrays/mpi_comm.jl:23-153, be replayed on them.) */
int cmdg_fillsendbuf(double *sendbuf, const double *buf, const int64_t *vmapsend, int64_t nvmap,
                     int32_t Np, int32_t nstate);
int cmdg_transferrecvbuf(double *buf, const double *recvbuf, const int64_t *vmaprecv,
                         int64_t nvmap, int32_t Np, int32_t nstate);
/* Transport set-up.  Multi-process: RCCL point-to-point; `unique_id` is the 128-byte
 * ncclUniqueId made by cmdg_comm_unique_id on rank 0 and broadcast by the caller. */
int cmdg_comm_unique_id(void *out128);
int cmdg_comm_init_rccl(cmdg_handle h, const void *unique_id128, int32_t rank, int32_t nranks);
/* Transport self-check: sends `count` doubles to this rank itself with the same
 * ncclGroupStart / ncclRecv / ncclSend / ncclGroupEnd sequence the halo uses, on the halo
 * stream, and verifies the payload.  Needs cmdg_comm_init_rccl first (nranks may be 1). */
int cmdg_comm_selftest(cmdg_handle h, int64_t count);
/* Single-process: connect n handles (handle r plays rank r) through device copies. */
int cmdg_comm_connect_local(cmdg_handle *handles, int32_t n);

/* Lock-step drivers for handles connected with cmdg_comm_connect_local (one host thread
 * plays every rank): arrays of n per-rank pointers. */
int cmdg_group_rhs(cmdg_handle *handles, int32_t n, double **tendency, double **Q, double t,
                   double alpha, double beta);
/* one ghost exchange of a state-like array per handle (begin on all, then end on all):
 * e.g. state_auxiliary columns after initialisation (SpaceDiscretization.jl create_state +
 * MPIStateArrays.begin/end_ghost_exchange!) */
int cmdg_group_halo(cmdg_handle *handles, int32_t n, double **arrays, int32_t nstate);
int cmdg_group_lsrk_run(cmdg_handle *handles, int32_t n, double **Q, double **dQ, double t,
                        double dt, int64_t nsteps, int32_t nstages, const double *rka,
                        const double *rkb, const double *rkc);

/* ---- reductions (MPIStateArrays.jl:583-644), local part; caller all-reduces ---- */
/* out = sum over real elements of M .* A.^2  (weighted) or A.^2 */
int cmdg_norm2_local(cmdg_handle h, const double *A, int32_t nstate, int32_t weighted,
                     double *out_host);
/* out = sum over real elements of M .* (A - B).^2 */
int cmdg_distance2_local(cmdg_handle h, const double *A, const double *B, int32_t nstate,
                         double *out_host);

/* ---- Courant numbers and time-step selection ------------------------------------------ */
/* local_courant functions of src/Atmos/Model/courant.jl:29-83 and, for the ocean model,
 * src/Ocean/HydrostaticBoussinesq/Courant.jl:13-111 (which adds viscous_courant) */
enum {
    CMDG_ADVECTIVE_COURANT = 0, CMDG_NONDIFFUSIVE_COURANT = 1, CMDG_DIFFUSIVE_COURANT = 2,
    CMDG_VISCOUS_COURANT = 3
};
/* courant(local_courant, dg, m, Q, dt, simtime, direction) (SpaceDiscretization.jl:307-365):
 * maximum over this rank's real nodes of the law's local Courant number, with dx the
 * minimum neighbour distance of the node in `direction`; -inf when the rank owns no element.
 * The caller applies MPI.Allreduce(max) (:364).  calculate_dt (DGMethods.jl:79-83) is
 * Courant_number / cmdg_courant(NONDIFFUSIVE, dt = 1).  Blocks until the value is on the host. */
int cmdg_courant(cmdg_handle h, int32_t kind, const double *Q, double dt, double simtime,
                 int32_t direction, double *out_host);
/* min_node_distance(grid, direction) (Grids.jl:455-486): rank-local minimum, +inf for an empty
 * rank; the caller applies MPI.Allreduce(min). */
int cmdg_min_node_distance(cmdg_handle h, int32_t direction, double *out_host);

/* ---- column (stack) integrals ------------------------------------------------------ */
/* indefinite_stack_integral! / reverse_indefinite_stack_integral! (DGModel.jl:445-529, kernels
 * DGModel_kernels.jl:1903-2104) on stacked topologies (elements of a stack contiguous,
 * e = ev + (eh - 1) nvertelem).  The law's integral_load/set_auxiliary_state! hooks are
 * carried as a field combination: integrand_s = scale_s * field_s, where field_s is column
 * src_col[s] (0-based) of the prognostic state (src_is_state[s] != 0) or of the auxiliary
 * array; the upward integral goes to auxiliary column dst_col[s]; the reverse integral reads
 * auxiliary column rsrc_col[s] and writes (value at the top of the stack) - value to rdst_col[s]. */
#define CMDG_STACK_MAXOUT 8
typedef struct cmdg_stack_integral_desc {
    int32_t nout; /* 1..CMDG_STACK_MAXOUT */
    int32_t src_is_state[CMDG_STACK_MAXOUT], src_col[CMDG_STACK_MAXOUT];
    double scale[CMDG_STACK_MAXOUT];
    int32_t dst_col[CMDG_STACK_MAXOUT];
    int32_t rsrc_col[CMDG_STACK_MAXOUT], rdst_col[CMDG_STACK_MAXOUT];
} cmdg_stack_integral_desc;
/* Q (Np, nstate, nelem) may be NULL when no integrand reads the state; aux (Np, naux, nelem) is
 * updated in place on the real elements; Imat = HOST (Nq, Nq) column-major grid.Imat[dim]
 * (Grids.jl:1184-1207).  Enqueued on the compute stream. */
int cmdg_indefinite_stack_integral(cmdg_handle h, const double *Q, int32_t nstate, double *aux,
                                   int32_t naux, int32_t nvertelem, const double *Imat,
                                   const cmdg_stack_integral_desc *d);
int cmdg_reverse_indefinite_stack_integral(cmdg_handle h, double *aux, int32_t naux,
                                           int32_t nvertelem, const cmdg_stack_integral_desc *d);

/* ---- element filters (src/Numerics/Mesh/Filters.jl) ------------------------------ */
typedef struct cmdg_filter_s *cmdg_filter;
/* AbstractFilter: spectral = Exponential / BoydVandeven / Cutoff (Filters.jl:172-307,
 * kernel_apply_filter! :651-794); MassPreservingCutoffFilter (:316-347, kernel :893-1071);
 * TMARFilter (:369, kernel :796-884) */
enum { CMDG_FILTER_SPECTRAL = 0, CMDG_FILTER_MASS_PRESERVING = 1, CMDG_FILTER_TMAR = 2 };
/* AbstractFilterTarget: FilterIndices (Filters.jl:72-100), AtmosFilterPerturbations and
 * AtmosSpecificFilterPerturbations (src/Atmos/Model/filters.jl:4-118, dry model: the
 * state is rho, rho u[3], rho e and the reference state is read from state_auxiliary) */
enum {
    CMDG_TARGET_INDICES = 0, CMDG_TARGET_ATMOS_PERTURBATIONS = 1,
    CMDG_TARGET_ATMOS_SPECIFIC_PERTURBATIONS = 2
};
#define CMDG_MAX_FILTER_STATES 32
typedef struct cmdg_filter_desc {
    int32_t kind;        /* CMDG_FILTER_* */
    int32_t target;      /* CMDG_TARGET_* */
    int32_t direction;   /* CMDG_*_DIRECTION keyword of Filters.apply! (spectral kinds) */
    int32_t nindices;    /* FilterIndices: number of filtered states */
    int32_t indices[CMDG_MAX_FILTER_STATES]; /* 1-based state indices */
    int32_t aux_ref_rho, aux_ref_rhoe; /* atmos targets: 0-based state_auxiliary columns of
                                          ref_state.rho and ref_state.rho e */
    const double *filter_h; /* HOST (Nq, Nq) column-major filter.filter_matrices[1] */
    const double *filter_v; /* HOST (Nq, Nq) column-major filter.filter_matrices[end] */
} cmdg_filter_desc;
/* filter object bound to a handle: the filter struct + target + direction of one
 * `Filters.apply!(Q, target, grid, filter; direction, state_auxiliary)` call site */
int cmdg_filter_create(cmdg_handle h, const cmdg_filter_desc *d, cmdg_filter *out);
int cmdg_filter_destroy(cmdg_handle h, cmdg_filter f);
/* Filters.apply_async! (Filters.jl:440-607): enqueue on the handle's compute stream, real
 * elements only, in place.  Q is a device array (Np, nstate, nelem); atmos targets read the
 * handle's state_auxiliary.  Follow with cmdg_synchronize for Filters.apply! (:408-421). */
int cmdg_filter_apply(cmdg_handle h, cmdg_filter f, double *Q, int32_t nstate);
/* filters the DG operator and the time stepper apply themselves:
 *   gradient_filter  on state_gradient_flux after the gradient pass   (DGModel.jl:185-193)
 *   tendency_filter  on the tendency at the end of the evaluation      (DGModel.jl:417-425)
 *   step_filter      on Q after every completed LSRK step -- the EveryXSimulationSteps(1)
 *                    callback of experiments/AtmosGCM/heldsuarez.jl:261-272
 * NULL clears a slot.  The filters must outlive their use. */
int cmdg_set_filters(cmdg_handle h, cmdg_filter gradient_filter, cmdg_filter tendency_filter,
                     cmdg_filter step_filter);

/* ---- law-specific update_auxiliary_state! / update_auxiliary_state_gradient! ----------- */
/* Laws whose auxiliary state needs more than a nodal refresh override these two methods in
 * the reference (BalanceLaws/interface.jl:276-305; called at DGModel.jl:110-116,161-172 and
 * :210-222,355-361).  Their bodies are compositions of operators this library has; the hooks
 * record such a composition and the operator runs it at the reference's call sites:
 *   before the gradient pass (real elements): pre_filter[i] applied to Q
 *       -- HBModel: vertical cutoff filter on u, exponential filter on theta
 *          (hydrostatic_boussinesq_model.jl:654-680)
 *   after the gradient pass (real elements, and ghost elements once their gradient flux
 *   arrived):  aux[:, copy_aux_col[i]] = copy_scale[i] * gradflux[:, copy_gf_col[i]];
 *       upward column integral; downward column integral; aux[:, surf_dst_col[i]] over each
 *       stack = aux[:, surf_src_col[i]] at the top node of the stack
 *       -- HBModel: w = -div_h u, (w, pkin) integrals, wz0 (:693-726) */
#define CMDG_MAX_HOOK_OPS 4
typedef struct cmdg_rhs_hooks {
    int32_t npre;
    cmdg_filter pre_filter[CMDG_MAX_HOOK_OPS];
    int32_t ncopy;
    int32_t copy_gf_col[CMDG_MAX_HOOK_OPS], copy_aux_col[CMDG_MAX_HOOK_OPS];
    double copy_scale[CMDG_MAX_HOOK_OPS];
    int32_t has_integral, has_reverse_integral;
    cmdg_stack_integral_desc integral;         /* src/scale/dst: upward integral */
    cmdg_stack_integral_desc reverse_integral; /* rsrc/rdst: downward integral; both act on
                                                  the handle's state_auxiliary */
    int32_t nsurf;
    int32_t surf_src_col[CMDG_MAX_HOOK_OPS], surf_dst_col[CMDG_MAX_HOOK_OPS];
    int32_t nvertelem;
    const double *Imat; /* HOST (Nq, Nq) column-major grid.Imat[dim] */
    /* compute_flow_deviation!(dg, ::HBModel, ::Coupled, Q, t)
     * (src/Ocean/SplitExplicit/HydrostaticBoussinesqCoupling.jl:43-85), run after the pre
     * filters: aux[flow_ud_col + c] = Q[flow_u_col + c] - (1 / flow_H) * (column integral of
     * Q[flow_u_col + c]), c = 0, 1 */
    int32_t has_flow_deviation, flow_u_col, flow_ud_col;
    double flow_H;
    /* SplitExplicit01's OceanModel does everything in update_auxiliary_state!
     * (src/Ocean/SplitExplicit01/OceanModel.jl:432-541), i.e. BEFORE the gradient pass, and takes
     * the horizontal divergence from a DG operator of its own (Continuity3dModel):
     *   pre filters; pre_rhs_handle (if not NULL) is evaluated on Q with increment = false and
     *   column pre_rhs_src_col of its tendency goes to auxiliary column pre_rhs_dst_aux_col;
     *   then, with ops_before_gradients != 0, the integral / reverse integral / surface
     *   operations above run here instead of after the gradient pass; then the flow deviation.
     * Partitioned grids: the nested operator lives on the same partition (same neighbours) and
     * exchanges with its own communicator (RCCL: cmdg_comm_init_rccl on it too; local transport:
     * the nested operators form a group of their own, and cmdg_group_rhs runs them in lock step
     * between the two halves of the composition).  After the exchange of Q the operator repeats
     * the column operators over the received face pencils of the ghost stacks (the kinematic
     * pressure rank-boundary faces read on their plus side), as it does for the flow deviation.
     * Lifetime: the nested handle may be destroyed first -- cmdg_destroy detaches it from every
     * handle whose hooks name it; such a handle then fails every evaluation with CMDG_ERR_INVALID
     * ("the nested operator of this handle was destroyed") until cmdg_set_rhs_hooks gives it new
     * hooks (or NULL): it never goes on computing a different law. */
    int32_t ops_before_gradients;
    cmdg_handle pre_rhs_handle;
    int32_t pre_rhs_src_col, pre_rhs_dst_aux_col;
} cmdg_rhs_hooks;
/* hooks == NULL clears them.  Filters must outlive their use. */
int cmdg_set_rhs_hooks(cmdg_handle h, const cmdg_rhs_hooks *hooks);

/* ---- split-explicit ocean (src/Numerics/ODESolvers/SplitExplicitMethod.jl:1-177 and
 * src/Ocean/SplitExplicit/Communication.jl) --------------------------------------------
 * `slow` is the 3-D HydrostaticBoussinesqModel handle (Coupled), `fast` the ShallowWaterModel
 * handle on the one-layer extrusion of the 2-D grid; both share the horizontal element order
 * and the horizontal polynomial order.  Columns are 0-based; vector fields take two
 * consecutive columns. */
typedef struct cmdg_ocean_coupling_desc {
    int32_t nvertelem;  /* stack size of the slow grid */
    double H;           /* problem.H */
    const double *Imat; /* HOST (Nq, Nq) column-major vertical grid.Imat of the slow grid */
    int32_t slow_u_col, slow_eta_col;  /* slow state: u[2], eta */
    int32_t slow_dGu_col;              /* slow auxiliary: dG_u[2] */
    int32_t fast_eta_col, fast_U_col;  /* fast state: eta, U[2] */
    int32_t fast_GU_col, fast_du_col;  /* fast auxiliary: G_U[2], Delta_u[2] */
} cmdg_ocean_coupling_desc;
/* initialize_states! (Communication.jl:1-12): slow.aux.dG_u = -0 */
int cmdg_ocean_initialize_states(cmdg_handle slow, cmdg_handle fast,
                                 const cmdg_ocean_coupling_desc *d);
/* tendency_from_slow_to_fast! (Communication.jl:14-70): int du = column integral of the
 * slow tendency dQ_slow's u; fast.aux.G_U = int du, slow.aux.dG_u -= int du / H */
int cmdg_ocean_tendency_from_slow_to_fast(cmdg_handle slow, cmdg_handle fast,
                                          const cmdg_ocean_coupling_desc *d,
                                          const double *dQ_slow);
/* reconcile_from_fast_to_slow! (Communication.jl:100-170): fast.aux.Delta_u = 1/H (U - int u),
 * Q_slow.u += Delta_u through the column, Q_slow.eta = Q_fast.eta */
int cmdg_ocean_reconcile_from_fast_to_slow(cmdg_handle slow, cmdg_handle fast,
                                           const cmdg_ocean_coupling_desc *d, double *Q_slow,
                                           const double *Q_fast);
/* ---- the older split-explicit ocean, src/Ocean/SplitExplicit01 -------------------------
 * dostep!(Qvec, split::SplitExplicitLSRK2nSolver, param, time)
 * (SplitExplicitLSRK2nMethod.jl:81-190) with the exchange functions of
 * src/Ocean/SplitExplicit01/Communication.jl: per slow stage initialize_fast_state! (number and
 * size of the barotropic sub-steps, averaging window), initialize_adjustment!, the slow
 * right-hand side twice (increment = false for the barotropic forcing, increment = true for the
 * stage), tendency_from_slow_to_fast!, update!, the sub-steps with cummulate_fast_solution!,
 * reconcile_from_fast_to_slow!.  `slow` = CMDG_PHYSICS_OCEAN_SE01 with its hooks installed,
 * `fast` = CMDG_PHYSICS_BAROTROPIC_SE01 on the one-layer extrusion of the 2-D grid, created on
 * the same device (CMDG_ERR_INVALID otherwise).
 * numImplSteps > 0 (implicit vertical diffusion, IVDCModel.jl) is not carried. */
typedef struct cmdg_ocean01_desc {
    int32_t nvertelem;          /* stack size of the slow grid */
    double H;                   /* problem.H */
    const double *Imat;         /* HOST (Nq, Nq) column-major vertical grid.Imat of the slow grid */
    int32_t add_fast_substeps;  /* OceanModel.add_fast_substeps */
    /* the fast solver's own LSRK scheme (split.fast_solver, SplitExplicitLSRK2nMethod.jl:150-165):
     * nstages_fast > 0 with its three HOST coefficient arrays; 0 = the slow solver's scheme (the
     * reference's simple_box_2dt.jl builds both solvers from LSRK54CarpenterKennedy) */
    int32_t nstages_fast;
    const double *rka_fast, *rkb_fast, *rkc_fast;
} cmdg_ocean01_desc;
int cmdg_split_explicit01_step(cmdg_handle slow, cmdg_handle fast, const cmdg_ocean01_desc *d,
                               double *Q3, double *dQ3, double *dQ2fast, double *Q2, double *dQ2,
                               double t, double dt, double dt_fast, int32_t nstages,
                               const double *rka, const double *rkb, const double *rkc);
/* the same step for the n (slow, fast) pairs of one process whose slow models, fast models and
 * nested continuity operators are each connected by cmdg_comm_connect_local (pair i = rank i);
 * with the RCCL transport every rank calls cmdg_split_explicit01_step on its own pair */
int cmdg_group_split_explicit01_step(cmdg_handle *slow, cmdg_handle *fast, int32_t n,
                                     const cmdg_ocean01_desc *d, double **Q3, double **dQ3,
                                     double **dQ2fast, double **Q2, double **dQ2, double t, double dt,
                                     double dt_fast, int32_t nstages, const double *rka,
                                     const double *rkb, const double *rkc);

/* update!() of the LSRK methods on the handle's real elements (LowStorageRungeKuttaMethod.jl:
 * 146-166): Q += rkb_dt * dQ; dQ *= rka_next */
int cmdg_lsrk_update(cmdg_handle h, double *dQ, double *Q, double rka_next, double rkb_dt);
/* dostep!(Q, ssp::StrongStabilityPreservingRungeKutta, p, time)
 * (src/Numerics/ODESolvers/StrongStabilityPreservingRungeKuttaMethod.jl:117-165, update! kernel
 * :167-190): Qstage = Q; per stage Rstage = rhs(Qstage, t + rkc[s] dt) (increment = false) and
 * Qstage = rka[2 s] Q + rka[2 s + 1] Qstage + dt rkb[s] Rstage; finally Q = Qstage.  rka is the
 * (nstages, 2) coefficient matrix row-major; Rstage and Qstage are caller-owned arrays of Q's
 * shape. */
int cmdg_ssprk_step(cmdg_handle h, double *Q, double *Rstage, double *Qstage, double t, double dt,
                    int32_t nstages, const double *rka, const double *rkb, const double *rkc);

/* dostep!(Q, lsrk3n::LowStorageRungeKutta3N, p, time)
 * (src/Numerics/ODESolvers/LowStorageRungeKutta3NMethod.jl:153-226, Fyfe 1966): dR = -0; per
 * stage dQ += rhs(Q, t + rkc[s] dt), then Q += rkb[2 s] dt dQ + rkb[2 s + 1] dt dR,
 * dR += rka[2 s' + 1] dQ, dQ *= rka[2 s'] with s' = (s + 1) mod nstages.  rka and rkb are the
 * (nstages, 2) matrices row-major; dQ (zero before the first step) and dR are caller-owned
 * arrays of Q's shape. */
int cmdg_ls3n_step(cmdg_handle h, double *Q, double *dQ, double *dR, double t, double dt,
                   int32_t nstages, const double *rka, const double *rkb, const double *rkc);

/* dostep!(Qslow, split::SplitExplicitSolver, param, time) (SplitExplicitMethod.jl:70-177): one
 * slow step of size dt_slow whose every stage sub-steps the fast model with full LSRK steps
 * of at most dt_fast.  dQ_slow / dQ_fast are the LSRK tendency accumulators (zero before the
 * first step), dQ2fast a scratch of the slow state's shape.  coupled == 0 runs the two
 * models side by side without the exchange (the `Uncoupled` variant of the test). */
int cmdg_split_explicit_step(cmdg_handle slow, cmdg_handle fast,
                             const cmdg_ocean_coupling_desc *d, int32_t coupled, double *Q_slow,
                             double *dQ_slow, double *dQ2fast, double *Q_fast, double *dQ_fast,
                             double t, double dt_slow, double dt_fast, int32_t nstages,
                             const double *rka, const double *rkb, const double *rkc);

/* the same step for n (slow, fast) pairs connected by cmdg_comm_connect_local (slow handles
 * among themselves, fast handles among themselves), driven in lock step by one host thread; the
 * element partition keeps whole columns, and slow[i] / fast[i] own the same columns */
int cmdg_group_split_explicit_step(cmdg_handle *slow, cmdg_handle *fast, int32_t n,
                                   const cmdg_ocean_coupling_desc *d, int32_t coupled,
                                   double **Q_slow, double **dQ_slow, double **dQ2fast,
                                   double **Q_fast, double **dQ_fast, double t, double dt_slow,
                                   double dt_fast, int32_t nstages, const double *rka,
                                   const double *rkb, const double *rkc);

/* ---- measurement --------------------------------------------------------------- */
enum {
    CMDG_K_GRADIENTS = 0, CMDG_K_DIVGRAD = 1, CMDG_K_GRADLAP = 2, CMDG_K_TENDENCY = 3,
    CMDG_K_PACK = 4, CMDG_K_UNPACK = 5, CMDG_K_UPDATE_AUX = 6, CMDG_K_FILTER = 7,
    CMDG_K_STACK_INTEGRAL = 8,
    /* halo: the transport between pack and unpack (RCCL group or device copies) on the halo
     * stream, and the time the compute stream had to wait for an exchange to finish (the part of
     * an exchange NOT hidden behind interior work; one record per exchange, zero when hidden) */
    CMDG_K_TRANSPORT = 9, CMDG_K_HALO_EXPOSED = 10,
    /* the exterior launches of the four passes of a handle with neighbours (their interior
     * launches stay under CMDG_K_GRADIENTS ... CMDG_K_TENDENCY) */
    CMDG_K_GRADIENTS_EXT = 11, CMDG_K_DIVGRAD_EXT = 12, CMDG_K_GRADLAP_EXT = 13,
    CMDG_K_TENDENCY_EXT = 14, CMDG_K_COUNT = 15
};
/* bracket every launch with HIP events on the launch stream (off by default) */
int cmdg_profile_enable(cmdg_handle h, int32_t on);
int cmdg_profile_get(cmdg_handle h, int32_t kernel, double *total_ms, int64_t *launches);
int cmdg_profile_reset(cmdg_handle h);

#ifdef __cplusplus
}
#endif
#endif /* CMDG_H */
