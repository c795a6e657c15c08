// Host-side engine: owns streams, events, halo buffers and work buffers of one handle
// and enqueues the passes of one right-hand-side evaluation in the order of the
// reference's `(dg::DGModel)(tendency, Q, _, t, alpha, beta)`
// (src/Numerics/DGMethods/DGModel.jl:85-427).  The physics/polynomial-order specific
// kernel launches live in EngineT<P, NQ>.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cmdg.h"
#include "kernels.h"

namespace cmdg {

enum { SLOT_Q = 0, SLOT_GF = 1, SLOT_HG = 2, SLOT_HD = 3, NSLOT = 4 };
enum { TRANSPORT_NONE = 0, TRANSPORT_LOCAL = 1, TRANSPORT_RCCL = 2 };

struct HaloSlot {
    double *sendbuf = nullptr, *recvbuf = nullptr;
    hipEvent_t ev_packed = nullptr, ev_done = nullptr, ev_pulled = nullptr;
    bool active = false;  // begin issued, end pending
    int nvar = 0;         // columns per position of the packed buffers
    int ncol = 0;         // columns of the array (>= nvar: the leading nvar travel)
    double *array = nullptr;
    // the array whose nodes of vmapsend an exterior launch has already written to sendbuf
    // (halo_pack then launches nothing); NULL: sendbuf is stale
    const double *fresh_for = nullptr;
    int fresh_nvar = 0;
};

struct RhsCtx {
    double *tendency = nullptr;
    double *Qin = nullptr;   // state read by this evaluation (ghosts refreshed in place)
    double *Qout = nullptr;  // LSRK: updated state
    double t = 0, alpha = 1, beta = 0;
    const double *tptr = nullptr;  // time in device memory instead (captured steps)
    bool lsrk = false;         // fused update inside k_tendency
    bool update_after = false; // separate update!() after the (filtered) tendency
    double rkb_dt = 0, rka_next = 0;
    // the law's update_auxiliary_state!(realelems) composition has run already (group_rhs runs the
    // nested operators of a local group in lock step before segment 0)
    bool pre_done = false;
};

// one `Filters.apply!` call site: filter + target + direction (include/cmdg.h)
struct FilterObj {
    int kind = 0, target = 0, direction = 0;
    int nindices = 0;
    int indices[CMDG_MAX_FILTER_STATES] = {0};
    int aux_ref_rho = 0, aux_ref_rhoe = 0;
    double *d_Fh = nullptr, *d_Fv = nullptr;
};

// roctx range around the host-side enqueue of a phase (the reference instruments the same five
// halo phases with NVTX, MPIStateArrays.jl:419-439,465-480): a no-op unless the roctx library is
// in the process (rocprofv3 --marker-trace) or CMDG_ROCTX=1 asks for it to be loaded
void roctx_push(const char *name);
void roctx_pop();
// CMDG_DBG_SYNC=<bitmask>: localise a missing stream dependency by turning one class of
// event edges at a time into a host-side hipStreamSynchronize (scripts/probe/priority_order_sweep.sh)
//   1 order() of the split-explicit steppers   2 halo_pack: compute -> halo stream
//   4 halo_pack: neighbours' ev_pulled         8 halo_end: neighbours' ev_packed
//   16 halo_end: ev_done -> compute stream     32 before_direct_send
//   64 interior_begin / exterior_begin         128 the join at the end of segment 5
//   256 device synchronize before every group_rhs   512 device synchronize after every segment
int dbg_sync();
hipError_t ev_record(hipEvent_t &e, hipStream_t s);
struct Range {
    explicit Range(const char *name) { roctx_push(name); }
    ~Range() { roctx_pop(); }
    Range(const Range &) = delete;
    Range &operator=(const Range &) = delete;
};

struct ProfRec {
    int kernel;
    hipEvent_t e0, e1;
    bool clamp;  // record max(0, elapsed): e1 may precede e0 (exposed halo time)
};

// CMDG_OPT_ASYNC_RUN: cmdg_lsrk_run hands the run to a thread of the handle's own and returns; the
// caller's thread is not the one that spends a millisecond per step inside hipGraphLaunch (or
// posting RCCL groups).  One job at a time, in order; every other ABI entry of the handle first
// waits until the worker is idle (DevGuard), so the handle is still driven by one thread at a time.
// A failure of a deferred run is reported by the next cmdg_synchronize.
struct RunWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<int()>> jobs;
    bool stop = false, busy = false;
    int deferred_rc = 0;
    std::string deferred_err;
    std::thread::id tid;
    void start()
    {
        th = std::thread([this] {
            std::unique_lock<std::mutex> lk(m);
            for (;;) {
                cv.wait(lk, [this] { return stop || !jobs.empty(); });
                if (jobs.empty()) return;  // (stop, drained)
                auto job = std::move(jobs.front());
                jobs.pop_front();
                busy = true;
                lk.unlock();
                const int r = job();
                lk.lock();
                busy = false;
                if (r && !deferred_rc) deferred_rc = r;
                cv.notify_all();
            }
        });
        tid = th.get_id();
    }
    void submit(std::function<int()> f)
    {
        {
            std::lock_guard<std::mutex> lk(m);
            jobs.push_back(std::move(f));
        }
        cv.notify_all();
    }
    void wait_idle()
    {
        if (std::this_thread::get_id() == tid) return;  // (the worker's own calls)
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this] { return jobs.empty() && !busy; });
    }
    ~RunWorker()
    {
        if (th.joinable()) {
            {
                std::lock_guard<std::mutex> lk(m);
                stop = true;
            }
            cv.notify_all();
            th.join();
        }
    }
};

struct EngineBase {
    RunWorker *worker = nullptr;  // CMDG_OPT_ASYNC_RUN
    // ---- configuration (copied from cmdg_desc) ---------------------------------------
    int NQ = 0, NQV = 0, Np = 0, Nfp = 0;  // horizontal / vertical points per direction
    int64_t nreal = 0, nghost = 0, nelem = 0;
    int ns = 0, naux = 0, ngrad = 0, ngf = 0, ngl = 0, nhyp = 0;
    int nf_first = 0, direction = 0, diffusion_direction = 0, stacked = 0;
    GridDev g{};
    const int64_t *d_interior = nullptr, *d_exterior = nullptr;
    // CMDG_OPT_STACK_HEIGHT: the caller's lists and the engine's own tiled copies of them
    const int64_t *d_interior_user = nullptr, *d_exterior_user = nullptr;
    int64_t *d_interior_tiled = nullptr, *d_exterior_tiled = nullptr;
    int set_stack_height(int nv);
    int set_stream_priority(int level);  // CMDG_OPT_STREAM_PRIORITY
    int stream_priority = 0;
    // CMDG_OPT_TENDENCY_PAIRS: the tendency pass takes horizontally adjacent elements two to a
    // work-group and keeps the xi1 face they share on chip (TendencyShape<..., PAIR>, kernels.h);
    // the lists: (e0, e1) 1-based per work-group, e1 < 0 unrelated, 0 none
    bool tendency_four_waves = false;  // CMDG_OPT_TENDENCY_FOUR_WAVES
    bool tendency_pairs = false;
    int64_t *d_pairs[2] = {nullptr, nullptr};  // interior, exterior
    int64_t npairs[2] = {0, 0};                // work-groups
    int64_t nshared[2] = {0, 0};               // ... of which share a face
    int build_pairs();
    virtual bool law_pairable() const = 0;
    int64_t ninterior = 0, nexterior = 0;
    const uint8_t *d_activedofs = nullptr;
    double *d_D = nullptr;
    int32_t *d_faceP = nullptr;  // digested face tables (GridDev::faceP / faceG)
    double *d_faceG = nullptr;
    const int64_t *d_vmapsend = nullptr, *d_vmaprecv = nullptr;
    int64_t nvmapsend = 0, nvmaprecv = 0;
    std::vector<int> nabrtorank;
    std::vector<int64_t> nabrsend, nabrrecv;  // 2*nnabr (first,last) 1-based
    double *aux = nullptr, *gf = nullptr, *hypgrad = nullptr, *hypdiv = nullptr;
    double *derived = nullptr;  // (Np, NDER, nelem), library-owned
    bool own_gf = false, own_hg = false, own_hd = false;
    // the caller's Qhypervisc_grad / state_gradient_flux (reference layout) when the working copy is
    // node-major: written by cmdg_export_* only; gf_scratch: reference-layout copy for a gradient filter
    double *hypgrad_user = nullptr, *gf_user = nullptr, *gf_scratch = nullptr;
    bool node_major(const double *array) const
    {
        return (CMDG_HG_NODE_MAJOR && array == hypgrad) || (array == gf && gf_node_major());
    }
    // ---- runtime -----------------------------------------------------------------------
    int dev = 0;  // the device this engine was created on (every ABI entry binds to it)
    hipStream_t s_comp = nullptr, s_comm = nullptr;
    hipEvent_t ev_comp = nullptr;
    HaloSlot slot[NSLOT];
    int slot_nvar_max = 0;
    double *W[2] = {nullptr, nullptr};  // LSRK work states
    double *d_partial = nullptr;        // reduction scratch
    int transport = TRANSPORT_NONE;
    int rank = 0, nranks = 1;
    std::vector<EngineBase *> group;    // local transport: engine of every rank
    void *nccl_comm = nullptr;
    const FilterObj *gradient_filter = nullptr, *tendency_filter = nullptr,
                    *step_filter = nullptr;
    bool profiling = false;
    std::vector<ProfRec> prof;
    double prof_ms[CMDG_K_COUNT] = {0};
    int64_t prof_n[CMDG_K_COUNT] = {0};
    std::string err;

    virtual ~EngineBase();
    int init(const cmdg_desc *d);
    int fail(int code, const std::string &msg)
    {
        err = msg;
        return code;
    }
    bool communicate() const { return !nabrtorank.empty(); }

    // physics / order specific launches; `exterior`: the launch of the exterior element list of a
    // handle with neighbours (writes the send buffers of what it produces, see HaloDev)
    virtual void launch_gradients(const RhsCtx &c, const int64_t *elems, int64_t n, bool exterior, hipStream_t st) = 0;
    virtual void launch_divgrad(const RhsCtx &c, const int64_t *elems, int64_t n, bool exterior, hipStream_t st) = 0;
    virtual void launch_gradlap(const RhsCtx &c, const int64_t *elems, int64_t n, bool exterior, hipStream_t st) = 0;
    virtual void launch_tendency(const RhsCtx &c, const int64_t *elems, int64_t n, bool exterior, hipStream_t st) = 0;
    virtual void launch_update_aux(const RhsCtx &c, int64_t e0, int64_t e1) = 0;
    virtual bool has_update_aux() const = 0;
    virtual bool law_needs_gradflux() const = 0;
    virtual bool gf_node_major() const = 0;  // state_gradient_flux kept (ngf, Np, nelem) inside the library
    bool keep_gradflux = false;  // CMDG_OPT_KEEP_GRADFLUX
    // is state_gradient_flux formed (and exchanged) by an evaluation?
    bool gf_live() const
    {
        return ngf > 0 && (keep_gradflux || law_needs_gradflux() || has_hooks || gradient_filter);
    }
    virtual bool fused_update_aux() const = 0;
    // introspection (cmdg_query): per-node columns of the law's time-invariant derived fields,
    // auxiliary columns its nodal refresh rewrites, elements per work-group of the tendency pass
    virtual int law_nder() const = 0;
    virtual int law_nupd() const = 0;
    virtual int tendency_epb() const = 0;
    virtual int law_state_read(int pass) const = 0;
    virtual int law_aux_read(int pass) const = 0;
    virtual int init_derived() = 0;
    // mode 0: per-element minimum node distance, mode 1: per-element maximum Courant number
    virtual int launch_courant(int mode, int kind, const double *Q, double dt, double t, int dir,
                               double *out_elem) = 0;

    // ---- ghost exchange without pack / unpack launches (HaloDev, cmdg_common.h) -------------
    // tables built at create from vmapsend / vmaprecv / the digested face table; *_ok = they
    // could be built (every node of vmapsend in an exterior element, every ghost node received
    // once, every ghost node a face reads received)
    int32_t *d_sendoff = nullptr, *d_ghostslot = nullptr;
    SendEnt *d_sendent = nullptr;
    bool direct_send_ok = false, direct_recv_ok = false;
    bool reference_halo = false;  // CMDG_OPT_REFERENCE_HALO: pack and unpack as the reference does
    int init_halo_tables();
    bool direct_send() const { return direct_send_ok && !reference_halo; }
    // consumers may read the receive buffers unless somebody reads the ghost ELEMENTS of Q: a
    // nodal update_auxiliary_state! of the ghosts that is not fused away, or the hooks
    bool direct_recv() const
    {
        return direct_recv_ok && !reference_halo && !has_hooks && !(has_update_aux() && !fused_update_aux());
    }
    // what a launch is handed: receive side for every launch, send side for exterior ones
    HaloDev halo_dev(bool exterior, double *send0, double *send1) const
    {
        HaloDev h{};
        h.nreal = nreal;
        if (!communicate()) return h;
        if (exterior && direct_send()) {
            h.sendoff = d_sendoff;
            h.sendent = d_sendent;
            h.send[0] = send0;
            h.send[1] = send1;
        }
        if (direct_recv()) {
            h.ghostslot = d_ghostslot;
            h.recvQ = slot[SLOT_Q].recvbuf;
            h.recvGF = slot[SLOT_GF].recvbuf;
            h.recvHG = slot[SLOT_HG].recvbuf;
            h.recvHD = slot[SLOT_HD].recvbuf;
        }
        return h;
    }
    // an exterior launch (on stream st) is about to overwrite sendbuf of slot s: with the local
    // transport the neighbours must have pulled its previous payload
    int before_direct_send(int s, hipStream_t st);
    // ---- two pipelines (handles with neighbours whose exchanges run direct both ways) ---------
    // The exterior launches E_p of the passes and the exchanges X_p they feed form a serial chain
    // X_p -> E_p -> X_(p+1) -> E_(p+1) ...; it runs on the halo stream with no event hop inside,
    // while the interior launches I_p run on the compute stream.  A pass reads what the previous
    // pass wrote for the element and its face neighbours, so I_p waits for E_(p-1) and E_p waits
    // for I_(p-1) (events of alternating parity): the interior work of a pass hides the exchanges
    // of two, and a step costs max(chain, compute) instead of the sum over passes of
    // max(I_p, X_p) + E_p.
    hipEvent_t ev_int[2] = {nullptr, nullptr}, ev_ext[2] = {nullptr, nullptr};
    int64_t pass_seq = 0;  // passes started on this handle
    int64_t host_post_ns = 0, host_post_n = 0;  // host time inside halo_post (RCCL group calls)
    hipEvent_t prof_ext_done = nullptr;  // profiling: end of the last exterior launch
    bool no_pipeline = false;  // CMDG_OPT_HALO_PIPELINE = 0
    bool pipelined(bool comm) const
    {
        return comm && !no_pipeline && direct_send() && direct_recv() && !gradient_filter &&
               !tendency_filter && (!has_update_aux() || fused_update_aux());
    }
    void invalidate_sends()
    {
        for (auto &h : slot) h.fresh_for = nullptr;
    }
    void mark_fresh(int s, const double *array, int nvar)
    {
        slot[s].fresh_for = array;
        slot[s].fresh_nvar = nvar;
    }

    // ---- a whole LSRK step as a HIP graph (CMDG_OPT_STEP_GRAPH) ---------------------------
    // cmdg_lsrk_run can record one step into a HIP graph and replay it: the evaluation times come
    // from device memory, advanced by a one-thread kernel at the head of the graph exactly as
    // updatetime! accumulates them; the first step of every run is issued eagerly, the capture uses
    // events of its own.  Built for the partitioned case -- at 5 400 elements per rank a step is
    // bound by the HOST: posting an RCCL group costs 55 us of host time, 20 of them per step, next
    // to 40 kernel launches and 100 event operations (1.43 ms of enqueueing for a 1.56 ms step) --
    // but RCCL operations inside a capture crash hipStreamEndCapture on this stack, so handles that
    // exchange stay eager (graph_eligible) and the option serves single-rank handles only, where
    // the device is the bound anyway.  Anything else a capture cannot hold (profiling, filters,
    // hooks, an unfused nodal refresh) keeps a run eager too.
    bool step_graph = false;        // the option
    bool capturing = false;         // rhs_segment is being recorded
    int cap_interior = 0, cap_exterior = 0;  // launches begun in this capture
    // (every record of a capture gets an event of its own: 4 passes x 16 stages at most)
    static constexpr int NGEV = 64;
    hipEvent_t gev_int[NGEV] = {nullptr}, gev_ext[NGEV] = {nullptr}, gev_fork = nullptr;
    int cap_pass = 0;  // passes begun in this capture
    hipGraphExec_t graph_exec = nullptr;
    struct GraphKey {
        const double *Q = nullptr, *dQ = nullptr;
        double dt = 0;
        int nstages = 0;
        double coef[48] = {0};
        bool pipe = false, comm = false;
        bool operator==(const GraphKey &o) const
        {
            if (Q != o.Q || dQ != o.dQ || dt != o.dt || nstages != o.nstages || pipe != o.pipe || comm != o.comm)
                return false;
            for (int i = 0; i < 48; ++i)
                if (coef[i] != o.coef[i]) return false;
            return true;
        }
    } graph_key;
    double *d_gtime = nullptr;      // [t_next, dt, times[16], rkc[16]]
    int64_t graph_steps = 0;        // steps replayed from the graph (cmdg_query)
    bool graph_failed = false;      // a capture failed: this handle stays eager
    bool graph_eligible() const;
    // whatever changes the launches of an evaluation (options, filters, hooks, profiling) makes a
    // recorded step stale: the next run records again
    void drop_graph()
    {
        if (graph_exec) {
            hipStreamSynchronize(s_comp);
            hipGraphExecDestroy(graph_exec);
            graph_exec = nullptr;
        }
    }
    int capture_step(double *Q, double *dQ, double dt, int nstages, const double *rka,
                     const double *rkb, const double *rkc);
    int run_steps(double *Q, double *dQ, double t, double dt, int64_t nsteps, int nstages,
                  const double *rka, const double *rkb, const double *rkc);

    // orchestration
    static constexpr int NSEG = 6;
    int rhs_segment(int seg, const RhsCtx &c);
    int rhs_async(const RhsCtx &c);
    int lsrk_step(double *Q, double *dQ, double t, double dt, int nstages, const double *rka,
                  const double *rkb, const double *rkc, bool continued = false);
    // (nvar columns per packed position = the leading columns of the ncol-column array; ncol = 0:
    // the whole array, nvar == ncol, as the reference packs)
    // on_halo_stream (pipelined()): producer and consumer are launches of the halo stream itself
    int export_hypgrad(double *dst);
    int export_gradflux(double *dst);
    int halo_begin(int s, double *array, int nvar, int ncol = 0, bool on_halo_stream = false);
    // begin_ghost_exchange! in two halves, so that exchanges that begin at the same point of an
    // evaluation are packed one after the other and posted in ONE RCCL group
    int halo_pack(int s, double *array, int nvar, int ncol = 0, bool on_halo_stream = false);
    int halo_post(const int *slots, int nslots);
    // unpack = false: the consumers read the receive buffer (direct_recv())
    int halo_end(int s, double *array, int nvar, bool unpack = true, bool on_halo_stream = false);
    void abort_exchanges();  // after a failed call: no exchange is left "begun"
    int ensure_work();
    int synchronize();
    int wsum2(const double *A, const double *B, int nvar, int weighted, double *out);
    int courant(int mode, int kind, const double *Q, double dt, double t, int dir, double *out);
    double *d_elemred = nullptr;  // (nreal) per-element extrema
    int stack_integral(bool reverse, const double *Q, int nstate, double *aux_arr, int naux_arr,
                       int nvert, const double *Imat_host, const cmdg_stack_integral_desc *d,
                       int64_t h0 = 0, int64_t nh = -1);
    bool has_hooks = false;
    // the nested operator of the hooks was destroyed: evaluations fail until new hooks are set
    bool hooks_orphaned = false;
    cmdg_rhs_hooks hooks{};
    // handles whose hooks evaluate this one as their nested operator (hooks.pre_rhs_handle):
    // cmdg_destroy of this handle detaches it from them
    std::vector<EngineBase *> nested_in;
    int set_hooks(const cmdg_rhs_hooks *hk);
    int run_pre_hooks(const RhsCtx &c);
    // ... in two halves around the evaluation of the nested operator (hooks.pre_rhs_handle)
    int run_pre_hooks_a(const RhsCtx &c, RhsCtx &nested);
    int run_pre_hooks_b(const RhsCtx &c);
    int integrate_velocity(const double *X, int nstate, int col, int nvert, int64_t h0 = 0,
                           int64_t nh = -1);
    int flow_deviation(double *Q, int64_t h0, int64_t nh);
    double *d_flowint = nullptr;  // (Np, 2, nelem) column integral of the horizontal velocity
    double *d_preT = nullptr;     // tendency of the nested operator of the hooks (pre_rhs_handle)
    int run_column_ops(const RhsCtx &c, int64_t e0, int64_t e1);
    // the column operators of a recorded composition in one launch (columns.h k_column_chain,
    // k_flow_deviation); CMDG_FUSED_COLUMNS=0 issues them one by one as recorded (A/B, tests)
    // (levels, for A/B: 1 the hooks' column operators, 2 + the pair of pre filters; fusing the
    // stepper's coupling kernels as well was measured slower, profiles/r04_ab_ocean_fused_columns.txt)
    int fused_columns = 2;
    bool column_chain(const RhsCtx &c, int64_t e0, int64_t e1, bool with_copies);
    bool filter_pair(double *Q);
    int run_gradient_hooks(const RhsCtx &c, int64_t e0, int64_t e1);
    double *d_Imat = nullptr;
    double *d_Dv = nullptr;  // vertical derivative matrix when the vertical order differs
    int filter_create(const cmdg_filter_desc *d, FilterObj **out);
    int filter_apply(const FilterObj *f, double *Q, int nstate);

    // profiling brackets
    void prof_begin(int kernel, hipStream_t st);
    void prof_end(hipStream_t st);
    void prof_collect();
};

}  // namespace cmdg

struct cmdg_context {
    cmdg::EngineBase *eng = nullptr;
    std::string err;
};

namespace cmdg {
// Every ABI entry runs with the engine's device current: lazily allocated work buffers and the
// kernels of a handle land on the GPU the handle was created on, whatever device the calling
// thread switched to in between; the caller's current device is restored on return.
struct DevGuard {
    int prev = -1;
    bool changed = false;
    explicit DevGuard(const EngineBase *e)
    {
        if (e->worker) e->worker->wait_idle();  // deferred runs of this handle come first
        if (hipGetDevice(&prev) == hipSuccess && prev != e->dev)
            changed = hipSetDevice(e->dev) == hipSuccess;
    }
    ~DevGuard()
    {
        if (changed) (void)hipSetDevice(prev);
    }
    DevGuard(const DevGuard &) = delete;
    DevGuard &operator=(const DevGuard &) = delete;
};
}  // namespace cmdg

namespace cmdg {

int group_rhs(std::vector<EngineBase *> &g, std::vector<RhsCtx> &c, bool keep_fresh = false);
int group_lsrk_step(std::vector<EngineBase *> &g, double **Q, double **dQ, double t, double dt,
                    int nstages, const double *rka, const double *rkb, const double *rkc,
                    bool continued = false, const double *stage_times_dev = nullptr);

// ---------------------------------------------------------------------------------
template <class P, int NQ_, int NQV_ = NQ_>
struct EngineT : EngineBase {
    typename P::Params prm;
    PassArgs<P> make_args(const RhsCtx &c, const int64_t *elems, int64_t n, int dir) const
    {
        PassArgs<P> a;
        a.prm = prm;
        a.g = g;
        a.elems = elems;
        a.nelems = n;
        a.Q = c.Qin;
        a.aux = aux;
        a.aux_rw = aux;
        a.derived = derived;
        a.gf = gf;
        a.hypgrad = hypgrad;
        a.hypdiv = hypdiv;
        a.tendency = c.tendency;
        a.Qout = c.Qout;
        a.t = c.t;
        a.tptr = c.tptr;
        a.alpha = c.alpha;
        a.beta = c.beta;
        a.rkb_dt = c.rkb_dt;
        a.rka_next = c.rka_next;
        a.direction = dir;
        a.model_dir = direction;
        a.nf_first = nf_first;
        a.h = HaloDev{};
        a.h.nreal = nreal;
        return a;
    }
    void launch_gradients(const RhsCtx &c, const int64_t *elems, int64_t n, bool exterior, hipStream_t st) override
    {
        if (n <= 0) return;
        Range range_(exterior ? "cmdg:gradients:exterior" : "cmdg:gradients");
        prof_begin(exterior ? CMDG_K_GRADIENTS_EXT : CMDG_K_GRADIENTS, st);
        PassArgs<P> args = make_args(c, elems, n, diffusion_direction);
        args.h = halo_dev(exterior, gf_live() ? slot[SLOT_GF].sendbuf : nullptr,
                          ngl > 0 ? slot[SLOT_HG].sendbuf : nullptr);
        if (gf_live())
            hipLaunchKernelGGL((k_gradients<P, NQ_, NQV_, true>), dim3((unsigned)n), dim3(KDims<NQ_, NQV_>::NT), 0,
                               st, args);
        else
            hipLaunchKernelGGL((k_gradients<P, NQ_, NQV_, false>), dim3((unsigned)n), dim3(KDims<NQ_, NQV_>::NT), 0,
                               st, args);
        prof_end(st);
    }
    void launch_divgrad(const RhsCtx &c, const int64_t *elems, int64_t n, bool exterior, hipStream_t st) override
    {
        if (n <= 0) return;
        Range range_(exterior ? "cmdg:divgrad:exterior" : "cmdg:divgrad");
        prof_begin(exterior ? CMDG_K_DIVGRAD_EXT : CMDG_K_DIVGRAD, st);
        PassArgs<P> args = make_args(c, elems, n, diffusion_direction);
        args.h = halo_dev(exterior, slot[SLOT_HD].sendbuf, nullptr);
        hipLaunchKernelGGL((k_divgrad<P, NQ_, NQV_>), dim3((unsigned)n), dim3(KDims<NQ_, NQV_>::NT), 0, st,
                           args);
        prof_end(st);
    }
    void launch_gradlap(const RhsCtx &c, const int64_t *elems, int64_t n, bool exterior, hipStream_t st) override
    {
        if (n <= 0) return;
        Range range_(exterior ? "cmdg:gradlap:exterior" : "cmdg:gradlap");
        prof_begin(exterior ? CMDG_K_GRADLAP_EXT : CMDG_K_GRADLAP, st);
        PassArgs<P> args = make_args(c, elems, n, diffusion_direction);
        args.h = halo_dev(exterior, slot[SLOT_HG].sendbuf, nullptr);
        hipLaunchKernelGGL((k_gradlap<P, NQ_, NQV_>), dim3((unsigned)n), dim3(KDims<NQ_, NQV_>::NT), 0, st,
                           args);
        prof_end(st);
    }
    void launch_tendency(const RhsCtx &c, const int64_t *elems, int64_t n, bool exterior, hipStream_t st) override
    {
        if (n <= 0) return;
        Range range_(exterior ? "cmdg:tendency:exterior" : "cmdg:tendency");
        prof_begin(exterior ? CMDG_K_TENDENCY_EXT : CMDG_K_TENDENCY, st);
#ifdef CMDG_GF_ALWAYS
        const bool gfl = true;
#else
        const bool gfl = P::needs_gradflux(prm);
#endif
        using SH = TendencyShape<P, NQ_, NQV_>;
        PassArgs<P> args = make_args(c, elems, n, direction);
        args.h = halo_dev(exterior, c.lsrk ? slot[SLOT_Q].sendbuf : nullptr, nullptr);
        const bool recv = args.h.ghostslot != nullptr && exterior;  // (interior elements have no ghost neighbour)
        if (!recv) args.h.ghostslot = nullptr;
        if constexpr (SH::PAIRABLE) {
            const int w = elems == d_exterior ? 1 : 0;
            if (tendency_pairs && d_pairs[w] && (elems == d_interior || elems == d_exterior)) {
                using SP = TendencyShape<P, NQ_, NQV_, true>;
                args.elems = d_pairs[w];
                args.nelems = 2 * npairs[w];
                const dim3 pgrid((unsigned)npairs[w]), pblock(SP::NT);
#define CMDG_TENDP(L, G)                                                                              \
    do {                                                                                              \
        if constexpr (KDims<NQ_, NQV_>::Np <= 125) {                                                  \
            constexpr size_t lds = sizeof(double) * TendencyLds<P, NQ_, NQV_, G, true>::doubles;      \
            if (recv)                                                                                 \
                hipLaunchKernelGGL((k_tendency_pair_small<P, NQ_, NQV_, L, G, true>), pgrid, pblock,  \
                                   lds, st, args);                                                    \
            else                                                                                      \
                hipLaunchKernelGGL((k_tendency_pair_small<P, NQ_, NQV_, L, G, false>), pgrid, pblock, \
                                   lds, st, args);                                                    \
        } else if (recv)                                                                              \
            hipLaunchKernelGGL((k_tendency<P, NQ_, NQV_, L, G, true, TEND_FUSED, true>), pgrid,       \
                               pblock, 0, st, args);                                                  \
        else                                                                                          \
            hipLaunchKernelGGL((k_tendency<P, NQ_, NQV_, L, G, false, TEND_FUSED, true>), pgrid,      \
                               pblock, 0, st, args);                                                  \
    } while (0)
                if (c.lsrk) {
                    if (gfl) CMDG_TENDP(true, true);
                    else CMDG_TENDP(true, false);
                } else {
                    if (gfl) CMDG_TENDP(false, true);
                    else CMDG_TENDP(false, false);
                }
#undef CMDG_TENDP
                prof_end(st);
                return;
            }
        }
        if constexpr (CMDG_TEND_FOUR_WAVES != 0 && KDims<NQ_, NQV_>::Np > 125 && node_cache_size<P>::value == 0 &&
                      NQ_ == NQV_) {
            if (tendency_four_waves) {  // CMDG_OPT_TENDENCY_FOUR_WAVES (k_tendency_big)
                const dim3 bgrid((unsigned)n), bblock(256);
#define CMDG_TENDB(L, G)                                                                                 \
    do {                                                                                                 \
        if (recv)                                                                                        \
            hipLaunchKernelGGL((k_tendency_big<P, NQ_, NQV_, L, G, true>), bgrid, bblock, 0, st, args);  \
        else                                                                                             \
            hipLaunchKernelGGL((k_tendency_big<P, NQ_, NQV_, L, G, false>), bgrid, bblock, 0, st, args); \
    } while (0)
                if (c.lsrk) {
                    if (gfl) CMDG_TENDB(true, true);
                    else CMDG_TENDB(true, false);
                } else {
                    if (gfl) CMDG_TENDB(false, true);
                    else CMDG_TENDB(false, false);
                }
#undef CMDG_TENDB
                prof_end(st);
                return;
            }
        }
        const dim3 grid((unsigned)SH::blocks(n)), block(SH::NT);
        // large elements: volume half, then interface half + update (TendencyShape::SPLIT)
#define CMDG_TEND(L, G)                                                                             \
    do {                                                                                            \
        if constexpr (SH::SPLIT) {                                                                  \
            hipLaunchKernelGGL((k_tendency<P, NQ_, NQV_, false, G, false, TEND_VOLUME>), grid,      \
                               dim3(SH::NTV), 0, st, args);                                         \
            if (recv)                                                                               \
                hipLaunchKernelGGL((k_tendency<P, NQ_, NQV_, L, G, true, TEND_FACES>), grid, block, \
                                   0, st, args);                                                    \
            else                                                                                    \
                hipLaunchKernelGGL((k_tendency<P, NQ_, NQV_, L, G, false, TEND_FACES>), grid,       \
                                   block, 0, st, args);                                             \
        } else if (recv)                                                                            \
            hipLaunchKernelGGL((k_tendency<P, NQ_, NQV_, L, G, true>), grid, block, 0, st, args);   \
        else                                                                                        \
            hipLaunchKernelGGL((k_tendency<P, NQ_, NQV_, L, G, false>), grid, block, 0, st, args);  \
    } while (0)
        if (c.lsrk) {
            if (gfl) CMDG_TEND(true, true);
            else CMDG_TEND(true, false);
        } else {
            if (gfl) CMDG_TEND(false, true);
            else CMDG_TEND(false, false);
        }
#undef CMDG_TEND
        prof_end(st);
    }
    void launch_update_aux(const RhsCtx &c, int64_t e0, int64_t e1) override
    {
        if constexpr (P::HAS_UPDATE_AUX) {
            if (e1 <= e0 || !P::update_aux_active(prm)) return;
            const int64_t n = (e1 - e0) * KDims<NQ_, NQV_>::Np;
            prof_begin(CMDG_K_UPDATE_AUX, s_comp);
            hipLaunchKernelGGL((k_update_aux<P, NQ_, NQV_>), dim3((unsigned)((n + 255) / 256)), dim3(256),
                               0, s_comp, prm, c.Qin, aux, d_activedofs, c.t, e0, e1);
            prof_end(s_comp);
        }
    }
    int launch_courant(int mode, int kind, const double *Q, double dt, double t, int dir,
                       double *out_elem) override
    {
        constexpr int NT = KDims<NQ_, NQV_>::Np <= 128 ? 128 : 256;
        if (mode == 1 && !P::HAS_COURANT)
            return fail(CMDG_ERR_UNSUPPORTED, "this balance law defines no local Courant number");
        if (mode == 0)
            hipLaunchKernelGGL((k_courant<P, NQ_, NQV_, 0>), dim3((unsigned)nreal), dim3(NT), 0, s_comp, prm,
                               g.vgeo, g.nvgeo, Q, aux, gf, kind, dt, t, dir, out_elem);
        else
            hipLaunchKernelGGL((k_courant<P, NQ_, NQV_, 1>), dim3((unsigned)nreal), dim3(NT), 0, s_comp, prm,
                               g.vgeo, g.nvgeo, Q, aux, gf, kind, dt, t, dir, out_elem);
        return CMDG_OK;
    }
    bool has_update_aux() const override { return P::HAS_UPDATE_AUX && P::update_aux_active(prm); }
    bool law_needs_gradflux() const override { return P::needs_gradflux(prm); }
    bool gf_node_major() const override { return cmdg::gf_node_major<P>::value; }
    bool fused_update_aux() const override { return P::HAS_UPDATE_AUX && P::FUSE_UPDATE_AUX; }
    int law_nder() const override { return P::HAS_SOURCE ? P::NDER : 0; }
    int law_nupd() const override { return P::HAS_UPDATE_AUX ? P::NUPD : 0; }
    int tendency_epb() const override { return TendencyShape<P, NQ_, NQV_>::EPB; }
    bool law_pairable() const override { return TendencyShape<P, NQ_, NQV_>::PAIRABLE; }
    int law_state_read(int pass) const override { return law_reads<P>::state(pass); }
    int law_aux_read(int pass) const override { return law_reads<P>::aux(pass); }
    int init_derived() override
    {
        if constexpr (P::NDER > 0) {
            const int64_t n = nelem * KDims<NQ_, NQV_>::Np;
            if (hipMalloc(&derived, sizeof(double) * n * P::NDER) != hipSuccess)
                return fail(CMDG_ERR_HIP, "hipMalloc(derived) failed");
            hipLaunchKernelGGL((k_init_derived<P, NQ_, NQV_>), dim3((unsigned)((n + 255) / 256)), dim3(256),
                               0, s_comp, prm, aux, derived, nelem);
            if (hipStreamSynchronize(s_comp) != hipSuccess)
                return fail(CMDG_ERR_HIP, "k_init_derived failed");
        }
        return CMDG_OK;
    }
};

// what a plug-in and the library must agree on (cmdg_load_plugin): the layout of the engine base
// class and of the descriptor, folded into one number
inline unsigned long engine_abi_stamp()
{
    return (unsigned long)sizeof(EngineBase) * 1000003ul + (unsigned long)sizeof(cmdg_desc) * 10007ul +
           (unsigned long)sizeof(RhsCtx) * 101ul + (unsigned long)sizeof(cmdg_rhs_hooks);
}

template <class P, int NQ_, int NQV_ = NQ_>
EngineBase *make_engine(const cmdg_desc *d)
{
    auto *e = new EngineT<P, NQ_, NQV_>();
    e->NQ = NQ_;
    e->NQV = NQV_;
    e->ns = P::NS;
    e->naux = P::NAUX;
    e->ngrad = P::NGRAD;
    e->ngf = P::NGF;
    e->ngl = P::NGL;
    e->nhyp = P::NHYP;
    P::make_params(e->prm, d->iparam, d->dparam);
    return e;
}

// factories implemented per physics family (one translation unit each)
EngineBase *make_engine_advdiff(const cmdg_desc *d, std::string &err);
int counts_advdiff(const int32_t *iparam, int32_t out[6]);
EngineBase *make_engine_atmos(const cmdg_desc *d, std::string &err);
int counts_atmos(const int32_t *iparam, int32_t out[6]);
EngineBase *make_engine_ocean(const cmdg_desc *d, std::string &err);
int counts_ocean(const int32_t *iparam, int32_t out[6]);
EngineBase *make_engine_sw(const cmdg_desc *d, std::string &err);
EngineBase *make_engine_moist(const cmdg_desc *d, std::string &err);
int counts_moist(const int32_t *iparam, int32_t out[6]);
int counts_sw(const int32_t *iparam, int32_t out[6]);
EngineBase *make_engine_pgrad(const cmdg_desc *d, std::string &err);
EngineBase *make_engine_se01(const cmdg_desc *d, std::string &err);
int counts_se01(int32_t physics_id, int32_t out[6]);
int counts_pgrad(const int32_t *iparam, int32_t out[6]);

}  // namespace cmdg
