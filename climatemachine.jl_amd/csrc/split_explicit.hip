// Split-explicit barotropic / baroclinic ocean stepper: the exchange functions of
// src/Ocean/SplitExplicit/Communication.jl and dostep! of
// src/Numerics/ODESolvers/SplitExplicitMethod.jl:70-177 over two engines (slow 3-D
// HydrostaticBoussinesqModel, fast ShallowWaterModel on the one-layer extrusion of the 2-D grid).
// Everything is enqueued on the slow engine's compute stream except the fast model's own
// sub-steps; the two streams are ordered with events, the host never waits.
#include <cmath>
#include <vector>

#include "columns.h"
#include "engine.h"
#include "filters.h"

using namespace cmdg;

namespace {

int set_err2(cmdg_handle h, int code)
{
    if (h && h->eng && code != CMDG_OK) h->err = h->eng->err;
    return code;
}

unsigned nblocks(int64_t n) { return (unsigned)std::min<int64_t>((n + 255) / 256, 65535); }

// make stream `later` wait for everything enqueued so far on `earlier`
int order(EngineBase *e, hipStream_t earlier, hipStream_t later)
{
    if (earlier == later) return CMDG_OK;
    if (dbg_sync() & 1) (void)hipStreamSynchronize(earlier);
    if (ev_record(e->ev_comp, earlier) != hipSuccess ||
        hipStreamWaitEvent(later, e->ev_comp, 0) != hipSuccess)
        return e->fail(CMDG_ERR_HIP, "split explicit: stream ordering failed");
    return CMDG_OK;
}

int check(cmdg_handle slow, cmdg_handle fast, const cmdg_ocean_coupling_desc *d)
{
    if (!slow || !fast || !d) return CMDG_ERR_INVALID;
    EngineBase *s = slow->eng, *f = fast->eng;
    if (!s->stacked || d->nvertelem < 1 || s->nreal % d->nvertelem)
        return s->fail(CMDG_ERR_INVALID, "ocean coupling: slow grid is not stacked by nvertelem");
    if (f->nreal != s->nreal / d->nvertelem)
        return s->fail(CMDG_ERR_INVALID, "ocean coupling: fast grid must hold one element per stack");
    if (f->Np % (s->NQ * s->NQ) || s->Np != s->NQ * s->NQ * s->NQ)
        return s->fail(CMDG_ERR_INVALID, "ocean coupling: horizontal polynomial orders differ");
    if (!(d->H > 0)) return s->fail(CMDG_ERR_INVALID, "ocean coupling: H");
    if (d->slow_u_col < 0 || d->slow_u_col + 2 > s->ns || d->slow_eta_col < 0 ||
        d->slow_eta_col >= s->ns || d->slow_dGu_col < 0 || d->slow_dGu_col + 2 > s->naux ||
        d->fast_eta_col < 0 || d->fast_eta_col >= f->ns || d->fast_U_col < 0 ||
        d->fast_U_col + 2 > f->ns || d->fast_GU_col < 0 || d->fast_GU_col + 2 > f->naux ||
        d->fast_du_col < 0 || d->fast_du_col + 2 > f->naux)
        return s->fail(CMDG_ERR_INVALID, "ocean coupling: column out of range");
    if (!s->d_Imat) {
        if (!d->Imat) return s->fail(CMDG_ERR_INVALID, "ocean coupling: Imat is NULL");
        if (hipMalloc(&s->d_Imat, sizeof(double) * s->NQ * s->NQ) != hipSuccess ||
            hipMemcpy(s->d_Imat, d->Imat, sizeof(double) * s->NQ * s->NQ, hipMemcpyHostToDevice) !=
                hipSuccess)
            return s->fail(CMDG_ERR_HIP, "ocean coupling: Imat upload failed");
    }
    return CMDG_OK;
}

int initialize_states(EngineBase *s, const cmdg_ocean_coupling_desc *d)
{
    const int64_t n = (int64_t)s->nreal * 2 * s->Np;
    hipLaunchKernelGGL(k_fill_columns, dim3(nblocks(n)), dim3(256), 0, s->s_comp, s->aux, s->naux,
                       d->slow_dGu_col, 2, -0.0, s->Np, (int64_t)s->nreal);
    return CMDG_OK;
}

int slow_to_fast(EngineBase *s, EngineBase *f, const cmdg_ocean_coupling_desc *d, const double *dQ)
{
    const int Nij = s->NQ * s->NQ, nv = d->nvertelem, Nqk2 = f->Np / Nij;
    const int64_t nh = s->nreal / nv;
    if (int r = s->integrate_velocity(dQ, s->ns, d->slow_u_col, d->nvertelem)) return r;
    if (int r = order(s, f->s_comp, s->s_comp)) return r;
    hipLaunchKernelGGL(k_top_to_layer, dim3(nblocks(nh * f->Np)), dim3(256), 0, s->s_comp, f->aux,
                       f->naux, d->fast_GU_col, (const double *)s->d_flowint, Nij, s->NQ, nv, Nqk2, nh);
    hipLaunchKernelGGL(k_column_minus_top_over_H, dim3(nblocks((int64_t)s->nreal * s->Np)), dim3(256),
                       0, s->s_comp, s->aux, s->naux, d->slow_dGu_col, (const double *)s->aux, s->naux,
                       d->slow_dGu_col, (const double *)s->d_flowint, d->H, Nij, s->NQ, nv,
                       (int64_t)0, nh);
    return order(s, s->s_comp, f->s_comp);
}

int fast_to_slow(EngineBase *s, EngineBase *f, const cmdg_ocean_coupling_desc *d, double *Q3,
                 const double *Q2)
{
    if (int r = s->integrate_velocity(Q3, s->ns, d->slow_u_col, d->nvertelem)) return r;
    const int Nij = s->NQ * s->NQ, nv = d->nvertelem, Nqk2 = f->Np / Nij;
    const int64_t nh = s->nreal / nv;
    if (int r = order(s, f->s_comp, s->s_comp)) return r;
    hipLaunchKernelGGL(k_reconcile_layer, dim3(nblocks(nh * f->Np)), dim3(256), 0, s->s_comp, f->aux,
                       f->naux, d->fast_du_col, Q2, f->ns, d->fast_U_col,
                       (const double *)s->d_flowint, d->H, Nij, s->NQ, nv, Nqk2, nh);
    hipLaunchKernelGGL(k_reconcile_column, dim3(nblocks((int64_t)s->nreal * s->Np)), dim3(256), 0,
                       s->s_comp, Q3, s->ns, d->slow_u_col, d->slow_eta_col, Q2, f->ns, d->fast_U_col,
                       d->fast_eta_col, (const double *)s->d_flowint, d->H, Nij, s->NQ, nv, Nqk2, nh);
    return order(s, s->s_comp, f->s_comp);
}

int lsrk_update(EngineBase *e, double *dQ, double *Q, double rka_next, double rkb_dt)
{
    const int64_t n = (int64_t)e->Np * e->ns * e->nreal;
    hipLaunchKernelGGL(k_lsrk_update, dim3(nblocks(n)), dim3(256), 0, e->s_comp, dQ, Q, rka_next,
                       rkb_dt, n);
    return CMDG_OK;
}

// update! of StrongStabilityPreservingRungeKuttaMethod.jl:167-190
__global__ void k_ssprk_update(const double *__restrict__ R, const double *__restrict__ Q,
                               double *__restrict__ Qstage, double rka1, double rka2, double rkb,
                               double dt, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        Qstage[i] = rka1 * Q[i] + rka2 * Qstage[i] + dt * rkb * R[i];
}

// update! of LowStorageRungeKutta3NMethod.jl:201-226
__global__ void k_ls3n_update(double *__restrict__ dQ, double *__restrict__ dR, double *__restrict__ Q,
                              double rka1, double rka2, double rkb1, double rkb2, double dt,
                              int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        Q[i] += rkb1 * dt * dQ[i] + rkb2 * dt * dR[i];
        dR[i] += rka2 * dQ[i];
        dQ[i] *= rka1;
    }
}
__global__ void k_fill(double *__restrict__ a, double v, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        a[i] = v;
}

int launch_status(EngineBase *e)
{
    hipError_t r = hipGetLastError();
    if (r != hipSuccess) return e->fail(CMDG_ERR_HIP, std::string("split explicit launch: ") + hipGetErrorString(r));
    return CMDG_OK;
}

}  // namespace

extern "C" {

int cmdg_ocean_initialize_states(cmdg_handle slow, cmdg_handle fast, const cmdg_ocean_coupling_desc *d)
{
    if (int r = check(slow, fast, d)) return set_err2(slow, r);
    DevGuard guard_(slow->eng);
    initialize_states(slow->eng, d);
    return set_err2(slow, launch_status(slow->eng));
}

int cmdg_ocean_tendency_from_slow_to_fast(cmdg_handle slow, cmdg_handle fast,
                                          const cmdg_ocean_coupling_desc *d, const double *dQ_slow)
{
    if (int r = check(slow, fast, d)) return set_err2(slow, r);
    DevGuard guard_(slow->eng);
    if (!dQ_slow) return CMDG_ERR_INVALID;
    if (int r = slow_to_fast(slow->eng, fast->eng, d, dQ_slow)) return set_err2(slow, r);
    return set_err2(slow, launch_status(slow->eng));
}

int cmdg_ocean_reconcile_from_fast_to_slow(cmdg_handle slow, cmdg_handle fast,
                                           const cmdg_ocean_coupling_desc *d, double *Q_slow,
                                           const double *Q_fast)
{
    if (int r = check(slow, fast, d)) return set_err2(slow, r);
    DevGuard guard_(slow->eng);
    if (!Q_slow || !Q_fast) return CMDG_ERR_INVALID;
    if (int r = fast_to_slow(slow->eng, fast->eng, d, Q_slow, Q_fast)) return set_err2(slow, r);
    return set_err2(slow, launch_status(slow->eng));
}

int cmdg_lsrk_update(cmdg_handle h, double *dQ, double *Q, double rka_next, double rkb_dt)
{
    if (!h || !dQ || !Q) return CMDG_ERR_INVALID;
    DevGuard guard_(h->eng);
    lsrk_update(h->eng, dQ, Q, rka_next, rkb_dt);
    return set_err2(h, launch_status(h->eng));
}

int cmdg_ls3n_step(cmdg_handle h, double *Q, double *dQ, double *dR, double t, double dt,
                   int32_t nstages, const double *rka, const double *rkb, const double *rkc)
{
    if (!h || !Q || !dQ || !dR || !rka || !rkb || !rkc || nstages < 1) return CMDG_ERR_INVALID;
    DevGuard guard_(h->eng);
    EngineBase *e = h->eng;
    const int64_t n = (int64_t)e->Np * e->ns * e->nreal;
    hipLaunchKernelGGL(k_fill, dim3(nblocks(n)), dim3(256), 0, e->s_comp, dR, 0.0, n);  // `rv_dR .= -0`: integer -0, i.e. +0.0
    for (int s = 0; s < nstages; ++s) {
        RhsCtx c;
        c.tendency = dQ;
        c.Qin = Q;
        c.t = t + rkc[s] * dt;
        c.alpha = 1.0;
        c.beta = 1.0;  // increment = true
        if (int r = e->rhs_async(c)) return set_err2(h, r);
        const int sn = (s + 1) % nstages;
        hipLaunchKernelGGL(k_ls3n_update, dim3(nblocks(n)), dim3(256), 0, e->s_comp, dQ, dR, Q,
                           rka[2 * sn], rka[2 * sn + 1], rkb[2 * s], rkb[2 * s + 1], dt, n);
    }
    return set_err2(h, launch_status(e));
}

int cmdg_ssprk_step(cmdg_handle h, double *Q, double *Rstage, double *Qstage, double t, double dt,
                    int32_t nstages, const double *rka, const double *rkb, const double *rkc)
{
    if (!h || !Q || !Rstage || !Qstage || !rka || !rkb || !rkc || nstages < 1) return CMDG_ERR_INVALID;
    DevGuard guard_(h->eng);
    EngineBase *e = h->eng;
    const int64_t n = (int64_t)e->Np * e->ns * e->nreal;
    if (hipMemcpyAsync(Qstage, Q, sizeof(double) * n, hipMemcpyDeviceToDevice, e->s_comp) != hipSuccess)
        return set_err2(h, e->fail(CMDG_ERR_HIP, "ssprk: copy failed"));
    for (int s = 0; s < nstages; ++s) {
        RhsCtx c;
        c.tendency = Rstage;
        c.Qin = Qstage;
        c.t = t + rkc[s] * dt;
        c.alpha = 1.0;
        c.beta = 0.0;
        if (int r = e->rhs_async(c)) return set_err2(h, r);
        hipLaunchKernelGGL(k_ssprk_update, dim3(nblocks(n)), dim3(256), 0, e->s_comp,
                           (const double *)Rstage, (const double *)Q, Qstage, rka[2 * s],
                           rka[2 * s + 1], rkb[s], dt, n);
    }
    if (hipMemcpyAsync(Q, Qstage, sizeof(double) * n, hipMemcpyDeviceToDevice, e->s_comp) != hipSuccess)
        return set_err2(h, e->fail(CMDG_ERR_HIP, "ssprk: copy failed"));
    return set_err2(h, launch_status(e));
}

// dostep!(Qslow, split::SplitExplicitSolver, param, time) for n (slow, fast) pairs in lock step:
// one pair per rank.  With the RCCL transport a process drives its own pair (n = 1); handles
// connected by cmdg_comm_connect_local are driven together by one host thread.
static int group_split_explicit_step(int n, cmdg_handle *slow, cmdg_handle *fast,
                                     const cmdg_ocean_coupling_desc *d, int coupled, double **Q3,
                                     double **dQ3, double **dQ2fast, double **Q2, double **dQ2,
                                     double t, double dt, double dt_fast, int nstages,
                                     const double *rka, const double *rkb, const double *rkc)
{
    std::vector<EngineBase *> S(n), F(n);
    if (!slow[0]) return CMDG_ERR_INVALID;
    DevGuard guard_(slow[0]->eng);
    for (int i = 0; i < n; ++i) {
        if (int r = check(slow[i], fast[i], d)) return set_err2(slow[i], r);
        if (!Q3[i] || !dQ3[i] || !dQ2fast[i] || !Q2[i] || !dQ2[i]) return CMDG_ERR_INVALID;
        S[i] = slow[i]->eng;
        F[i] = fast[i]->eng;
    }
    auto bail = [&](std::vector<EngineBase *> &E, int r) {
        for (int i = 0; i < n; ++i)
            if (!E[i]->err.empty()) {
                S[0]->err = E[i]->err;
                break;
            }
        return set_err2(slow[0], r);
    };
    std::vector<RhsCtx> c(n);
    for (int st = 0; st < nstages; ++st) {
        const double ts = t + rkc[st] * dt;
        for (int i = 0; i < n; ++i) {
            if (coupled) initialize_states(S[i], d);
            c[i] = RhsCtx();
            c[i].Qin = Q3[i];
            c[i].t = ts;
            c[i].alpha = 1.0;
            c[i].tendency = dQ2fast[i];  // slow.rhs!(dQ2fast, Qslow, ...; increment = false)
            c[i].beta = 0.0;
        }
        if (int r = group_rhs(S, c)) return bail(S, r);
        for (int i = 0; i < n; ++i) {
            if (coupled)
                if (int r = slow_to_fast(S[i], F[i], d, dQ2fast[i])) return bail(S, r);
            c[i].tendency = dQ3[i];  // slow.rhs!(dQslow, Qslow, ...; increment = true)
            c[i].beta = 1.0;
        }
        if (int r = group_rhs(S, c)) return bail(S, r);
        // fractional time for the fast sub-steps of this stage
        const double gamma = st == nstages - 1 ? 1 - rkc[st] : rkc[st + 1] - rkc[st];
        const int nsub = dt_fast > 0 ? (int)std::ceil(gamma * dt / dt_fast) : 1;
        const double fdt = gamma * dt / nsub;
        for (int sub = 0; sub < nsub; ++sub)
            if (int r = group_lsrk_step(F, Q2, dQ2, ts + sub * fdt, fdt, nstages, rka, rkb, rkc))
                return bail(F, r);
        for (int i = 0; i < n; ++i) {
            lsrk_update(S[i], dQ3[i], Q3[i], rka[(st + 1) % nstages], rkb[st] * dt);
            if (coupled)
                if (int r = fast_to_slow(S[i], F[i], d, Q3[i], Q2[i])) return bail(S, r);
        }
    }
    return set_err2(slow[0], launch_status(S[0]));
}

int cmdg_split_explicit_step(cmdg_handle slow, cmdg_handle fast, const cmdg_ocean_coupling_desc *d,
                             int32_t coupled, double *Q3, double *dQ3, double *dQ2fast, double *Q2,
                             double *dQ2, double t, double dt, double dt_fast, int32_t nstages,
                             const double *rka, const double *rkb, const double *rkc)
{
    if (!slow || !fast || !d || !rka || !rkb || !rkc || nstages < 1) return CMDG_ERR_INVALID;
    if (slow->eng->transport == TRANSPORT_LOCAL && slow->eng->communicate())
        return set_err2(slow, slow->eng->fail(CMDG_ERR_INVALID,
                                              "handles connected locally must be driven by "
                                              "cmdg_group_split_explicit_step"));
    return group_split_explicit_step(1, &slow, &fast, d, coupled, &Q3, &dQ3, &dQ2fast, &Q2, &dQ2, t,
                                     dt, dt_fast, nstages, rka, rkb, rkc);
}

int cmdg_group_split_explicit_step(cmdg_handle *slow, cmdg_handle *fast, int32_t n,
                                   const cmdg_ocean_coupling_desc *d, int32_t coupled, double **Q3,
                                   double **dQ3, double **dQ2fast, double **Q2, double **dQ2,
                                   double t, double dt, double dt_fast, int32_t nstages,
                                   const double *rka, const double *rkb, const double *rkc)
{
    if (!slow || !fast || n < 1 || !d || !Q3 || !dQ3 || !dQ2fast || !Q2 || !dQ2 || !rka || !rkb ||
        !rkc || nstages < 1)
        return CMDG_ERR_INVALID;
    for (int i = 0; i < n; ++i)
        if (!slow[i] || !fast[i]) return CMDG_ERR_INVALID;
    return group_split_explicit_step(n, slow, fast, d, coupled, Q3, dQ3, dQ2fast, Q2, dQ2, t, dt,
                                     dt_fast, nstages, rka, rkb, rkc);
}

}  // extern "C"
