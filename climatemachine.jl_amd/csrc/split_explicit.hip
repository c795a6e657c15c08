// Split-explicit barotropic / baroclinic ocean stepper: the exchange functions of
// src/Ocean/SplitExplicit/Communication.jl and dostep! of
// src/Numerics/ODESolvers/SplitExplicitMethod.jl:70-177 over two engines (slow 3-D
// HydrostaticBoussinesqModel, fast ShallowWaterModel on the one-layer extrusion of the 2-D grid).
// Everything is enqueued on the slow engine's compute stream except the fast model's own
// sub-steps; the two streams are ordered with events, the host never waits.
#include <cmath>

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
    if (hipEventRecord(e->ev_comp, earlier) != hipSuccess ||
        hipStreamWaitEvent(later, e->ev_comp, 0) != hipSuccess)
        return e->fail(CMDG_ERR_HIP, "split explicit: stream ordering failed");
    return CMDG_OK;
}

int check(cmdg_handle slow, cmdg_handle fast, const cmdg_ocean_coupling_desc *d)
{
    if (!slow || !fast || !d) return CMDG_ERR_INVALID;
    EngineBase *s = slow->eng, *f = fast->eng;
    // the flow deviation u_d of ghost columns would have to be integrated from the received
    // face pencils (as update_auxiliary_state! on ghost elements does); not built yet
    if (s->communicate() || f->communicate())
        return s->fail(CMDG_ERR_UNSUPPORTED,
                       "ocean coupling: partitioned (multi-rank) grids are not supported; run the "
                       "split-explicit ocean on one rank per column set without ghost elements");
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
    if (int r = s->integrate_velocity(dQ, s->ns, d->slow_u_col, d->nvertelem)) return r;
    const int Nij = s->NQ * s->NQ, nv = d->nvertelem, Nqk2 = f->Np / Nij;
    const int64_t nh = s->nreal / nv;
    if (int r = order(s, f->s_comp, s->s_comp)) return r;
    hipLaunchKernelGGL(k_top_to_layer, dim3(nblocks(nh * f->Np)), dim3(256), 0, s->s_comp, f->aux,
                       f->naux, d->fast_GU_col, (const double *)s->d_flowint, Nij, s->NQ, nv, Nqk2, nh);
    hipLaunchKernelGGL(k_column_minus_top_over_H, dim3(nblocks((int64_t)s->nreal * s->Np)), dim3(256),
                       0, s->s_comp, s->aux, s->naux, d->slow_dGu_col, (const double *)s->aux, s->naux,
                       d->slow_dGu_col, (const double *)s->d_flowint, d->H, Nij, s->NQ, nv, nh);
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
    initialize_states(slow->eng, d);
    return set_err2(slow, launch_status(slow->eng));
}

int cmdg_ocean_tendency_from_slow_to_fast(cmdg_handle slow, cmdg_handle fast,
                                          const cmdg_ocean_coupling_desc *d, const double *dQ_slow)
{
    if (int r = check(slow, fast, d)) return set_err2(slow, r);
    if (!dQ_slow) return CMDG_ERR_INVALID;
    if (int r = slow_to_fast(slow->eng, fast->eng, d, dQ_slow)) return set_err2(slow, r);
    return set_err2(slow, launch_status(slow->eng));
}

int cmdg_ocean_reconcile_from_fast_to_slow(cmdg_handle slow, cmdg_handle fast,
                                           const cmdg_ocean_coupling_desc *d, double *Q_slow,
                                           const double *Q_fast)
{
    if (int r = check(slow, fast, d)) return set_err2(slow, r);
    if (!Q_slow || !Q_fast) return CMDG_ERR_INVALID;
    if (int r = fast_to_slow(slow->eng, fast->eng, d, Q_slow, Q_fast)) return set_err2(slow, r);
    return set_err2(slow, launch_status(slow->eng));
}

int cmdg_lsrk_update(cmdg_handle h, double *dQ, double *Q, double rka_next, double rkb_dt)
{
    if (!h || !dQ || !Q) return CMDG_ERR_INVALID;
    lsrk_update(h->eng, dQ, Q, rka_next, rkb_dt);
    return set_err2(h, launch_status(h->eng));
}

int cmdg_ssprk_step(cmdg_handle h, double *Q, double *Rstage, double *Qstage, double t, double dt,
                    int32_t nstages, const double *rka, const double *rkb, const double *rkc)
{
    if (!h || !Q || !Rstage || !Qstage || !rka || !rkb || !rkc || nstages < 1) return CMDG_ERR_INVALID;
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

int cmdg_split_explicit_step(cmdg_handle slow, cmdg_handle fast, const cmdg_ocean_coupling_desc *d,
                             int32_t coupled, double *Q3, double *dQ3, double *dQ2fast, double *Q2,
                             double *dQ2, double t, double dt, double dt_fast, int32_t nstages,
                             const double *rka, const double *rkb, const double *rkc)
{
    if (int r = check(slow, fast, d)) return set_err2(slow, r);
    if (!Q3 || !dQ3 || !dQ2fast || !Q2 || !dQ2 || !rka || !rkb || !rkc || nstages < 1)
        return CMDG_ERR_INVALID;
    EngineBase *s = slow->eng, *f = fast->eng;
#define TRYS(x)                                  \
    do {                                         \
        if (int r_ = (x)) return set_err2(slow, r_); \
    } while (0)
#define TRYF(x)                                                 \
    do {                                                        \
        if (int r_ = (x)) {                                     \
            s->err = f->err;                                    \
            return set_err2(slow, r_);                          \
        }                                                       \
    } while (0)
    for (int st = 0; st < nstages; ++st) {
        const double ts = t + rkc[st] * dt;
        if (coupled) initialize_states(s, d);
        RhsCtx c;
        c.Qin = Q3;
        c.t = ts;
        c.alpha = 1.0;
        // slow.rhs!(dQ2fast, Qslow, param, slow_stage_time, increment = false)
        c.tendency = dQ2fast;
        c.beta = 0.0;
        TRYS(s->rhs_async(c));
        if (coupled) TRYS(slow_to_fast(s, f, d, dQ2fast));
        // slow.rhs!(dQslow, Qslow, param, slow_stage_time, increment = true)
        c.tendency = dQ3;
        c.beta = 1.0;
        TRYS(s->rhs_async(c));
        // fractional time for the fast sub-steps of this stage
        const double gamma = st == nstages - 1 ? 1 - rkc[st] : rkc[st + 1] - rkc[st];
        const int nsub = dt_fast > 0 ? (int)std::ceil(gamma * dt / dt_fast) : 1;
        const double fdt = gamma * dt / nsub;
        for (int sub = 0; sub < nsub; ++sub)
            TRYF(f->lsrk_step(Q2, dQ2, ts + sub * fdt, fdt, nstages, rka, rkb, rkc));
        lsrk_update(s, dQ3, Q3, rka[(st + 1) % nstages], rkb[st] * dt);
        if (coupled) TRYS(fast_to_slow(s, f, d, Q3, Q2));
    }
#undef TRYS
#undef TRYF
    return set_err2(slow, launch_status(s));
}

}  // extern "C"
