// The older split-explicit ocean stepper of the reference (src/Ocean/SplitExplicit01):
// dostep! of SplitExplicitLSRK2nMethod.jl:81-190 and the exchange functions of
// src/Ocean/SplitExplicit01/Communication.jl over two engines -- slow: OceanModel (3-D, with the
// recorded update_auxiliary_state! composition), fast: BarotropicModel on the one-layer
// extrusion of the 2-D grid.  Slow work is enqueued on the slow engine's stream, barotropic
// sub-steps on the fast engine's; events order the two around each exchange and the host never
// waits: the second slow evaluation + update! of a stage overlap that stage's sub-steps.
#include <cmath>
#include <vector>

#include "columns.h"
#include "engine.h"
#include "filters.h"
#include "physics_ocean01.h"

using namespace cmdg;

namespace {

unsigned nblocks01(int64_t n) { return (unsigned)std::min<int64_t>((n + 255) / 256, 65535); }

int order01(EngineBase *e, hipStream_t earlier, hipStream_t later)
{
    if (earlier == later) return CMDG_OK;
    if (ev_record(e->ev_comp, earlier) != hipSuccess ||
        hipStreamWaitEvent(later, e->ev_comp, 0) != hipSuccess)
        return e->fail(CMDG_ERR_HIP, "split explicit 01: stream ordering failed");
    return CMDG_OK;
}

// dst[:, dcol .. dcol + ncol - 1, e] (op)= src[:, scol .., e]   op: 0 copy, 1 add
__global__ void k01_cols(double *__restrict__ dst, int ndst, int dcol, const double *__restrict__ src,
                         int nsrc, int scol, int ncol, int op, int Np, int64_t nelems)
{
    const int64_t n = nelems * ncol * Np;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / ((int64_t)ncol * Np);
        const int r = (int)(i % ((int64_t)ncol * Np));
        const double v = src[r + (int64_t)Np * (scol + (int64_t)nsrc * e)];
        double &d = dst[r + (int64_t)Np * (dcol + (int64_t)ndst * e)];
        d = op ? d + v : v;
    }
}
__global__ void k01_scale(double *__restrict__ A, int nA, int col, int ncol, double f, int Np,
                          int64_t nelems)
{
    const int64_t n = nelems * ncol * Np;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / ((int64_t)ncol * Np);
        const int r = (int)(i % ((int64_t)ncol * Np));
        A[r + (int64_t)Np * (col + (int64_t)nA * e)] *= f;
    }
}

// reconcile_from_fast_to_slow!, 2-D part (Communication.jl:254-300, 308-322):
// Delta_u = (U_c - int u) / H; at the last stage eta_diag = eta of the slow model at the
// surface, Delta_eta = eta_c - eta_diag
__global__ void k01_reconcile_layer(double *__restrict__ A2, const double *__restrict__ Q3,
                                    const double *__restrict__ ia, double H, int last, int Nij,
                                    int Nqk3, int nvert, int Nqk2, int64_t nhorz)
{
    using B = BarotropicSE01;
    const int Np3 = Nij * Nqk3, Np2 = Nij * Nqk2;
    const int64_t n = nhorz * Np2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ijk = (int)(i % Np2);
        const int64_t eh = i / Np2;
        const int64_t et = (nvert - 1) + eh * nvert;
        const int top = ijk % Nij + Nij * (Nqk3 - 1);
        double *a = A2 + ijk + (int64_t)Np2 * B::NAUX * eh;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            double du = a[(int64_t)Np2 * (B::AUC + c)];
            du -= ia[top + (int64_t)Np3 * (c + 2 * et)];
            du /= H;
            a[(int64_t)Np2 * (B::ADU + c)] = du;
        }
        if (last) {
            const double ed = Q3[top + (int64_t)Np3 * (OceanSE01::ETA + (int64_t)OceanSE01::NS * et)];
            a[(int64_t)Np2 * B::AETAD] = ed;
            a[(int64_t)Np2 * B::ADETA] = a[(int64_t)Np2 * B::AETAC] - ed;
        }
    }
}
// ... 3-D part: u += Delta_u through the column; at the last stage eta = eta_c
__global__ void k01_reconcile_column(double *__restrict__ Q3, const double *__restrict__ A2, int last,
                                     int Nij, int Nqk3, int nvert, int Nqk2, int64_t nhorz)
{
    using B = BarotropicSE01;
    const int Np3 = Nij * Nqk3, Np2 = Nij * Nqk2;
    const int64_t n = nhorz * nvert * Np3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ijk = (int)(i % Np3);
        const int64_t e = i / Np3;
        const int64_t eh = e / nvert;
        const int ij = ijk % Nij;
        const double *a = A2 + ij + (int64_t)Np2 * B::NAUX * eh;
        double *q = Q3 + ijk + (int64_t)Np3 * OceanSE01::NS * e;
#pragma unroll
        for (int c = 0; c < 2; ++c) q[(int64_t)Np3 * (OceanSE01::U + c)] += a[(int64_t)Np2 * (B::ADU + c)];
        if (last) q[(int64_t)Np3 * OceanSE01::ETA] = a[(int64_t)Np2 * B::AETAC];
    }
}

int launch_status01(EngineBase *e)
{
    hipError_t r = hipGetLastError();
    if (r != hipSuccess) return e->fail(CMDG_ERR_HIP, std::string("split explicit 01 launch: ") + hipGetErrorString(r));
    return CMDG_OK;
}

}  // namespace

extern "C" int cmdg_split_explicit01_step(cmdg_handle slow, cmdg_handle fast, const cmdg_ocean01_desc *d,
                                          double *Q3, double *dQ3, double *dQ2fast, double *Q2,
                                          double *dQ2, double t, double dt, double dt_fast,
                                          int32_t nstages, const double *rka, const double *rkb,
                                          const double *rkc)
{
    if (!slow || !fast || !d || !Q3 || !dQ3 || !dQ2fast || !Q2 || !dQ2 || !rka || !rkb || !rkc ||
        nstages < 1)
        return CMDG_ERR_INVALID;
    DevGuard guard_(slow->eng);
    EngineBase *S = slow->eng, *F = fast->eng;
    using B = BarotropicSE01;
    using O = OceanSE01;
    auto bad = [&](const char *msg) {
        S->fail(CMDG_ERR_INVALID, msg);
        slow->err = S->err;
        return CMDG_ERR_INVALID;
    };
    if (S->ns != O::NS || S->naux != O::NAUX || S->ngf != O::NGF || F->ns != B::NS || F->naux != B::NAUX)
        return bad("split explicit 01: handles are not the OceanModel / BarotropicModel pair");
    if (!S->stacked || d->nvertelem < 1 || S->nreal % d->nvertelem || F->nreal != S->nreal / d->nvertelem)
        return bad("split explicit 01: the fast grid holds one element per stack of the slow grid");
    if (F->Np % (S->NQ * S->NQ) || S->Np != S->NQ * S->NQ * S->NQ || !(d->H > 0) || d->add_fast_substeps < 0)
        return bad("split explicit 01: grids / parameters");
    if (S->communicate() || F->communicate())
        return bad("split explicit 01: single rank only");
    if (F->dev != S->dev)  // the fast launches and the events between the two engines assume one device
        return bad("split explicit 01: the slow and the fast handle live on one device");
    // dostep!(Qfast, fast, ...) runs the FAST solver's scheme (SplitExplicitLSRK2nMethod.jl:150-165):
    // its own tableau when the caller gives one, the slow solver's otherwise
    const int nst_f = d->nstages_fast > 0 ? d->nstages_fast : nstages;
    const double *rka_f = d->nstages_fast > 0 ? d->rka_fast : rka;
    const double *rkb_f = d->nstages_fast > 0 ? d->rkb_fast : rkb;
    const double *rkc_f = d->nstages_fast > 0 ? d->rkc_fast : rkc;
    if (d->nstages_fast < 0 || (d->nstages_fast > 0 && (!rka_f || !rkb_f || !rkc_f)))
        return bad("split explicit 01: the fast solver's tableau");
    if (!S->d_Imat) {
        if (!d->Imat) return bad("split explicit 01: Imat is NULL");
        if (hipMalloc(&S->d_Imat, sizeof(double) * S->NQ * S->NQ) != hipSuccess ||
            hipMemcpy(S->d_Imat, d->Imat, sizeof(double) * S->NQ * S->NQ, hipMemcpyHostToDevice) != hipSuccess)
            return bad("split explicit 01: Imat upload failed");
    }
    const int Nij = S->NQ * S->NQ, nv = d->nvertelem, Nqk2 = F->Np / Nij, Np2 = F->Np, Np3 = S->Np;
    const int64_t nh = F->nreal, n3 = S->nreal;
    auto fcols = [&](double *dst, int ndst, int dcol, const double *src, int nsrc, int scol, int ncol,
                     int op, hipStream_t st) {
        hipLaunchKernelGGL(k01_cols, dim3(nblocks01(nh * ncol * Np2)), dim3(256), 0, st, dst, ndst, dcol,
                           src, nsrc, scol, ncol, op, Np2, nh);
    };
    auto fail_from = [&](EngineBase *e, int r) {
        slow->err = e->err;
        return r;
    };
    std::vector<EngineBase *> Fv{F};
    double *Q2v[1] = {Q2}, *dQ2v[1] = {dQ2};
    for (int s = 0; s < nstages; ++s) {
        const bool first = s == 0, last = s == nstages - 1;
        const double stage_time = t + rkc[s] * dt;
        const double fract_dt = last ? (1 - rkc[s]) * dt : (rkc[s + 1] - rkc[s]) * dt;
        // ---- initialize_fast_state! (Communication.jl:103-149)
        const int add = d->add_fast_substeps;
        int fs1, fs2, fs3;
        if (add == 0) {
            const int steps = dt_fast > 0 ? (int)std::ceil(fract_dt / dt_fast) : 1;
            fs1 = fs2 = fs3 = steps;
        } else {
            const int steps = dt_fast > 0 ? (int)std::ceil(fract_dt / dt_fast / add) : 1;
            fs2 = add * steps;
            fs1 = (add - 1) * steps;
            fs3 = (add + 1) * steps;
        }
        const double fdt = fract_dt / fs2;
        double count = 0.0;
        hipLaunchKernelGGL(k_fill_columns, dim3(nblocks01(nh * 3 * Np2)), dim3(256), 0, F->s_comp, F->aux,
                           B::NAUX, (int)B::AUC, 3, -0.0, Np2, nh);  // U_c, eta_c (adjacent columns)
        if (!first) {  // set fast-state to previously stored value
            fcols(Q2, B::NS, B::ETA, F->aux, B::NAUX, B::AETAS, 1, 0, F->s_comp);
            fcols(Q2, B::NS, B::U1, F->aux, B::NAUX, B::AUS, 2, 0, F->s_comp);
        }
        // ---- initialize_adjustment!: dG_u = 0
        hipLaunchKernelGGL(k_fill_columns, dim3(nblocks01(n3 * 2 * Np3)), dim3(256), 0, S->s_comp, S->aux,
                           O::NAUX, (int)O::ADGU, 2, 0.0, Np3, n3);
        // ---- slow.rhs!(dQ2fast, Qslow, ...; increment = false)
        RhsCtx c;
        c.Qin = Q3;
        c.t = stage_time;
        c.alpha = 1.0;
        c.tendency = dQ2fast;
        c.beta = 0.0;
        if (int r = S->rhs_async(c)) return fail_from(S, r);
        // ---- tendency_from_slow_to_fast! (Communication.jl:166-224)
        if (int r = S->integrate_velocity(dQ2fast, O::NS, O::U, nv)) return fail_from(S, r);
        if (int r = order01(S, F->s_comp, S->s_comp)) return fail_from(S, r);
        hipLaunchKernelGGL(k_top_to_layer, dim3(nblocks01(nh * Np2)), dim3(256), 0, S->s_comp, F->aux,
                           B::NAUX, (int)B::AGU, (const double *)S->d_flowint, Nij, S->NQ, nv, Nqk2, nh);
        hipLaunchKernelGGL(k_column_minus_top_over_H, dim3(nblocks01(n3 * Np3)), dim3(256), 0, S->s_comp,
                           S->aux, O::NAUX, (int)O::ADGU, (const double *)S->aux, O::NAUX, (int)O::ADGU,
                           (const double *)S->d_flowint, d->H, Nij, S->NQ, nv, (int64_t)0, n3 / nv);
        if (int r = order01(S, S->s_comp, F->s_comp)) return fail_from(S, r);
        // ---- slow.rhs!(dQslow, Qslow, ...; increment = true) and update!
        c.tendency = dQ3;
        c.beta = 1.0;
        if (int r = S->rhs_async(c)) return fail_from(S, r);
        {
            const int64_t n = (int64_t)Np3 * O::NS * n3;
            hipLaunchKernelGGL(k_lsrk_update, dim3(nblocks01(n)), dim3(256), 0, S->s_comp, dQ3, Q3,
                               rka[(s + 1) % nstages], rkb[s] * dt, n);
        }
        // ---- barotropic sub-steps with cummulate_fast_solution! (Communication.jl:226-252)
        for (int sub = 1; sub <= fs3; ++sub) {
            const double fast_time = stage_time + (sub - 1) * fdt;
            if (int r = group_lsrk_step(Fv, Q2v, dQ2v, fast_time, fdt, nst_f, rka_f, rkb_f, rkc_f))
                return fail_from(F, r);
            if (sub >= fs1) {
                fcols(F->aux, B::NAUX, B::AUC, Q2, B::NS, B::U1, 2, 1, F->s_comp);
                fcols(F->aux, B::NAUX, B::AETAC, Q2, B::NS, B::ETA, 1, 1, F->s_comp);
                count += 1.0;
            }
            if (sub == fs2) {
                fcols(F->aux, B::NAUX, B::AUS, Q2, B::NS, B::U1, 2, 0, F->s_comp);
                fcols(F->aux, B::NAUX, B::AETAS, Q2, B::NS, B::ETA, 1, 0, F->s_comp);
            }
        }
        // ---- reconcile_from_fast_to_slow! (Communication.jl:254-336)
        hipLaunchKernelGGL(k01_scale, dim3(nblocks01(nh * 3 * Np2)), dim3(256), 0, F->s_comp, F->aux,
                           B::NAUX, (int)B::AUC, 3, 1 / count, Np2, nh);
        if (int r = S->integrate_velocity(Q3, O::NS, O::U, nv)) return fail_from(S, r);
        if (int r = order01(S, F->s_comp, S->s_comp)) return fail_from(S, r);
        hipLaunchKernelGGL(k01_reconcile_layer, dim3(nblocks01(nh * Np2)), dim3(256), 0, S->s_comp, F->aux,
                           (const double *)Q3, (const double *)S->d_flowint, d->H, (int)last, Nij, S->NQ,
                           nv, Nqk2, nh);
        hipLaunchKernelGGL(k01_reconcile_column, dim3(nblocks01(n3 * Np3)), dim3(256), 0, S->s_comp, Q3,
                           (const double *)F->aux, (int)last, Nij, S->NQ, nv, Nqk2, nh);
        if (last) {  // reset fast-state to end of time-step value
            fcols(Q2, B::NS, B::ETA, F->aux, B::NAUX, B::AETAS, 1, 0, S->s_comp);
            fcols(Q2, B::NS, B::U1, F->aux, B::NAUX, B::AUS, 2, 0, S->s_comp);
        }
        if (int r = order01(S, S->s_comp, F->s_comp)) return fail_from(S, r);
    }
    int r = launch_status01(S);
    if (r) slow->err = S->err;
    return r;
}
