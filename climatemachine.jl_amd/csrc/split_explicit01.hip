// The older split-explicit ocean stepper of the reference (src/Ocean/SplitExplicit01):
// dostep! of SplitExplicitLSRK2nMethod.jl:81-190 and the exchange functions of
// src/Ocean/SplitExplicit01/Communication.jl over two engines -- slow: OceanModel (3-D, with the
// recorded update_auxiliary_state! composition), fast: BarotropicModel on the one-layer
// extrusion of the 2-D grid.  Slow work is enqueued on the slow engine's stream, barotropic
// sub-steps on the fast engine's; events order the two around each exchange and the host never
// waits: the second slow evaluation + update! of a stage overlap that stage's sub-steps.
// Partitioned runs: one (slow, fast) pair per rank, see group_split_explicit01_step.
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

// dostep!(Qslow, ::SplitExplicitLSRK2nSolver, param, time) for n (slow, fast) pairs in lock step: one
// pair per rank.  With the RCCL transport a process drives its own pair (n = 1); handles connected
// by cmdg_comm_connect_local (slow models, fast models and the nested continuity operators, each
// among themselves) are driven together by one host thread.  Every exchange function acts on a
// rank's real columns; the ghost stacks' flow deviation and kinematic pressure come from the
// operator itself (EngineBase::rhs_segment, segment 1).
static int group_split_explicit01_step(int n, cmdg_handle *slow, cmdg_handle *fast, const cmdg_ocean01_desc *d,
                                       double **Q3, double **dQ3, double **dQ2fast, double **Q2,
                                       double **dQ2, double t, double dt, double dt_fast,
                                       int32_t nstages, const double *rka, const double *rkb,
                                       const double *rkc)
{
    using B = BarotropicSE01;
    using O = OceanSE01;
    std::vector<EngineBase *> S(n), F(n);
    if (!slow[0]) return CMDG_ERR_INVALID;
    DevGuard guard_(slow[0]->eng);
    auto bad = [&](const char *msg) {
        slow[0]->eng->fail(CMDG_ERR_INVALID, msg);
        slow[0]->err = slow[0]->eng->err;
        return CMDG_ERR_INVALID;
    };
    for (int i = 0; i < n; ++i) {
        if (!slow[i] || !fast[i] || !Q3[i] || !dQ3[i] || !dQ2fast[i] || !Q2[i] || !dQ2[i]) return CMDG_ERR_INVALID;
        S[i] = slow[i]->eng;
        F[i] = fast[i]->eng;
        if (S[i]->ns != O::NS || S[i]->naux != O::NAUX || S[i]->ngf != O::NGF || F[i]->ns != B::NS ||
            F[i]->naux != B::NAUX)
            return bad("split explicit 01: handles are not the OceanModel / BarotropicModel pair");
        if (!S[i]->stacked || d->nvertelem < 1 || S[i]->nreal % d->nvertelem ||
            F[i]->nreal != S[i]->nreal / d->nvertelem)
            return bad("split explicit 01: the fast grid holds one element per stack of the slow grid");
        if (F[i]->Np % (S[i]->NQ * S[i]->NQ) || S[i]->Np != S[i]->NQ * S[i]->NQ * S[i]->NQ || !(d->H > 0) ||
            d->add_fast_substeps < 0)
            return bad("split explicit 01: grids / parameters");
        if (F[i]->dev != S[i]->dev)  // the fast launches and the events between the two engines assume one device
            return bad("split explicit 01: the slow and the fast handle live on one device");
        if (n > 1 && (S[i]->transport != TRANSPORT_LOCAL || F[i]->transport != TRANSPORT_LOCAL))
            return bad("split explicit 01: several pairs in one call need the local transport");
        if (!S[i]->d_Imat) {
            if (!d->Imat) return bad("split explicit 01: Imat is NULL");
            if (hipMalloc(&S[i]->d_Imat, sizeof(double) * S[i]->NQ * S[i]->NQ) != hipSuccess ||
                hipMemcpy(S[i]->d_Imat, d->Imat, sizeof(double) * S[i]->NQ * S[i]->NQ, hipMemcpyHostToDevice) !=
                    hipSuccess)
                return bad("split explicit 01: Imat upload failed");
        }
    }
    // dostep!(Qfast, fast, ...) runs the FAST solver's scheme (SplitExplicitLSRK2nMethod.jl:150-165):
    // its own tableau when the caller gives one, the slow solver's otherwise
    const int nst_f = d->nstages_fast > 0 ? d->nstages_fast : nstages;
    const double *rka_f = d->nstages_fast > 0 ? d->rka_fast : rka;
    const double *rkb_f = d->nstages_fast > 0 ? d->rkb_fast : rkb;
    const double *rkc_f = d->nstages_fast > 0 ? d->rkc_fast : rkc;
    if (d->nstages_fast < 0 || (d->nstages_fast > 0 && (!rka_f || !rkb_f || !rkc_f)))
        return bad("split explicit 01: the fast solver's tableau");
    const int Nij = S[0]->NQ * S[0]->NQ, nv = d->nvertelem, Nqk2 = F[0]->Np / Nij, Np2 = F[0]->Np, Np3 = S[0]->Np;
    auto fcols = [&](int i, double *dst, int ndst, int dcol, const double *src, int nsrc, int scol, int ncol,
                     int op, hipStream_t st) {
        const int64_t nh = F[i]->nreal;
        if (nh == 0) return;
        hipLaunchKernelGGL(k01_cols, dim3(nblocks01(nh * ncol * Np2)), dim3(256), 0, st, dst, ndst, dcol,
                           src, nsrc, scol, ncol, op, Np2, nh);
    };
    auto fail_from = [&](std::vector<EngineBase *> &E, int r) {
        for (int i = 0; i < n; ++i)
            if (!E[i]->err.empty()) {
                slow[0]->err = E[i]->err;
                break;
            }
        return r;
    };
    std::vector<RhsCtx> c(n);
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
        for (int i = 0; i < n; ++i) {
            const int64_t nh = F[i]->nreal, n3 = S[i]->nreal;
            if (nh > 0)
                hipLaunchKernelGGL(k_fill_columns, dim3(nblocks01(nh * 3 * Np2)), dim3(256), 0, F[i]->s_comp,
                                   F[i]->aux, B::NAUX, (int)B::AUC, 3, -0.0, Np2, nh);  // U_c, eta_c (adjacent columns)
            if (!first) {  // set fast-state to previously stored value
                fcols(i, Q2[i], B::NS, B::ETA, F[i]->aux, B::NAUX, B::AETAS, 1, 0, F[i]->s_comp);
                fcols(i, Q2[i], B::NS, B::U1, F[i]->aux, B::NAUX, B::AUS, 2, 0, F[i]->s_comp);
            }
            // ---- initialize_adjustment!: dG_u = 0
            if (n3 > 0)
                hipLaunchKernelGGL(k_fill_columns, dim3(nblocks01(n3 * 2 * Np3)), dim3(256), 0, S[i]->s_comp,
                                   S[i]->aux, O::NAUX, (int)O::ADGU, 2, 0.0, Np3, n3);
            // ---- slow.rhs!(dQ2fast, Qslow, ...; increment = false)
            c[i] = RhsCtx();
            c[i].Qin = Q3[i];
            c[i].t = stage_time;
            c[i].alpha = 1.0;
            c[i].tendency = dQ2fast[i];
            c[i].beta = 0.0;
        }
        if (int r = group_rhs(S, c)) return fail_from(S, r);
        for (int i = 0; i < n; ++i) {
            const int64_t nh = F[i]->nreal, n3 = S[i]->nreal;
            // ---- tendency_from_slow_to_fast! (Communication.jl:166-224)
            if (int r = S[i]->integrate_velocity(dQ2fast[i], O::NS, O::U, nv)) return fail_from(S, r);
            if (int r = order01(S[i], F[i]->s_comp, S[i]->s_comp)) return fail_from(S, r);
            if (nh > 0) {
                hipLaunchKernelGGL(k_top_to_layer, dim3(nblocks01(nh * Np2)), dim3(256), 0, S[i]->s_comp,
                                   F[i]->aux, B::NAUX, (int)B::AGU, (const double *)S[i]->d_flowint, Nij,
                                   S[i]->NQ, nv, Nqk2, nh);
                hipLaunchKernelGGL(k_column_minus_top_over_H, dim3(nblocks01(n3 * Np3)), dim3(256), 0,
                                   S[i]->s_comp, S[i]->aux, O::NAUX, (int)O::ADGU, (const double *)S[i]->aux,
                                   O::NAUX, (int)O::ADGU, (const double *)S[i]->d_flowint, d->H, Nij, S[i]->NQ,
                                   nv, (int64_t)0, n3 / nv);
            }
            if (int r = order01(S[i], S[i]->s_comp, F[i]->s_comp)) return fail_from(S, r);
            // ---- slow.rhs!(dQslow, Qslow, ...; increment = true)
            c[i].tendency = dQ3[i];
            c[i].beta = 1.0;
        }
        if (int r = group_rhs(S, c)) return fail_from(S, r);
        for (int i = 0; i < n; ++i) {  // ... and update!
            const int64_t nn = (int64_t)Np3 * O::NS * S[i]->nreal;
            if (nn > 0)
                hipLaunchKernelGGL(k_lsrk_update, dim3(nblocks01(nn)), dim3(256), 0, S[i]->s_comp, dQ3[i], Q3[i],
                                   rka[(s + 1) % nstages], rkb[s] * dt, nn);
        }
        // ---- barotropic sub-steps with cummulate_fast_solution! (Communication.jl:226-252)
        for (int sub = 1; sub <= fs3; ++sub) {
            const double fast_time = stage_time + (sub - 1) * fdt;
            if (int r = group_lsrk_step(F, Q2, dQ2, fast_time, fdt, nst_f, rka_f, rkb_f, rkc_f))
                return fail_from(F, r);
            for (int i = 0; i < n; ++i) {
                if (sub >= fs1) {
                    fcols(i, F[i]->aux, B::NAUX, B::AUC, Q2[i], B::NS, B::U1, 2, 1, F[i]->s_comp);
                    fcols(i, F[i]->aux, B::NAUX, B::AETAC, Q2[i], B::NS, B::ETA, 1, 1, F[i]->s_comp);
                }
                if (sub == fs2) {
                    fcols(i, F[i]->aux, B::NAUX, B::AUS, Q2[i], B::NS, B::U1, 2, 0, F[i]->s_comp);
                    fcols(i, F[i]->aux, B::NAUX, B::AETAS, Q2[i], B::NS, B::ETA, 1, 0, F[i]->s_comp);
                }
            }
            if (sub >= fs1) count += 1.0;
        }
        // ---- reconcile_from_fast_to_slow! (Communication.jl:254-336)
        for (int i = 0; i < n; ++i) {
            const int64_t nh = F[i]->nreal, n3 = S[i]->nreal;
            if (nh > 0)
                hipLaunchKernelGGL(k01_scale, dim3(nblocks01(nh * 3 * Np2)), dim3(256), 0, F[i]->s_comp, F[i]->aux,
                                   B::NAUX, (int)B::AUC, 3, 1 / count, Np2, nh);
            if (int r = S[i]->integrate_velocity(Q3[i], O::NS, O::U, nv)) return fail_from(S, r);
            if (int r = order01(S[i], F[i]->s_comp, S[i]->s_comp)) return fail_from(S, r);
            if (nh > 0) {
                hipLaunchKernelGGL(k01_reconcile_layer, dim3(nblocks01(nh * Np2)), dim3(256), 0, S[i]->s_comp,
                                   F[i]->aux, (const double *)Q3[i], (const double *)S[i]->d_flowint, d->H,
                                   (int)last, Nij, S[i]->NQ, nv, Nqk2, nh);
                hipLaunchKernelGGL(k01_reconcile_column, dim3(nblocks01(n3 * Np3)), dim3(256), 0, S[i]->s_comp,
                                   Q3[i], (const double *)F[i]->aux, (int)last, Nij, S[i]->NQ, nv, Nqk2, nh);
            }
            if (last) {  // reset fast-state to end of time-step value
                fcols(i, Q2[i], B::NS, B::ETA, F[i]->aux, B::NAUX, B::AETAS, 1, 0, S[i]->s_comp);
                fcols(i, Q2[i], B::NS, B::U1, F[i]->aux, B::NAUX, B::AUS, 2, 0, S[i]->s_comp);
            }
            if (int r = order01(S[i], S[i]->s_comp, F[i]->s_comp)) return fail_from(S, r);
        }
    }
    int r = launch_status01(S[0]);
    if (r) slow[0]->err = S[0]->err;
    return r;
}

extern "C" int cmdg_split_explicit01_step(cmdg_handle slow, cmdg_handle fast, const cmdg_ocean01_desc *d,
                                          double *Q3, double *dQ3, double *dQ2fast, double *Q2,
                                          double *dQ2, double t, double dt, double dt_fast,
                                          int32_t nstages, const double *rka, const double *rkb,
                                          const double *rkc)
{
    if (!slow || !fast || !d || !Q3 || !dQ3 || !dQ2fast || !Q2 || !dQ2 || !rka || !rkb || !rkc ||
        nstages < 1)
        return CMDG_ERR_INVALID;
    if (slow->eng->transport == TRANSPORT_LOCAL && slow->eng->communicate()) {
        slow->eng->fail(CMDG_ERR_INVALID, "handles connected locally must be driven by cmdg_group_split_explicit01_step");
        slow->err = slow->eng->err;
        return CMDG_ERR_INVALID;
    }
    return group_split_explicit01_step(1, &slow, &fast, d, &Q3, &dQ3, &dQ2fast, &Q2, &dQ2, t, dt, dt_fast,
                                       nstages, rka, rkb, rkc);
}

extern "C" int cmdg_group_split_explicit01_step(cmdg_handle *slow, cmdg_handle *fast, int32_t n,
                                                const cmdg_ocean01_desc *d, double **Q3, double **dQ3,
                                                double **dQ2fast, double **Q2, double **dQ2, double t,
                                                double dt, double dt_fast, int32_t nstages,
                                                const double *rka, const double *rkb, const double *rkc)
{
    if (!slow || !fast || n < 1 || !d || !Q3 || !dQ3 || !dQ2fast || !Q2 || !dQ2 || !rka || !rkb || !rkc ||
        nstages < 1)
        return CMDG_ERR_INVALID;
    for (int i = 0; i < n; ++i)
        if (!slow[i] || !fast[i]) return CMDG_ERR_INVALID;
    return group_split_explicit01_step(n, slow, fast, d, Q3, dQ3, dQ2fast, Q2, dQ2, t, dt, dt_fast, nstages,
                                       rka, rkb, rkc);
}
