/* ORACLE -- TEST INFRASTRUCTURE ONLY (see dg_oracle.h).
 *
 * CPU restatement of the reference's element-filter kernels, thread-for-thread:
 *   orc_apply_filter       kernel_apply_filter!       src/Numerics/Mesh/Filters.jl:651-794
 *   orc_apply_tmar_filter  kernel_apply_TMAR_filter!  Filters.jl:796-884
 *   orc_apply_mp_filter    kernel_apply_mp_filter!    Filters.jl:893-1071
 * with the filter targets FilterIndices (Filters.jl:72-100), AtmosFilterPerturbations and
 * AtmosSpecificFilterPerturbations (src/Atmos/Model/filters.jl:4-118; dry model).
 *
 * Pinned by the reference's own filter tests (test/Numerics/Mesh/filter.jl): the hex
 * filter matrices :20-26,:51-56 (host side) and the analytic low/high-mode application
 * tests :169-330, TMAR :349-399 and the mass-conservation test :440-509.
 *
 * The work-group of the reference is one element; its threads are the (i,j,k) loops
 * here and every @synchronize is a loop boundary, so sums run in the same order.
 */
#include "dg_oracle.h"
#include <stdlib.h>
#include <string.h>

enum { T_INDICES = 0, T_ATMOS_PERT = 1, T_ATMOS_SPECIFIC = 2 };

static void filter_argument(const orc_filter_target *tg, double *fs, const double *Q,
                            const double *aux)
{
    if (tg->kind == T_INDICES) {
        for (int s = 0; s < tg->nfs; ++s) fs[s] = Q[tg->idx[s] - 1];
    } else if (tg->kind == T_ATMOS_PERT) { /* filters.jl:11-28 */
        for (int s = 0; s < 5; ++s) fs[s] = Q[s];
        fs[0] -= aux[tg->aux_ref_rho];
        fs[4] -= aux[tg->aux_ref_rhoe];
    } else { /* filters.jl:57-77 */
        const double rho_inv = 1 / Q[0];
        const double rho_ref_inv = 1 / aux[tg->aux_ref_rho];
        for (int s = 0; s < 5; ++s) fs[s] = Q[s] * rho_inv;
        fs[4] -= aux[tg->aux_ref_rhoe] * rho_ref_inv;
    }
}

static void filter_result(const orc_filter_target *tg, double *Q, const double *fs,
                          const double *aux)
{
    if (tg->kind == T_INDICES) {
        for (int s = 0; s < tg->nfs; ++s) Q[tg->idx[s] - 1] = fs[s];
    } else if (tg->kind == T_ATMOS_PERT) { /* filters.jl:30-48 */
        for (int s = 0; s < 5; ++s) Q[s] = fs[s];
        Q[0] += aux[tg->aux_ref_rho];
        Q[4] += aux[tg->aux_ref_rhoe];
    } else { /* filters.jl:79-99 */
        const double rho = Q[0];
        const double ratio = rho / aux[tg->aux_ref_rho];
        for (int s = 0; s < 5; ++s) Q[s] = fs[s] * rho;
        Q[4] += aux[tg->aux_ref_rhoe] * ratio;
    }
}

static void dir_flags(int dim, int direction, int *f1, int *f2, int *f3)
{
    if (direction == ORC_EVERY) {
        *f1 = *f2 = 1;
        *f3 = dim == 2 ? 0 : 1;
    } else if (direction == ORC_HORIZONTAL) {
        *f1 = 1;
        *f2 = dim == 2 ? 0 : 1;
        *f3 = 0;
    } else {
        *f1 = 0;
        *f2 = dim == 2 ? 1 : 0;
        *f3 = dim == 2 ? 0 : 1;
    }
}

/* the three tensor passes of one element; s = LDS array (Np, nfs), acc = per-thread
 * accumulators (Np, nfs); on return acc holds l_Qfiltered */
static void filter_passes(int Nq1, int Nq2, int Nq3, int nfs, int f1, int f2, int f3,
                          const double *F, double *s, double *acc)
{
    const int Np = Nq1 * Nq2 * Nq3;
    memset(acc, 0, sizeof(double) * Np * nfs);
    if (f1) {
        for (int k = 0; k < Nq3; ++k)
            for (int j = 0; j < Nq2; ++j)
                for (int i = 0; i < Nq1; ++i) {
                    const int ijk = i + Nq1 * (j + Nq2 * k);
                    for (int n = 0; n < Nq1; ++n)
                        for (int fs = 0; fs < nfs; ++fs)
                            acc[ijk + Np * fs] +=
                                F[i + Nq1 * n] * s[n + Nq1 * (j + Nq2 * k) + Np * fs];
                }
        if (f2 || f3) {
            memcpy(s, acc, sizeof(double) * Np * nfs);
            memset(acc, 0, sizeof(double) * Np * nfs);
        }
    }
    if (f2) {
        for (int k = 0; k < Nq3; ++k)
            for (int j = 0; j < Nq2; ++j)
                for (int i = 0; i < Nq1; ++i) {
                    const int ijk = i + Nq1 * (j + Nq2 * k);
                    for (int n = 0; n < Nq2; ++n)
                        for (int fs = 0; fs < nfs; ++fs)
                            acc[ijk + Np * fs] +=
                                F[j + Nq2 * n] * s[i + Nq1 * (n + Nq2 * k) + Np * fs];
                }
        if (f3) {
            memcpy(s, acc, sizeof(double) * Np * nfs);
            memset(acc, 0, sizeof(double) * Np * nfs);
        }
    }
    if (f3) {
        for (int k = 0; k < Nq3; ++k)
            for (int j = 0; j < Nq2; ++j)
                for (int i = 0; i < Nq1; ++i) {
                    const int ijk = i + Nq1 * (j + Nq2 * k);
                    for (int n = 0; n < Nq3; ++n)
                        for (int fs = 0; fs < nfs; ++fs)
                            acc[ijk + Np * fs] +=
                                F[k + Nq3 * n] * s[i + Nq1 * (j + Nq2 * n) + Np * fs];
                }
    }
}

static void geom(int dim, const int *Nq, int *Nq1, int *Nq2, int *Nq3)
{
    *Nq1 = Nq[0];
    *Nq2 = Nq[1];
    *Nq3 = dim == 2 ? 1 : Nq[dim - 1];
}

/* Filters.jl:651-794.  F is the (Nq, Nq) column-major filter matrix of this launch. */
void orc_apply_filter(int dim, const int *Nq, int direction, double *Q, int nstate,
                      const double *aux, int naux, const orc_filter_target *tg, const double *F,
                      int64_t nrealelem)
{
    int Nq1, Nq2, Nq3, f1, f2, f3;
    geom(dim, Nq, &Nq1, &Nq2, &Nq3);
    dir_flags(dim, direction, &f1, &f2, &f3);
    const int Np = Nq1 * Nq2 * Nq3, nfs = tg->nfs;
#pragma omp parallel
    {
        double *s = (double *)malloc(sizeof(double) * Np * nfs * 2);
        double *acc = s + Np * nfs;
        double lQ[ORC_MAXS], lA[ORC_MAXS], lF[ORC_MAXS];
#pragma omp for
        for (int64_t e = 0; e < nrealelem; ++e) {
            for (int ijk = 0; ijk < Np; ++ijk) {
                for (int q = 0; q < nstate; ++q) lQ[q] = Q[ijk + Np * (q + nstate * e)];
                for (int a = 0; a < naux; ++a) lA[a] = aux[ijk + Np * (a + naux * e)];
                for (int q = 0; q < nfs; ++q) lF[q] = -0.0;
                filter_argument(tg, lF, lQ, lA);
                for (int q = 0; q < nfs; ++q) s[ijk + Np * q] = lF[q];
            }
            filter_passes(Nq1, Nq2, Nq3, nfs, f1, f2, f3, F, s, acc);
            for (int ijk = 0; ijk < Np; ++ijk) {
                for (int q = 0; q < nstate; ++q) lQ[q] = Q[ijk + Np * (q + nstate * e)];
                for (int a = 0; a < naux; ++a) lA[a] = aux[ijk + Np * (a + naux * e)];
                for (int q = 0; q < nfs; ++q) lF[q] = acc[ijk + Np * q];
                filter_result(tg, lQ, lF, lA);
                for (int q = 0; q < nstate; ++q) Q[ijk + Np * (q + nstate * e)] = lQ[q];
            }
        }
        free(s);
    }
}

/* the reference's shared-memory tree: for n = 11..1, if nreduce >= 2^n, thread ijk (1-based)
 * adds entry ijk + 2^(n-1) when ijk <= 2^(n-1) and the partner exists */
static void tree_reduce(double *v, int count, int nreduce)
{
    for (int n = 11; n >= 1; --n) {
        if (nreduce >= (1 << n)) {
            const int h = 1 << (n - 1);
            for (int ijk = 1; ijk <= count; ++ijk)
                if (ijk <= h && ijk + h <= count) v[ijk - 1] += v[ijk + h - 1];
        }
    }
}

static int next_pow2(int n)
{
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

/* Filters.jl:796-884; target must be FilterIndices; Mcol = 0-based vgeo column of _M */
void orc_apply_tmar_filter(int dim, const int *Nq, double *Q, int nstate,
                           const orc_filter_target *tg, const double *vgeo, int nvgeo, int Mcol,
                           int64_t nrealelem)
{
    const int Nq1 = Nq[0], Nq2 = dim == 2 ? 1 : Nq[1], Nq3 = Nq[dim - 1];
    const int Np = Nq1 * Nq2 * Nq3, nij = Nq1 * Nq2;
    const int nreduce = next_pow2(nij);
#pragma omp parallel
    {
        double *sMJQ = (double *)malloc(sizeof(double) * nij * 2);
        double *sMJQc = sMJQ + nij;
#pragma omp for
        for (int64_t e = 0; e < nrealelem; ++e) {
            for (int sf = 0; sf < tg->nfs; ++sf) {
                const int s = tg->idx[sf] - 1;
                double *q = Q + Np * (s + (int64_t)nstate * e);
                for (int ij = 0; ij < nij; ++ij) {
                    double MJQ = 0, MJQc = 0;
                    for (int k = 0; k < Nq3; ++k) {
                        const int ijk = ij + nij * k;
                        const double MJ = vgeo[ijk + Np * (Mcol + (int64_t)nvgeo * e)];
                        const double Qs = q[ijk];
                        const double Qc = Qs >= 0 ? Qs : 0.0;
                        MJQ += MJ * Qs;
                        MJQc += MJ * Qc;
                    }
                    sMJQ[ij] = MJQ;
                    sMJQc[ij] = MJQc;
                }
                tree_reduce(sMJQ, nij, nreduce);
                tree_reduce(sMJQc, nij, nreduce);
                const double avg = sMJQ[0], cavg = sMJQc[0];
                const double r = avg > 0 ? avg / cavg : 0.0;
                for (int ijk = 0; ijk < Np; ++ijk) {
                    const double Qs = q[ijk];
                    q[ijk] = Qs >= 0 ? r * Qs : 0.0;
                }
            }
        }
        free(sMJQ);
    }
}

/* Filters.jl:893-1071 */
void orc_apply_mp_filter(int dim, const int *Nq, int direction, double *Q, int nstate,
                         const double *aux, int naux, const orc_filter_target *tg,
                         const double *F, const double *vgeo, int nvgeo, int Mcol,
                         int64_t nrealelem)
{
    int Nq1, Nq2, Nq3, f1, f2, f3;
    geom(dim, Nq, &Nq1, &Nq2, &Nq3);
    dir_flags(dim, direction, &f1, &f2, &f3);
    const int Np = Nq1 * Nq2 * Nq3, nfs = tg->nfs;
    const int nreduce = next_pow2(Np);
#pragma omp parallel
    {
        double *s = (double *)malloc(sizeof(double) * Np * (2 * nfs + 3 * nstate + 1));
        double *acc = s + Np * nfs;
        double *MQB = acc + Np * nfs, *MQA = MQB + Np * nstate, *pQ = MQA + Np * nstate;
        double *lM = pQ + Np * nstate;
        double lQ[ORC_MAXS], lA[ORC_MAXS], lF[ORC_MAXS];
#pragma omp for
        for (int64_t e = 0; e < nrealelem; ++e) {
            for (int ijk = 0; ijk < Np; ++ijk) {
                for (int q = 0; q < nstate; ++q) lQ[q] = Q[ijk + Np * (q + nstate * e)];
                for (int a = 0; a < naux; ++a) lA[a] = aux[ijk + Np * (a + naux * e)];
                lM[ijk] = vgeo[ijk + Np * (Mcol + (int64_t)nvgeo * e)];
                for (int q = 0; q < nstate; ++q) MQB[ijk + Np * q] = lM[ijk] * lQ[q];
                for (int q = 0; q < nfs; ++q) lF[q] = -0.0;
                filter_argument(tg, lF, lQ, lA);
                for (int q = 0; q < nfs; ++q) s[ijk + Np * q] = lF[q];
            }
            filter_passes(Nq1, Nq2, Nq3, nfs, f1, f2, f3, F, s, acc);
            for (int ijk = 0; ijk < Np; ++ijk) {
                for (int q = 0; q < nstate; ++q) lQ[q] = Q[ijk + Np * (q + nstate * e)];
                for (int a = 0; a < naux; ++a) lA[a] = aux[ijk + Np * (a + naux * e)];
                for (int q = 0; q < nfs; ++q) lF[q] = acc[ijk + Np * q];
                filter_result(tg, lQ, lF, lA);
                for (int q = 0; q < nstate; ++q) {
                    pQ[ijk + Np * q] = lQ[q];
                    MQA[ijk + Np * q] = lM[ijk] * lQ[q];
                }
            }
            tree_reduce(lM, Np, nreduce);
            for (int q = 0; q < nstate; ++q) {
                tree_reduce(MQB + Np * q, Np, nreduce);
                tree_reduce(MQA + Np * q, Np, nreduce);
            }
            const double Minv = 1 / lM[0];
            for (int ijk = 0; ijk < Np; ++ijk)
                for (int q = 0; q < nstate; ++q)
                    Q[ijk + Np * (q + nstate * e)] =
                        pQ[ijk + Np * q] + Minv * (MQB[Np * q] - MQA[Np * q]);
        }
        free(s);
    }
}
