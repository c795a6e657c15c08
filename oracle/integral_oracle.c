/* ORACLE -- TEST INFRASTRUCTURE ONLY (see dg_oracle.h).
 *
 * Column (stack) integrals restated from the reference:
 *   orc_indefinite_stack_integral          kernel_indefinite_stack_integral!
 *                                          src/Numerics/DGMethods/DGModel_kernels.jl:1903-2010
 *   orc_reverse_indefinite_stack_integral  kernel_reverse_indefinite_stack_integral!  :2012-2104
 * launched as DGModel.jl:445-529 does (one work-group per horizontal element, one thread per
 * (i, j) pencil, elements of a stack contiguous: e = ev + (eh - 1) nvertelem).
 * Pinned by test/Numerics/DGMethods/integral_test.jl (IntegralTestModel, analytic integrals).
 */
#include <math.h>
#include <stdlib.h>

#include "dg_oracle.h"

void orc_indefinite_stack_integral(const orc_integral_law *law, const orc_grid *g, int nvertelem,
                                   const double *Q, double *aux, const double *Imat, int JcV,
                                   int64_t h0, int64_t h1)
{
    const int Nq1 = g->Nq[0], Nq2 = g->Nq[1], Nq3 = g->Nq[2], Np = g->Np;
    const int nout = law->nout, ns = law->ns, naux = law->naux;
#pragma omp parallel for
    for (int64_t eh = h0; eh < h1; ++eh)
        for (int j = 0; j < Nq2; ++j)
            for (int i = 0; i < Nq1; ++i) {
                double lint[ORC_MAXS][16], lker[ORC_MAXS][16], lQ[ORC_MAXS], lA[ORC_MAXS];
                for (int k = 0; k < Nq3; ++k)
                    for (int s = 0; s < nout; ++s) lint[s][k] = 0;
                for (int ev = 0; ev < nvertelem; ++ev) {
                    const int64_t e = ev + eh * nvertelem;
                    for (int k = 0; k < Nq3; ++k) {
                        const int ijk = i + Nq1 * (j + Nq2 * k);
                        const double Jc = g->vgeo[ijk + (int64_t)Np * (JcV + (int64_t)g->nvgeo * e)];
                        for (int s = 0; s < ns; ++s) lQ[s] = Q[ijk + (int64_t)Np * (s + (int64_t)ns * e)];
                        for (int s = 0; s < naux; ++s)
                            lA[s] = aux[ijk + (int64_t)Np * (s + (int64_t)naux * e)];
                        double col[ORC_MAXS];
                        law->load(law->p, col, lQ, lA);
                        for (int s = 0; s < nout; ++s) lker[s][k] = col[s] * Jc;
                    }
                    for (int s = 0; s < nout; ++s)
                        for (int k = 0; k < Nq3; ++k)
                            for (int n = 0; n < Nq3; ++n)
                                lint[s][k] += Imat[k + Nq3 * n] * lker[s][n];
                    for (int k = 0; k < Nq3; ++k) {
                        const int ijk = i + Nq1 * (j + Nq2 * k);
                        double col[ORC_MAXS], la[ORC_MAXS];
                        for (int s = 0; s < nout; ++s) col[s] = lint[s][k];
                        for (int s = 0; s < naux; ++s)
                            la[s] = aux[ijk + (int64_t)Np * (s + (int64_t)naux * e)];
                        law->set(law->p, la, col);
                        for (int s = 0; s < naux; ++s)
                            aux[ijk + (int64_t)Np * (s + (int64_t)naux * e)] = la[s];
                        /* reset the background value for the next element */
                        for (int s = 0; s < nout; ++s) lint[s][k] = lint[s][Nq3 - 1];
                    }
                }
            }
}

void orc_reverse_indefinite_stack_integral(const orc_integral_law *law, const orc_grid *g,
                                           int nvertelem, const double *Q, double *aux,
                                           int64_t h0, int64_t h1)
{
    const int Nq1 = g->Nq[0], Nq2 = g->Nq[1], Nq3 = g->Nq[2], Np = g->Np;
    const int nout = law->nrout, ns = law->ns, naux = law->naux;
#pragma omp parallel for
    for (int64_t eh = h0; eh < h1; ++eh)
        for (int j = 0; j < Nq2; ++j)
            for (int i = 0; i < Nq1; ++i) {
                double lT[ORC_MAXS], lV[ORC_MAXS], lQ[ORC_MAXS], lA[ORC_MAXS];
                {
                    const int ijk = i + Nq1 * (j + Nq2 * (Nq3 - 1));
                    const int64_t et = (nvertelem - 1) + eh * nvertelem;
                    for (int s = 0; s < ns; ++s) lQ[s] = Q[ijk + (int64_t)Np * (s + (int64_t)ns * et)];
                    for (int s = 0; s < naux; ++s)
                        lA[s] = aux[ijk + (int64_t)Np * (s + (int64_t)naux * et)];
                    law->rload(law->p, lT, lQ, lA);
                }
                for (int ev = 0; ev < nvertelem; ++ev) {
                    const int64_t e = ev + eh * nvertelem;
                    for (int k = 0; k < Nq3; ++k) {
                        const int ijk = i + Nq1 * (j + Nq2 * k);
                        for (int s = 0; s < ns; ++s) lQ[s] = Q[ijk + (int64_t)Np * (s + (int64_t)ns * e)];
                        for (int s = 0; s < naux; ++s)
                            lA[s] = aux[ijk + (int64_t)Np * (s + (int64_t)naux * e)];
                        law->rload(law->p, lV, lQ, lA);
                        for (int s = 0; s < nout; ++s) lV[s] = lT[s] - lV[s];
                        law->rset(law->p, lA, lV);
                        for (int s = 0; s < naux; ++s)
                            aux[ijk + (int64_t)Np * (s + (int64_t)naux * e)] = lA[s];
                    }
                }
            }
}

/* ---- IntegralTestModel{3}: aux = int.a int.b rev_int.a rev_int.b coord[3] a b rev_a rev_b -- */
static void it_load(const void *p, double *f, const double *Q, const double *aux)
{
    (void)p; (void)Q;
    const double x = aux[4], y = aux[5], z = aux[6];
    f[0] = x + y;
    f[1] = 2 * x + sin(x) * y - (z - 1) * (z - 1) * (y * y);
}
static void it_set(const void *p, double *aux, const double *I) { (void)p; aux[0] = I[0]; aux[1] = I[1]; }
static void it_rload(const void *p, double *I, const double *Q, const double *aux)
{
    (void)p; (void)Q;
    I[0] = aux[0];
    I[1] = aux[1];
}
static void it_rset(const void *p, double *aux, const double *I) { (void)p; aux[2] = I[0]; aux[3] = I[1]; }

orc_integral_law *orc_integral_test_law(void)
{
    orc_integral_law *l = (orc_integral_law *)calloc(1, sizeof(*l));
    l->nout = l->nrout = 2;
    l->ns = 0;
    l->naux = 11;
    l->load = it_load;
    l->set = it_set;
    l->rload = it_rload;
    l->rset = it_rset;
    return l;
}

/* ---- integrand_s = scale_s * field_s, stored to aux column dst_s (cmdg_stack_integral_desc) -- */
typedef struct {
    int nout, is_state[ORC_MAXS], src[ORC_MAXS], dst[ORC_MAXS], rsrc[ORC_MAXS], rdst[ORC_MAXS];
    double scale[ORC_MAXS];
} fields_t;
static void fl_load(const void *p, double *f, const double *Q, const double *aux)
{
    const fields_t *m = (const fields_t *)p;
    for (int s = 0; s < m->nout; ++s) f[s] = m->scale[s] * (m->is_state[s] ? Q[m->src[s]] : aux[m->src[s]]);
}
static void fl_set(const void *p, double *aux, const double *I)
{
    const fields_t *m = (const fields_t *)p;
    for (int s = 0; s < m->nout; ++s) aux[m->dst[s]] = I[s];
}
static void fl_rload(const void *p, double *I, const double *Q, const double *aux)
{
    const fields_t *m = (const fields_t *)p;
    (void)Q;
    for (int s = 0; s < m->nout; ++s) I[s] = aux[m->rsrc[s]];
}
static void fl_rset(const void *p, double *aux, const double *I)
{
    const fields_t *m = (const fields_t *)p;
    for (int s = 0; s < m->nout; ++s) aux[m->rdst[s]] = I[s];
}
orc_integral_law *orc_integral_fields_law(int nout, const int *src_is_state, const int *src_col,
                                          const double *scale, const int *dst_col,
                                          const int *rsrc_col, const int *rdst_col, int ns,
                                          int naux)
{
    orc_integral_law *l = (orc_integral_law *)calloc(1, sizeof(*l));
    fields_t *m = (fields_t *)calloc(1, sizeof(*m));
    m->nout = nout;
    for (int s = 0; s < nout; ++s) {
        m->is_state[s] = src_is_state ? src_is_state[s] : 0;
        m->src[s] = src_col ? src_col[s] : 0;
        m->scale[s] = scale ? scale[s] : 1.0;
        m->dst[s] = dst_col ? dst_col[s] : 0;
        m->rsrc[s] = rsrc_col ? rsrc_col[s] : 0;
        m->rdst[s] = rdst_col ? rdst_col[s] : 0;
    }
    l->nout = l->nrout = nout;
    l->ns = ns;
    l->naux = naux;
    l->p = m;
    l->load = fl_load;
    l->set = fl_set;
    l->rload = fl_rload;
    l->rset = fl_rset;
    return l;
}
void orc_integral_law_free(orc_integral_law *law)
{
    if (!law) return;
    free((void *)law->p);
    free(law);
}
