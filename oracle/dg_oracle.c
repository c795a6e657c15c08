/* ORACLE -- TEST INFRASTRUCTURE ONLY (see dg_oracle.h).
 *
 * Each function restates one KernelAbstractions kernel of the reference
 * (src/Numerics/DGMethods/DGModel_kernels.jl) as plain loops: one OpenMP
 * iteration per workgroup (= element), the work-items (i,j) or face node n as
 * inner loops, and the same accumulation order as the reference.  Compile with
 * -ffp-contract=off: the reference's Julia code does not fuse multiply-adds.
 */
#include "dg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* vgeo column ids, 0-based (Grids.jl:76-92) */
enum { XI1X1 = 0, XI2X1, XI3X1, XI1X2, XI2X2, XI3X2, XI1X3, XI2X3, XI3X3, VM, VMI };
/* sgeo row ids (Grids.jl:129-130) */
enum { SN1 = 0, SN2, SN3, SSM, SVMI };

#define NEGZERO (-0.0)
#define MAXNQ 16

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
int orc_get_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* bench.py's cpu_baseline only.  Copies an (nchunks, chunk_bytes) array with the loop schedule of
 * the kernels below (static, over elements), so that every thread first touches the pages of the
 * elements it will work on: on a multi-socket host a numpy-initialised array lives on the NUMA
 * node of the one thread that filled it. */
void orc_first_touch_copy(void *dst, const void *src, int64_t nchunks, int64_t chunk_bytes)
{
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < nchunks; ++c)
        memcpy((char *)dst + c * chunk_bytes, (const char *)src + c * chunk_bytes, (size_t)chunk_bytes);
}

/* a[i] = b[i] + s * c[i]: one streaming pass (24 B per entry), the host's memory-bandwidth
 * ceiling next to the cpu_baseline figure */
void orc_stream_triad(double *a, const double *b, const double *c, double s, int64_t n)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) a[i] = b[i] + s * c[i];
}

static inline double VG(const orc_grid *g, int ijk, int col, int64_t e)
{
    return g->vgeo[ijk + (int64_t)g->Np * (col + (int64_t)g->nvgeo * e)];
}
static inline double SG(const orc_grid *g, int c, int n, int f, int64_t e)
{
    return g->sgeo[c + 5 * (n + (int64_t)g->Nfp * (f + (int64_t)g->nface * e))];
}
static inline void fillnz(double *a, int n)
{
    for (int i = 0; i < n; ++i) a[i] = NEGZERO;
}
static inline void loadv(double *dst, const double *arr, int ijk, int nvar, int64_t e, int Np)
{
    for (int s = 0; s < nvar; ++s) dst[s] = arr[ijk + (int64_t)Np * (s + (int64_t)nvar * e)];
}

/* ------------------------------------------------------------------ */
/* volume_tendency!  DGModel_kernels.jl:64-309 (generic) / :312-548 (vertical) */
void orc_volume_tendency(const orc_physics *ph, const orc_grid *g, int model_dir, int direction,
                         double *tendency, const double *Q, const double *gf, const double *hyp,
                         const double *aux, double t, double alpha, double beta, int add_source)
{
    const int ns = ph->ns, Np = g->Np;
    const int Nq1 = g->Nq[0], Nq2 = g->Nq[1], Nq3 = g->Nq[2];
    const double *Dh = g->D[0], *Dv = g->D[2];
#pragma omp parallel
    {
        double *lt = (double *)malloc(sizeof(double) * Np * ns);       /* local_tendency[k,s] per (i,j) */
        double *sh = (double *)malloc(sizeof(double) * 2 * Nq1 * Nq2 * ns); /* shared_flux */
        double lQ[ORC_MAXS], lgf[ORC_MAXS], lhyp[ORC_MAXS], laux[ORC_MAXS * 2];
        double F[3 * ORC_MAXS], Ft[3 * ORC_MAXS], f3[ORC_MAXS], S[ORC_MAXS];
#pragma omp for schedule(static)
        for (int64_t e = 0; e < g->nreal; ++e) {
            for (int q = 0; q < Np * ns; ++q) lt[q] = 0.0;
            if (direction != ORC_VERTICAL) {
                for (int k = 0; k < Nq3; ++k) {
                    for (int j = 0; j < Nq2; ++j)
                        for (int i = 0; i < Nq1; ++i) {
                            const int ijk = i + Nq1 * (j + Nq2 * k);
                            const double M = VG(g, ijk, VM, e);
                            const double x11 = VG(g, ijk, XI1X1, e), x12 = VG(g, ijk, XI1X2, e),
                                         x13 = VG(g, ijk, XI1X3, e);
                            const double x21 = VG(g, ijk, XI2X1, e), x22 = VG(g, ijk, XI2X2, e),
                                         x23 = VG(g, ijk, XI2X3, e);
                            loadv(lQ, Q, ijk, ns, e, Np);
                            loadv(laux, aux, ijk, ph->naux, e, Np);
                            loadv(lgf, gf, ijk, ph->ngf, e, Np);
                            loadv(lhyp, hyp, ijk, ph->nhyp, e, Np);
                            fillnz(F, 3 * ns);
                            ph->flux_first_order(ph->p, F, lQ, laux, t, model_dir);
                            double *s1 = sh + (size_t)ns * (i + Nq1 * j) * 2;
                            for (int s = 0; s < ns; ++s) {
                                s1[2 * s + 0] = F[0 + 3 * s];
                                s1[2 * s + 1] = F[1 + 3 * s];
                                f3[s] = F[2 + 3 * s];
                            }
                            fillnz(F, 3 * ns);
                            ph->flux_second_order(ph->p, F, lQ, lgf, lhyp, laux, t);
                            for (int s = 0; s < ns; ++s) {
                                s1[2 * s + 0] += F[0 + 3 * s];
                                s1[2 * s + 1] += F[1 + 3 * s];
                                f3[s] += F[2 + 3 * s];
                            }
                            for (int s = 0; s < ns; ++s) {
                                const double F1 = s1[2 * s], F2 = s1[2 * s + 1], F3 = f3[s];
                                s1[2 * s + 0] = M * (x11 * F1 + x12 * F2 + x13 * F3);
                                s1[2 * s + 1] = M * (x21 * F1 + x22 * F2 + x23 * F3);
                            }
                            if (add_source) {
                                fillnz(S, ns);
                                ph->source(ph->p, S, lQ, lgf, laux, t, model_dir);
                                for (int s = 0; s < ns; ++s) lt[ijk + Np * s] += S[s];
                            }
                        }
                    /* @synchronize ; weak "inside metrics" derivative */
                    for (int j = 0; j < Nq2; ++j)
                        for (int i = 0; i < Nq1; ++i) {
                            const int ijk = i + Nq1 * (j + Nq2 * k);
                            const double MI = VG(g, ijk, VMI, e);
                            for (int s = 0; s < ns; ++s)
                                for (int n = 0; n < Nq1; ++n) {
                                    lt[ijk + Np * s] += MI * Dh[n + Nq1 * i] *
                                                        sh[((size_t)ns * (n + Nq1 * j) + s) * 2 + 0];
                                    lt[ijk + Np * s] += MI * Dh[n + Nq1 * j] *
                                                        sh[((size_t)ns * (i + Nq1 * n) + s) * 2 + 1];
                                }
                        }
                }
            } else {
                for (int j = 0; j < Nq2; ++j)
                    for (int i = 0; i < Nq1; ++i)
                        for (int k = 0; k < Nq3; ++k) {
                            const int ijk = i + Nq1 * (j + Nq2 * k);
                            const double M = VG(g, ijk, VM, e);
                            const double z1 = VG(g, ijk, XI3X1, e), z2 = VG(g, ijk, XI3X2, e),
                                         z3 = VG(g, ijk, XI3X3, e);
                            loadv(lQ, Q, ijk, ns, e, Np);
                            loadv(laux, aux, ijk, ph->naux, e, Np);
                            loadv(lgf, gf, ijk, ph->ngf, e, Np);
                            loadv(lhyp, hyp, ijk, ph->nhyp, e, Np);
                            fillnz(F, 3 * ns);
                            ph->flux_first_order(ph->p, F, lQ, laux, t, model_dir);
                            for (int q = 0; q < 3 * ns; ++q) Ft[q] = F[q];
                            fillnz(F, 3 * ns);
                            ph->flux_second_order(ph->p, F, lQ, lgf, lhyp, laux, t);
                            for (int q = 0; q < 3 * ns; ++q) Ft[q] += F[q];
                            for (int s = 0; s < ns; ++s) {
                                const double F1 = Ft[3 * s], F2 = Ft[3 * s + 1], F3 = Ft[3 * s + 2];
                                Ft[3 * s] = M * (z1 * F1 + z2 * F2 + z3 * F3);
                            }
                            for (int n = 0; n < Nq3; ++n) {
                                const int ijn = i + Nq1 * (j + Nq2 * n);
                                const double MI = VG(g, ijn, VMI, e);
                                for (int s = 0; s < ns; ++s)
                                    lt[ijn + Np * s] += MI * Dv[k + Nq3 * n] * Ft[3 * s];
                            }
                            if (add_source) {
                                fillnz(S, ns);
                                ph->source(ph->p, S, lQ, lgf, laux, t, model_dir);
                                for (int s = 0; s < ns; ++s) lt[ijk + Np * s] += S[s];
                            }
                        }
            }
            for (int s = 0; s < ns; ++s)
                for (int ijk = 0; ijk < Np; ++ijk) {
                    double *T = &tendency[ijk + (int64_t)Np * (s + (int64_t)ns * e)];
                    if (beta != 0)
                        *T = alpha * lt[ijk + Np * s] + beta * (*T);
                    else
                        *T = alpha * lt[ijk + Np * s];
                }
        }
        free(lt);
        free(sh);
    }
}

/* ------------------------------------------------------------------ */
/* shared face prologue (DGModel_kernels.jl:673-692 and equivalents) */
typedef struct {
    double n[3], sM, vMI;
    int64_t eM, eP;
    int vidM, vidP, bctag;
} face_pt;

static inline void face_setup(const orc_grid *g, int64_t e, int f, int n, face_pt *fp)
{
    fp->n[0] = SG(g, SN1, n, f, e);
    fp->n[1] = SG(g, SN2, n, f, e);
    fp->n[2] = SG(g, SN3, n, f, e);
    fp->sM = SG(g, SSM, n, f, e);
    fp->vMI = SG(g, SVMI, n, f, e);
    fp->bctag = (int)g->elemtobndy[f + (int64_t)g->nface * e];
    const int64_t idM = g->vmapM[n + (int64_t)g->Nfp * (f + (int64_t)g->nface * e)];
    const int64_t idP = g->vmapP[n + (int64_t)g->Nfp * (f + (int64_t)g->nface * e)];
    fp->eM = e;
    fp->eP = (idP - 1) / g->Np;
    fp->vidM = (int)((idM - 1) % g->Np);
    fp->vidP = (int)((idP - 1) % g->Np);
    if (fp->bctag != 0) {
        fp->eP = e;
        fp->vidP = fp->vidM;
    }
}
static inline void face_range(const orc_grid *g, int direction, int *f0, int *f1)
{
    *f0 = 0;
    *f1 = g->nface;
    if (direction == ORC_VERTICAL) *f0 = g->nface - 2;
    if (direction == ORC_HORIZONTAL) *f1 = g->nface - 2;
}
static inline int face_npts(const orc_grid *g, int f) { return g->Np / g->Nq[f / 2]; }

/* numerical_flux_first_order!  NumericalFluxes.jl:223-285 (Rusanov), :300-340 (central) */
static void nf_first_order(const orc_physics *ph, double *fluxn, const double *n, const double *QM,
                           const double *auxM, const double *QP, const double *auxP, double t,
                           int facedir)
{
    const int ns = ph->ns;
    if (ph->nf_first >= ORC_NF_ROE) {  /* methods a law defines for itself */
        ph->numerical_flux_law(ph->p, ph->nf_first, fluxn, n, QM, auxM, QP, auxP, t, facedir);
        return;
    }
    double FM[3 * ORC_MAXS], FP[3 * ORC_MAXS];
    fillnz(FM, 3 * ns);
    ph->flux_first_order(ph->p, FM, QM, auxM, t, facedir);
    fillnz(FP, 3 * ns);
    ph->flux_first_order(ph->p, FP, QP, auxP, t, facedir);
    const double nh[3] = {n[0] / 2, n[1] / 2, n[2] / 2};
    for (int s = 0; s < ns; ++s)
        fluxn[s] += (FM[3 * s] + FP[3 * s]) * nh[0] + (FM[3 * s + 1] + FP[3 * s + 1]) * nh[1] +
                    (FM[3 * s + 2] + FP[3 * s + 2]) * nh[2];
    if (ph->nf_first == ORC_NF_RUSANOV) {
        double wM[ORC_MAXS], wP[ORC_MAXS];
        ph->wavespeed(ph->p, wM, n, QM, auxM, t, facedir);
        ph->wavespeed(ph->p, wP, n, QP, auxP, t, facedir);
        double pen[ORC_MAXS];
        for (int s = 0; s < ns; ++s) {
            const double mw = wM[s] > wP[s] ? wM[s] : wP[s];
            pen[s] = mw * (QM[s] - QP[s]);
        }
        if (ph->update_penalty) ph->update_penalty(ph->p, pen, n, QM, QP);
        for (int s = 0; s < ns; ++s) fluxn[s] += pen[s] / 2;
    }
}

/* ------------------------------------------------------------------ */
/* dgsem_interface_tendency!  DGModel_kernels.jl:588-901 */
void orc_interface_tendency(const orc_physics *ph, const orc_grid *g, int direction,
                            double *tendency, const double *Q, const double *gf,
                            const double *hyp, const double *aux, double t, const int64_t *elems,
                            int64_t nelems, double alpha)
{
    const int ns = ph->ns, Np = g->Np, naux = ph->naux, ngf = ph->ngf, nhyp = ph->nhyp;
    const int Nqk = g->Nq[2];
    int f0, f1;
    face_range(g, direction, &f0, &f1);
#pragma omp parallel for schedule(static)
    for (int64_t eI = 0; eI < nelems; ++eI) {
        const int64_t e = elems[eI] - 1;
        double QM[ORC_MAXS], gfM[ORC_MAXS], hypM[ORC_MAXS], auxM[ORC_MAXS * 2];
        double QPn[ORC_MAXS], QPd[ORC_MAXS], auxPn[ORC_MAXS * 2], auxPd[ORC_MAXS * 2];
        double gfP[ORC_MAXS], hypP[ORC_MAXS];
        double Q1[ORC_MAXS], gf1[ORC_MAXS], aux1[ORC_MAXS * 2];
        double flux[ORC_MAXS], FM[3 * ORC_MAXS], FP[3 * ORC_MAXS];
        for (int f = f0; f < f1; ++f) {
            const int facedir = f < g->nface - 2 ? ORC_HORIZONTAL : ORC_VERTICAL;
            const int npts = face_npts(g, f);
            for (int n = 0; n < npts; ++n) {
                face_pt fp;
                face_setup(g, e, f, n, &fp);
                loadv(QM, Q, fp.vidM, ns, fp.eM, Np);
                loadv(gfM, gf, fp.vidM, ngf, fp.eM, Np);
                loadv(hypM, hyp, fp.vidM, nhyp, fp.eM, Np);
                loadv(auxM, aux, fp.vidM, naux, fp.eM, Np);
                loadv(QPn, Q, fp.vidP, ns, fp.eP, Np);
                memcpy(QPd, QPn, sizeof(double) * ns);
                loadv(gfP, gf, fp.vidP, ngf, fp.eP, Np);
                loadv(hypP, hyp, fp.vidP, nhyp, fp.eP, Np);
                loadv(auxPn, aux, fp.vidP, naux, fp.eP, Np);
                memcpy(auxPd, auxPn, sizeof(double) * naux);
                fillnz(flux, ns);
                if (fp.bctag == 0) {
                    nf_first_order(ph, flux, fp.n, QM, auxM, QPn, auxPn, t, facedir);
                    /* CentralNumericalFluxSecondOrder  NumericalFluxes.jl:670-715 */
                    fillnz(FM, 3 * ns);
                    ph->flux_second_order(ph->p, FM, QM, gfM, hypM, auxM, t);
                    fillnz(FP, 3 * ns);
                    ph->flux_second_order(ph->p, FP, QPd, gfP, hypP, auxPd, t);
                    const double nh[3] = {fp.n[0] / 2, fp.n[1] / 2, fp.n[2] / 2};
                    for (int s = 0; s < ns; ++s)
                        flux[s] += (FM[3 * s] + FP[3 * s]) * nh[0] +
                                   (FM[3 * s + 1] + FP[3 * s + 1]) * nh[1] +
                                   (FM[3 * s + 2] + FP[3 * s + 2]) * nh[2];
                } else {
                    memset(Q1, 0, sizeof(Q1));
                    memset(gf1, 0, sizeof(gf1));
                    memset(aux1, 0, sizeof(aux1));
                    if (f == 4) { /* bottom face: first interior node (:786-816) */
                        loadv(Q1, Q, n + Nqk * Nqk, ns, fp.eM, Np);
                        loadv(gf1, gf, n + Nqk * Nqk, ngf, fp.eM, Np);
                        loadv(aux1, aux, n + Nqk * Nqk, naux, fp.eM, Np);
                    }
                    /* numerical_boundary_flux_first_order!  NumericalFluxes.jl:163-205 */
                    ph->boundary_state(ph->p, ORC_BS_FIRST, fp.bctag, QPn, auxPn, fp.n, QM, auxM, t,
                                       Q1, aux1);
                    nf_first_order(ph, flux, fp.n, QM, auxM, QPn, auxPn, t, facedir);
                    /* normal_boundary_flux_second_order!  NumericalFluxes.jl:872-918 */
                    fillnz(FP, 3 * ns);
                    ph->boundary_flux_second_order(ph->p, fp.bctag, FP, QPd, gfP, hypP, auxPd, fp.n,
                                                   QM, gfM, hypM, auxM, t, Q1, gf1, aux1);
                    for (int s = 0; s < ns; ++s)
                        flux[s] += FP[3 * s] * fp.n[0] + FP[3 * s + 1] * fp.n[1] +
                                   FP[3 * s + 2] * fp.n[2];
                }
                for (int s = 0; s < ns; ++s)
                    tendency[fp.vidM + (int64_t)Np * (s + (int64_t)ns * fp.eM)] -=
                        alpha * fp.vMI * fp.sM * flux[s];
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* volume_gradients!  DGModel_kernels.jl:934-1130 (generic) / :1132-1328 (vertical) */
void orc_volume_gradients(const orc_physics *ph, const orc_grid *g, int direction,
                          const double *Q, double *gf, double *hypgrad, const double *aux,
                          double t, int increment)
{
    const int ns = ph->ns, Np = g->Np, ngrad = ph->ngrad, ngf = ph->ngf, ngl = ph->ngl;
    const int Nq1 = g->Nq[0], Nq2 = g->Nq[1], Nq3 = g->Nq[2];
    const double *Dh = g->D[0], *Dv = g->D[2];
    const int nhg = 3 * ngl;
#pragma omp parallel
    {
        double *ltg = (double *)malloc(sizeof(double) * 3 * ngrad * Np); /* [d + 3*(s + ngrad*ijk)] */
        double *sh = (double *)malloc(sizeof(double) * Np * ngrad);      /* G at every node */
        double *Gz = (double *)malloc(sizeof(double) * ngrad * Nq3);
        double lQ[ORC_MAXS], laux[ORC_MAXS * 2], G[ORC_MAXS], lgf[ORC_MAXS];
#pragma omp for schedule(static)
        for (int64_t e = 0; e < g->nreal; ++e) {
            fillnz(ltg, 3 * ngrad * Np);
            for (int ijk = 0; ijk < Np; ++ijk) {
                loadv(lQ, Q, ijk, ns, e, Np);
                loadv(laux, aux, ijk, ph->naux, e, Np);
                fillnz(G, ngrad);
                ph->gradient_argument(ph->p, G, lQ, laux, t);
                for (int s = 0; s < ngrad; ++s) sh[ijk + Np * s] = G[s];
            }
            if (direction != ORC_VERTICAL) {
                for (int k = 0; k < Nq3; ++k)
                    for (int j = 0; j < Nq2; ++j)
                        for (int i = 0; i < Nq1; ++i) {
                            const int ijk = i + Nq1 * (j + Nq2 * k);
                            const double x11 = VG(g, ijk, XI1X1, e), x12 = VG(g, ijk, XI1X2, e),
                                         x13 = VG(g, ijk, XI1X3, e);
                            const double x21 = VG(g, ijk, XI2X1, e), x22 = VG(g, ijk, XI2X2, e),
                                         x23 = VG(g, ijk, XI2X3, e);
                            for (int s = 0; s < ngrad; ++s) {
                                double G1 = 0.0, G2 = 0.0;
                                for (int n = 0; n < Nq1; ++n) {
                                    G1 += Dh[i + Nq1 * n] * sh[n + Nq1 * (j + Nq2 * k) + Np * s];
                                    G2 += Dh[j + Nq1 * n] * sh[i + Nq1 * (n + Nq2 * k) + Np * s];
                                }
                                double *l = ltg + 3 * (s + ngrad * ijk);
                                l[0] += x11 * G1;
                                l[1] += x12 * G1;
                                l[2] += x13 * G1;
                                l[0] += x21 * G2;
                                l[1] += x22 * G2;
                                l[2] += x23 * G2;
                            }
                        }
            } else {
                for (int j = 0; j < Nq2; ++j)
                    for (int i = 0; i < Nq1; ++i) {
                        fillnz(Gz, ngrad * Nq3);
                        for (int k = 0; k < Nq3; ++k)
                            for (int s = 0; s < ngrad; ++s)
                                for (int n = 0; n < Nq3; ++n)
                                    Gz[s + ngrad * n] +=
                                        Dv[n + Nq3 * k] * sh[i + Nq1 * (j + Nq2 * k) + Np * s];
                        for (int k = 0; k < Nq3; ++k) {
                            const int ijk = i + Nq1 * (j + Nq2 * k);
                            const double z1 = VG(g, ijk, XI3X1, e), z2 = VG(g, ijk, XI3X2, e),
                                         z3 = VG(g, ijk, XI3X3, e);
                            for (int s = 0; s < ngrad; ++s) {
                                double *l = ltg + 3 * (s + ngrad * ijk);
                                l[0] += z1 * Gz[s + ngrad * k];
                                l[1] += z2 * Gz[s + ngrad * k];
                                l[2] += z3 * Gz[s + ngrad * k];
                            }
                        }
                    }
            }
            for (int ijk = 0; ijk < Np; ++ijk) {
                const double *l = ltg + 3 * ngrad * ijk;
                for (int s = 0; s < ngl; ++s)
                    for (int d = 0; d < 3; ++d) {
                        double *h = &hypgrad[ijk + (int64_t)Np * (3 * s + d + (int64_t)nhg * e)];
                        const double v = l[d + 3 * ph->hv_indexmap[s]];
                        if (increment)
                            *h += v;
                        else
                            *h = v;
                    }
                if (ngf > 0) {
                    loadv(lQ, Q, ijk, ns, e, Np);
                    loadv(laux, aux, ijk, ph->naux, e, Np);
                    fillnz(lgf, ngf);
                    ph->gradient_flux(ph->p, lgf, l, lQ, laux, t);
                    for (int s = 0; s < ngf; ++s) {
                        double *o = &gf[ijk + (int64_t)Np * (s + (int64_t)ngf * e)];
                        if (increment)
                            *o += lgf[s];
                        else
                            *o = lgf[s];
                    }
                }
            }
        }
        free(ltg);
        free(sh);
        free(Gz);
    }
}

/* ------------------------------------------------------------------ */
/* dgsem_interface_gradients!  DGModel_kernels.jl:1365-1651 */
void orc_interface_gradients(const orc_physics *ph, const orc_grid *g, int direction,
                             const double *Q, double *gf, double *hypgrad, const double *aux,
                             double t, const int64_t *elems, int64_t nelems)
{
    const int ns = ph->ns, Np = g->Np, naux = ph->naux, ngrad = ph->ngrad, ngf = ph->ngf,
              ngl = ph->ngl;
    const int Nqk = g->Nq[2];
    const int nhg = 3 * ngl;
    int f0, f1;
    face_range(g, direction, &f0, &f1);
#pragma omp parallel for schedule(static)
    for (int64_t eI = 0; eI < nelems; ++eI) {
        const int64_t e = elems[eI] - 1;
        double QM[ORC_MAXS], auxM[ORC_MAXS * 2], GM[ORC_MAXS], nGM[3 * ORC_MAXS];
        double QP[ORC_MAXS], auxP[ORC_MAXS * 2], GP[ORC_MAXS];
        double lgf[ORC_MAXS], tg[3 * ORC_MAXS], visc[ORC_MAXS];
        double Q1[ORC_MAXS], aux1[ORC_MAXS * 2];
        for (int f = f0; f < f1; ++f) {
            const int npts = face_npts(g, f);
            for (int n = 0; n < npts; ++n) {
                face_pt fp;
                face_setup(g, e, f, n, &fp);
                loadv(QM, Q, fp.vidM, ns, fp.eM, Np);
                loadv(auxM, aux, fp.vidM, naux, fp.eM, Np);
                fillnz(GM, ngrad);
                ph->gradient_argument(ph->p, GM, QM, auxM, t);
                loadv(QP, Q, fp.vidP, ns, fp.eP, Np);
                loadv(auxP, aux, fp.vidP, naux, fp.eP, Np);
                fillnz(GP, ngrad);
                ph->gradient_argument(ph->p, GP, QP, auxP, t);
                fillnz(lgf, ngf);
                if (fp.bctag == 0) {
                    /* CentralNumericalFluxGradient  NumericalFluxes.jl:67-83 */
                    for (int s = 0; s < ngrad; ++s)
                        for (int d = 0; d < 3; ++d) tg[d + 3 * s] = fp.n[d] * (GP[s] + GM[s]) / 2;
                    if (ngf > 0) ph->gradient_flux(ph->p, lgf, tg, QM, auxM, t);
                } else {
                    memset(Q1, 0, sizeof(Q1));
                    memset(aux1, 0, sizeof(aux1));
                    if (f == 4) {
                        loadv(Q1, Q, n + Nqk * Nqk, ns, fp.eM, Np);
                        loadv(aux1, aux, n + Nqk * Nqk, naux, fp.eM, Np);
                    }
                    /* numerical_boundary_flux_gradient!  NumericalFluxes.jl:85-123 */
                    ph->boundary_state(ph->p, ORC_BS_GRADIENT, fp.bctag, QP, auxP, fp.n, QM, auxM, t,
                                       Q1, aux1);
                    ph->gradient_argument(ph->p, GP, QP, auxP, t);
                    for (int s = 0; s < ngrad; ++s)
                        for (int d = 0; d < 3; ++d) tg[d + 3 * s] = fp.n[d] * GP[s];
                    if (ngf > 0) ph->gradient_flux(ph->p, lgf, tg, QM, auxM, t);
                }
                for (int s = 0; s < ngrad; ++s)
                    for (int d = 0; d < 3; ++d) nGM[d + 3 * s] = fp.n[d] * GM[s];
                for (int s = 0; s < ngl; ++s) {
                    const int j = ph->hv_indexmap[s];
                    for (int d = 0; d < 3; ++d)
                        hypgrad[fp.vidM + (int64_t)Np * (3 * s + d + (int64_t)nhg * fp.eM)] +=
                            fp.vMI * fp.sM * (tg[d + 3 * j] - nGM[d + 3 * j]);
                }
                if (ngf > 0) {
                    memset(visc, 0, sizeof(double) * ngf);
                    ph->gradient_flux(ph->p, visc, nGM, QM, auxM, t);
                    for (int s = 0; s < ngf; ++s)
                        gf[fp.vidM + (int64_t)Np * (s + (int64_t)ngf * fp.eM)] +=
                            fp.vMI * fp.sM * (lgf[s] - visc[s]);
                }
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* volume_divergence_of_gradients!  DGModel_kernels.jl:2132-2224 / :2226-2329 */
void orc_volume_divergence_of_gradients(const orc_physics *ph, const orc_grid *g, int direction,
                                        const double *hypgrad, double *hypdiv, int increment)
{
    const int Np = g->Np, ngl = ph->ngl, nhyp = ph->nhyp;
    const int Nq1 = g->Nq[0], Nq2 = g->Nq[1], Nq3 = g->Nq[2];
    const double *Dh = g->D[0], *Dv = g->D[2];
    const int nhg = 3 * ngl;
#pragma omp parallel
    {
        double *ld = (double *)malloc(sizeof(double) * Np * ngl);
        double *sg = (double *)malloc(sizeof(double) * 2 * Nq1 * Nq2 * ngl);
        double lg[ORC_MAXS];
#pragma omp for schedule(static)
        for (int64_t e = 0; e < g->nreal; ++e) {
            for (int q = 0; q < Np * ngl; ++q) ld[q] = 0.0;
            if (direction != ORC_VERTICAL) {
                for (int k = 0; k < Nq3; ++k) {
                    for (int j = 0; j < Nq2; ++j)
                        for (int i = 0; i < Nq1; ++i) {
                            const int ijk = i + Nq1 * (j + Nq2 * k);
                            const double M = VG(g, ijk, VM, e);
                            const double x11 = VG(g, ijk, XI1X1, e), x12 = VG(g, ijk, XI1X2, e),
                                         x13 = VG(g, ijk, XI1X3, e);
                            const double x21 = VG(g, ijk, XI2X1, e), x22 = VG(g, ijk, XI2X2, e),
                                         x23 = VG(g, ijk, XI2X3, e);
                            for (int s = 0; s < ngl; ++s) {
                                const double G1 = hypgrad[ijk + (int64_t)Np * (3 * s + 0 + (int64_t)nhg * e)];
                                const double G2 = hypgrad[ijk + (int64_t)Np * (3 * s + 1 + (int64_t)nhg * e)];
                                const double G3 = hypgrad[ijk + (int64_t)Np * (3 * s + 2 + (int64_t)nhg * e)];
                                sg[((size_t)ngl * (i + Nq1 * j) + s) * 2 + 0] =
                                    M * (x11 * G1 + x12 * G2 + x13 * G3);
                                sg[((size_t)ngl * (i + Nq1 * j) + s) * 2 + 1] =
                                    M * (x21 * G1 + x22 * G2 + x23 * G3);
                            }
                        }
                    for (int j = 0; j < Nq2; ++j)
                        for (int i = 0; i < Nq1; ++i) {
                            const int ijk = i + Nq1 * (j + Nq2 * k);
                            const double MI = VG(g, ijk, VMI, e);
                            for (int s = 0; s < ngl; ++s)
                                for (int n = 0; n < Nq1; ++n) {
                                    ld[ijk + Np * s] -= MI * Dh[n + Nq1 * i] *
                                                        sg[((size_t)ngl * (n + Nq1 * j) + s) * 2 + 0];
                                    ld[ijk + Np * s] -= MI * Dh[n + Nq1 * j] *
                                                        sg[((size_t)ngl * (i + Nq1 * n) + s) * 2 + 1];
                                }
                        }
                }
            } else {
                for (int j = 0; j < Nq2; ++j)
                    for (int i = 0; i < Nq1; ++i)
                        for (int k = 0; k < Nq3; ++k) {
                            const int ijk = i + Nq1 * (j + Nq2 * k);
                            const double M = VG(g, ijk, VM, e);
                            const double z1 = VG(g, ijk, XI3X1, e), z2 = VG(g, ijk, XI3X2, e),
                                         z3 = VG(g, ijk, XI3X3, e);
                            for (int s = 0; s < ngl; ++s) {
                                const double G1 = hypgrad[ijk + (int64_t)Np * (3 * s + 0 + (int64_t)nhg * e)];
                                const double G2 = hypgrad[ijk + (int64_t)Np * (3 * s + 1 + (int64_t)nhg * e)];
                                const double G3 = hypgrad[ijk + (int64_t)Np * (3 * s + 2 + (int64_t)nhg * e)];
                                lg[s] = M * (z1 * G1 + z2 * G2 + z3 * G3);
                            }
                            for (int n = 0; n < Nq3; ++n) {
                                const int ijn = i + Nq1 * (j + Nq2 * n);
                                const double MI = VG(g, ijn, VMI, e);
                                for (int s = 0; s < ngl; ++s)
                                    ld[ijn + Np * s] -= MI * Dv[k + Nq3 * n] * lg[s];
                            }
                        }
            }
            /* Qhypervisc_div has nhyp columns, the first ngl are used (DGModel.jl:37-40) */
            for (int s = 0; s < ngl; ++s)
                for (int ijk = 0; ijk < Np; ++ijk) {
                    double *o = &hypdiv[ijk + (int64_t)Np * (s + (int64_t)nhyp * e)];
                    if (increment)
                        *o += ld[ijk + Np * s];
                    else
                        *o = ld[ijk + Np * s];
                }
        }
        free(ld);
        free(sg);
    }
}

/* interface_divergence_of_gradients!  DGModel_kernels.jl:2360-2494 */
void orc_interface_divergence_of_gradients(const orc_physics *ph, const orc_grid *g,
                                           int direction, const double *hypgrad, double *hypdiv,
                                           const double *aux, double t, const int64_t *elems,
                                           int64_t nelems)
{
    const int Np = g->Np, ngl = ph->ngl, nhyp = ph->nhyp, naux = ph->naux;
    const int nhg = 3 * ngl;
    int f0, f1;
    face_range(g, direction, &f0, &f1);
#pragma omp parallel for schedule(static)
    for (int64_t eI = 0; eI < nelems; ++eI) {
        const int64_t e = elems[eI] - 1;
        double gM[3 * ORC_MAXS], gP[3 * ORC_MAXS], ldiv[ORC_MAXS], auxM[ORC_MAXS * 2],
            auxP[ORC_MAXS * 2];
        for (int f = f0; f < f1; ++f) {
            const int npts = face_npts(g, f);
            for (int n = 0; n < npts; ++n) {
                face_pt fp;
                face_setup(g, e, f, n, &fp);
                for (int q = 0; q < nhg; ++q) {
                    gM[q] = hypgrad[fp.vidM + (int64_t)Np * (q + (int64_t)nhg * fp.eM)];
                    gP[q] = hypgrad[fp.vidP + (int64_t)Np * (q + (int64_t)nhg * fp.eP)];
                }
                if (fp.bctag != 0) {
                    loadv(auxM, aux, fp.vidM, naux, fp.eM, Np);
                    loadv(auxP, aux, fp.vidP, naux, fp.eP, Np);
                    /* numerical_boundary_flux_divergence!  NumericalFluxes.jl:732-763 */
                    ph->boundary_state_divergence(ph->p, fp.bctag, gP, auxP, fp.n, gM, auxM, t);
                }
                /* CentralNumericalFluxDivergence  NumericalFluxes.jl:720-730 */
                const double nh[3] = {fp.n[0] / 2, fp.n[1] / 2, fp.n[2] / 2};
                for (int s = 0; s < ngl; ++s)
                    ldiv[s] = (gP[3 * s] + gM[3 * s]) * nh[0] + (gP[3 * s + 1] + gM[3 * s + 1]) * nh[1] +
                              (gP[3 * s + 2] + gM[3 * s + 2]) * nh[2];
                for (int s = 0; s < ngl; ++s)
                    hypdiv[fp.vidM + (int64_t)Np * (s + (int64_t)nhyp * fp.eM)] +=
                        fp.vMI * fp.sM * ldiv[s];
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* volume_gradients_of_laplacians!  DGModel_kernels.jl:2525-2672 / :2674-2824 */
void orc_volume_gradients_of_laplacians(const orc_physics *ph, const orc_grid *g, int direction,
                                        double *hypgrad, const double *hypdiv, const double *Q,
                                        const double *aux, double t, int increment)
{
    const int ns = ph->ns, Np = g->Np, ngl = ph->ngl, nhyp = ph->nhyp;
    const int Nq1 = g->Nq[0], Nq2 = g->Nq[1], Nq3 = g->Nq[2];
    const double *Dh = g->D[0], *Dv = g->D[2];
    /* NB: Qhypervisc_grad is (Np, max(3*ngl, nhyp)) -- create_states.jl:22-26 gives 3*ngl
       columns for GradientLaplacian(); the hyperdiffusive write below uses nhyp columns of
       the same storage, so both strides are 3*ngl (== nhyp for every law in scope). */
    const int nhg = 3 * ngl;
#pragma omp parallel
    {
        double *lgl = (double *)malloc(sizeof(double) * 3 * ngl * Np);
        double *lz = (double *)malloc(sizeof(double) * ngl * Nq3);
        double lQ[ORC_MAXS], laux[ORC_MAXS * 2], lh[ORC_MAXS];
#pragma omp for schedule(static)
        for (int64_t e = 0; e < g->nreal; ++e) {
            fillnz(lgl, 3 * ngl * Np);
#define LAP(ijk_, s_) hypdiv[(ijk_) + (int64_t)Np * ((s_) + (int64_t)nhyp * e)]
            if (direction != ORC_VERTICAL) {
                for (int k = 0; k < Nq3; ++k)
                    for (int j = 0; j < Nq2; ++j)
                        for (int i = 0; i < Nq1; ++i) {
                            const int ijk = i + Nq1 * (j + Nq2 * k);
                            const double x11 = VG(g, ijk, XI1X1, e), x12 = VG(g, ijk, XI1X2, e),
                                         x13 = VG(g, ijk, XI1X3, e);
                            const double x21 = VG(g, ijk, XI2X1, e), x22 = VG(g, ijk, XI2X2, e),
                                         x23 = VG(g, ijk, XI2X3, e);
                            for (int s = 0; s < ngl; ++s) {
                                double l1 = 0.0, l2 = 0.0;
                                for (int n = 0; n < Nq1; ++n) {
                                    l1 += Dh[i + Nq1 * n] * LAP(n + Nq1 * (j + Nq2 * k), s);
                                    l2 += Dh[j + Nq1 * n] * LAP(i + Nq1 * (n + Nq2 * k), s);
                                }
                                double *l = lgl + 3 * (s + ngl * ijk);
                                l[0] = x11 * l1;
                                l[1] = x12 * l1;
                                l[2] = x13 * l1;
                                l[0] += x21 * l2;
                                l[1] += x22 * l2;
                                l[2] += x23 * l2;
                            }
                        }
            } else {
                for (int j = 0; j < Nq2; ++j)
                    for (int i = 0; i < Nq1; ++i) {
                        fillnz(lz, ngl * Nq3);
                        for (int k = 0; k < Nq3; ++k)
                            for (int s = 0; s < ngl; ++s)
                                for (int n = 0; n < Nq3; ++n)
                                    lz[s + ngl * n] +=
                                        Dv[n + Nq3 * k] * LAP(i + Nq1 * (j + Nq2 * k), s);
                        for (int k = 0; k < Nq3; ++k) {
                            const int ijk = i + Nq1 * (j + Nq2 * k);
                            const double z1 = VG(g, ijk, XI3X1, e), z2 = VG(g, ijk, XI3X2, e),
                                         z3 = VG(g, ijk, XI3X3, e);
                            for (int s = 0; s < ngl; ++s) {
                                double *l = lgl + 3 * (s + ngl * ijk);
                                l[0] += z1 * lz[s + ngl * k];
                                l[1] += z2 * lz[s + ngl * k];
                                l[2] += z3 * lz[s + ngl * k];
                            }
                        }
                    }
            }
#undef LAP
            for (int ijk = 0; ijk < Np; ++ijk) {
                loadv(lQ, Q, ijk, ns, e, Np);
                loadv(laux, aux, ijk, ph->naux, e, Np);
                fillnz(lh, nhyp);
                ph->post_gradient_laplacian(ph->p, lh, lgl + 3 * ngl * ijk, lQ, laux, t);
                for (int s = 0; s < nhyp; ++s) {
                    double *o = &hypgrad[ijk + (int64_t)Np * (s + (int64_t)nhg * e)];
                    if (increment)
                        *o += lh[s];
                    else
                        *o = lh[s];
                }
            }
        }
        free(lgl);
        free(lz);
    }
}

/* interface_gradients_of_laplacians!  DGModel_kernels.jl:2859-3026 */
void orc_interface_gradients_of_laplacians(const orc_physics *ph, const orc_grid *g,
                                           int direction, double *hypgrad, const double *hypdiv,
                                           const double *Q, const double *aux, double t,
                                           const int64_t *elems, int64_t nelems)
{
    const int ns = ph->ns, Np = g->Np, ngl = ph->ngl, nhyp = ph->nhyp, naux = ph->naux;
    const int nhg = 3 * ngl;
    int f0, f1;
    face_range(g, direction, &f0, &f1);
#pragma omp parallel for schedule(static)
    for (int64_t eI = 0; eI < nelems; ++eI) {
        const int64_t e = elems[eI] - 1;
        double lapM[ORC_MAXS], lapP[ORC_MAXS], lh[ORC_MAXS], G[3 * ORC_MAXS];
        double QM[ORC_MAXS], auxM[ORC_MAXS * 2], QP[ORC_MAXS], auxP[ORC_MAXS * 2];
        for (int f = f0; f < f1; ++f) {
            const int npts = face_npts(g, f);
            for (int n = 0; n < npts; ++n) {
                face_pt fp;
                face_setup(g, e, f, n, &fp);
                loadv(QM, Q, fp.vidM, ns, fp.eM, Np);
                loadv(auxM, aux, fp.vidM, naux, fp.eM, Np);
                loadv(QP, Q, fp.vidP, ns, fp.eP, Np);
                loadv(auxP, aux, fp.vidP, naux, fp.eP, Np);
                for (int s = 0; s < ngl; ++s) {
                    lapM[s] = hypdiv[fp.vidM + (int64_t)Np * (s + (int64_t)nhyp * fp.eM)];
                    lapP[s] = hypdiv[fp.vidP + (int64_t)Np * (s + (int64_t)nhyp * fp.eP)];
                }
                if (fp.bctag != 0) /* NumericalFluxes.jl:792-832 */
                    ph->boundary_state_higher_order(ph->p, fp.bctag, QP, auxP, lapP, fp.n, QM, auxM,
                                                    lapM, t);
                /* CentralNumericalFluxHigherOrder  NumericalFluxes.jl:768-790 */
                for (int s = 0; s < ngl; ++s)
                    for (int d = 0; d < 3; ++d) G[d + 3 * s] = fp.n[d] * (lapP[s] - lapM[s]) / 2;
                memset(lh, 0, sizeof(double) * nhyp);
                ph->post_gradient_laplacian(ph->p, lh, G, QM, auxM, t);
                for (int s = 0; s < nhyp; ++s)
                    hypgrad[fp.vidM + (int64_t)Np * (s + (int64_t)nhg * fp.eM)] +=
                        fp.vMI * fp.sM * lh[s];
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* kernel_nodal_update_auxiliary_state!  DGModel_kernels.jl:1769-1825 */
void orc_update_auxiliary_state(const orc_physics *ph, const orc_grid *g, const double *Q,
                                double *aux, double t, int64_t e0, int64_t e1,
                                const uint8_t *activedofs)
{
    if (!ph->update_aux) return;
    const int Np = g->Np, ns = ph->ns, naux = ph->naux;
#pragma omp parallel for schedule(static)
    for (int64_t e = e0; e < e1; ++e) {
        double lQ[ORC_MAXS], laux[ORC_MAXS * 2];
        for (int n = 0; n < Np; ++n) {
            if (!activedofs[n + e * Np]) continue;
            loadv(lQ, Q, n, ns, e, Np);
            loadv(laux, aux, n, naux, e, Np);
            ph->update_aux(ph->p, lQ, laux, t);
            for (int s = 0; s < naux; ++s) aux[n + (int64_t)Np * (s + (int64_t)naux * e)] = laux[s];
        }
    }
}

/* update!  LowStorageRungeKuttaMethod.jl:146-158 */
void orc_lsrk_update(double *dQ, double *Q, double rka, double rkb, double dt, int64_t n)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        Q[i] += rkb * dt * dQ[i];
        dQ[i] *= rka;
    }
}

/* kernel_fillsendbuf! / kernel_transferrecvbuf!  MPIStateArrays.jl:837-871 */
void orc_fillsendbuf(double *sendbuf, const double *buf, const int64_t *vmapsend, int64_t nvmap,
                     int Np, int nvar)
{
    for (int64_t i = 0; i < nvmap; ++i) {
        const int64_t e = (vmapsend[i] - 1) / Np, n = (vmapsend[i] - 1) % Np;
        for (int s = 0; s < nvar; ++s)
            sendbuf[s + (int64_t)nvar * i] = buf[n + (int64_t)Np * (s + (int64_t)nvar * e)];
    }
}
void orc_transferrecvbuf(double *buf, const double *recvbuf, const int64_t *vmaprecv,
                         int64_t nvmap, int Np, int nvar)
{
    for (int64_t i = 0; i < nvmap; ++i) {
        const int64_t e = (vmaprecv[i] - 1) / Np, n = (vmaprecv[i] - 1) % Np;
        for (int s = 0; s < nvar; ++s)
            buf[n + (int64_t)Np * (s + (int64_t)nvar * e)] = recvbuf[s + (int64_t)nvar * i];
    }
}

/* ---- Courant number (SpaceDiscretization.jl:307-365) --------------------------------- */
static inline double dist3(const orc_grid *g, int a, int b, int64_t e)
{
    const double d0 = VG(g, a, 12, e) - VG(g, b, 12, e); /* _x1.._x3 = columns 13..15 */
    const double d1 = VG(g, a, 13, e) - VG(g, b, 13, e);
    const double d2 = VG(g, a, 14, e) - VG(g, b, 14, e);
    return sqrt(d0 * d0 + d1 * d1 + d2 * d2);
}

void orc_min_neighbor_distance(const orc_grid *g, int direction, double *out)
{
    const int Nq1 = g->Nq[0], Nq2 = g->Nq[1], Nqk = g->Nq[2], Np = g->Np;
    const int m1 = direction != ORC_VERTICAL, m2 = direction != ORC_VERTICAL,
              m3 = direction != ORC_HORIZONTAL;
#pragma omp parallel for
    for (int64_t e = 0; e < g->nreal; ++e)
        for (int k = 0; k < Nqk; ++k)
            for (int j = 0; j < Nq2; ++j)
                for (int i = 0; i < Nq1; ++i) {
                    const int ijk = i + Nq1 * (j + Nq2 * k);
                    double md = INFINITY;
                    if (m1)
                        for (int ii = i - 1; ii <= i + 1; ii += 2)
                            if (ii >= 0 && ii < Nq1)
                                md = fmin(md, dist3(g, ijk, ii + Nq1 * (j + Nq2 * k), e));
                    if (m2)
                        for (int jj = j - 1; jj <= j + 1; jj += 2)
                            if (jj >= 0 && jj < Nq2)
                                md = fmin(md, dist3(g, ijk, i + Nq1 * (jj + Nq2 * k), e));
                    if (m3)
                        for (int kk = k - 1; kk <= k + 1; kk += 2)
                            if (kk >= 0 && kk < Nqk)
                                md = fmin(md, dist3(g, ijk, i + Nq1 * (j + Nq2 * kk), e));
                    out[ijk + (int64_t)Np * e] = md;
                }
}

void orc_local_courant(const orc_physics *ph, const orc_grid *g, int kind, double *pointwise,
                       const double *Q, const double *aux, const double *gf, double dt,
                       double simtime, int direction)
{
    const int Np = g->Np;
#pragma omp parallel for
    for (int64_t e = 0; e < g->nreal; ++e)
        for (int n = 0; n < Np; ++n) {
            double lQ[ORC_MAXS], lA[ORC_MAXS], lG[ORC_MAXS];
            loadv(lQ, Q, n, ph->ns, e, Np);
            loadv(lA, aux, n, ph->naux, e, Np);
            loadv(lG, gf, n, ph->ngf, e, Np);
            const double dx = pointwise[n + (int64_t)Np * e];
            pointwise[n + (int64_t)Np * e] =
                ph->courant(ph->p, kind, lQ, lA, lG, dx, dt, simtime, direction);
        }
}
