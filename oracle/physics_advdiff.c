/* ORACLE -- TEST INFRASTRUCTURE ONLY (see dg_oracle.h).
 *
 * Pointwise physics of the reference's test balance law `AdvectionDiffusion{1}`
 * (test/Numerics/DGMethods/advection_diffusion/advection_diffusion_model.jl:92-617)
 * and of the problems that supply its coefficients / boundary data:
 *   problem 0  Pseudo1D                 pseudo1D_advection_diffusion.jl:28-68
 *   problem 1  ConstantHyperDiffusion   periodic_3D_hyperdiffusion.jl:29-63
 *   problem 2  HyperDiffusionBC         hyperdiffusion_bc.jl (boundary data of that test)
 *   problem 3  HeatEqn (Pseudo1D heat)  pseudo1D_heat_eqn.jl
 *
 * Parameter block (shared *data* contract with the product's descriptor):
 *   iparam[0]=num_equations (1)  [1]=advection [2]=diffusion [3]=hyperdiffusion
 *   iparam[4]=flux_bc [5]=problem [6]=nbc [7..13]=bc bitmask of tag 1..7
 *   bc bit b: InhomogeneousBC{b} for b=0..3, HomogeneousBC{b-4} for b=4..7, bit 8 NoFlowBC
 *   (advection_sphere.jl:118-132)
 *   dparam: problem parameters (see each problem)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "dg_oracle.h"

typedef struct {
    int adv, diff, hyper, flux_bc, problem, nbc;
    int bc[8];
    double d[32];
    int ou, oD, oH; /* aux offsets of u, D, H (coord is 0..2) */
} advdiff_t;

#define BC_INHOM(o) (1 << (o))
#define BC_HOM(o) (1 << ((o) + 4))
#define BC_ANY(o) (BC_INHOM(o) | BC_HOM(o))

/* ---- problems ---------------------------------------------------- */
/* Pseudo1D{n, alpha, beta, mu, delta}: d[0..2]=n, d[3]=alpha, d[4]=beta, d[5]=mu, d[6]=delta */
static double p1d_rho(const advdiff_t *m, const double *x, double t)
{
    const double *n = m->d, al = m->d[3], be = m->d[4], mu = m->d[5], de = m->d[6];
    const double xn = n[0] * x[0] + n[1] * x[1] + n[2] * x[2];
    const double a = xn - mu - al * t;
    return exp(-(a * a) / (4 * be * (de + t))) / sqrt(1 + t / de);
}
static void p1d_grad(const advdiff_t *m, double *g, const double *x, double t)
{
    const double *n = m->d, al = m->d[3], be = m->d[4], mu = m->d[5], de = m->d[6];
    const double xn = n[0] * x[0] + n[1] * x[1] + n[2] * x[2];
    const double a = xn - mu - al * t;
    for (int i = 0; i < 3; ++i)
        g[i] = -(2 * n[i] * a / (4 * be * (de + t)) * exp(-(a * a) / (4 * be * (de + t))) /
                 sqrt(1 + t / de));
}
/* ConstantHyperDiffusion{dim, dir}: d[0..8]=D (column-major), d[9]=dim, d[10]=dir */
static double chd_rho(const advdiff_t *m, const double *x, double t)
{
    const int dim = (int)m->d[9], dir = (int)m->d[10];
    const double k[3] = {1, 2, 3};
    double c = 0;
    if (dir == ORC_EVERY || dir == ORC_HORIZONTAL) {
        const int dd = dir == ORC_EVERY ? dim : dim - 1;
        double s2 = 0, skd = 0;
        for (int i = 0; i < dd; ++i) s2 += k[i] * k[i];
        for (int j = 0; j < dd; ++j)
            for (int i = 0; i < dd; ++i) skd += k[i] * k[j] * m->d[i + 3 * j];
        c = s2 * skd;
    } else {
        c = k[dim - 1] * k[dim - 1] * (k[dim - 1] * k[dim - 1] * m->d[(dim - 1) + 3 * (dim - 1)]);
    }
    double kx = 0;
    for (int i = 0; i < dim; ++i) kx += k[i] * x[i];
    return sin(kx) * exp(-c * t);
}

/* ConstantHyperDiffusion{mu, k} of hyperdiffusion_bc.jl:25-112: d[0]=mu, d[1..3]=k */
static double hbc_e(const advdiff_t *m, double t)
{
    const double *k = m->d + 1;
    const double k2 = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
    return exp(-(k2 * k2) * m->d[0] * t);
}
static double hbc_rho(const advdiff_t *m, const double *x, double t)
{
    const double *k = m->d + 1;
    return cos(k[0] * x[0]) * cos(k[1] * x[1]) * cos(k[2] * x[2]) * hbc_e(m, t);
}
static void hbc_sincos(const advdiff_t *m, const double *x, double *v)
{
    const double *k = m->d + 1;
    v[0] = k[0] * sin(k[0] * x[0]) * cos(k[1] * x[1]) * cos(k[2] * x[2]);
    v[1] = k[1] * cos(k[0] * x[0]) * sin(k[1] * x[1]) * cos(k[2] * x[2]);
    v[2] = k[2] * cos(k[0] * x[0]) * cos(k[1] * x[1]) * sin(k[2] * x[2]);
}

/* HeatEqn{n, kappa, A} of pseudo1D_heat_eqn.jl:28-53: d[0..2]=n, d[3]=kappa, d[4]=A */
static double heat_rho(const advdiff_t *m, const double *x, double t)
{
    const double *n = m->d, ka = m->d[3], A = m->d[4];
    const double xn = n[0] * x[0] + n[1] * x[1] + n[2] * x[2];
    return xn + A * cos(ka * xn) * exp(-(ka * ka) * t);
}
static void heat_grad(const advdiff_t *m, double *g, const double *x, double t)
{ /* the exact gradient of normal_boundary_flux_second_order! (:79-88) */
    const double *n = m->d, ka = m->d[3], A = m->d[4];
    const double xn = n[0] * x[0] + n[1] * x[1] + n[2] * x[2];
    for (int i = 0; i < 3; ++i) g[i] = n[i] * (1 - A * ka * sin(ka * xn) * exp(-(ka * ka) * t));
}

static double problem_rho(const advdiff_t *m, const double *x, double t)
{
    switch (m->problem) {
    case 3: return heat_rho(m, x, t);
    case 0: return p1d_rho(m, x, t);
    case 1: return chd_rho(m, x, t);
    case 2: return hbc_rho(m, x, t);
    default: return 0.0;
    }
}
/* inhomogeneous_data!(Val(2), ...) and Val(3) (hyperdiffusion_bc.jl:80-112) */
static double problem_lap(const advdiff_t *m, const double *x, double t)
{
    if (m->problem != 2) return 0.0;
    const double *k = m->d + 1;
    const double k2 = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
    return -k2 * cos(k[0] * x[0]) * cos(k[1] * x[1]) * cos(k[2] * x[2]) * hbc_e(m, t);
}
static void problem_gradlap(const advdiff_t *m, double *g, const double *x, double t)
{
    g[0] = g[1] = g[2] = 0.0;
    if (m->problem != 2) return;
    const double *k = m->d + 1;
    const double k2 = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
    double v[3];
    hbc_sincos(m, x, v);
    const double e = hbc_e(m, t);
    for (int i = 0; i < 3; ++i) g[i] = (k2 * v[i]) * e;
}
static void problem_grad(const advdiff_t *m, double *g, const double *x, double t)
{
    switch (m->problem) {
    case 0: p1d_grad(m, g, x, t); break;
    case 3: heat_grad(m, g, x, t); break;
    case 2: { /* inhomogeneous_data!(Val(1), ::ConstantHyperDiffusion, ...)  hyperdiffusion_bc.jl:63-79 */
        double v[3];
        hbc_sincos(m, x, v);
        const double e = hbc_e(m, t);
        for (int i = 0; i < 3; ++i) g[i] = -v[i] * e;
        break;
    }
    default: g[0] = g[1] = g[2] = 0.0;
    }
}

/* ---- balance law -------------------------------------------------- */
static void ad_flux1(const void *p, double *F, const double *Q, const double *aux, double t, int dir)
{
    const advdiff_t *m = (const advdiff_t *)p;
    (void)t;
    (void)dir;
    if (m->adv)
        for (int d = 0; d < 3; ++d) F[d] += aux[m->ou + d] * Q[0];
}
static void ad_flux2(const void *p, double *F, const double *Q, const double *gf, const double *hyp,
                     const double *aux, double t)
{
    const advdiff_t *m = (const advdiff_t *)p;
    (void)Q;
    (void)aux;
    (void)t;
    if (m->diff)
        for (int d = 0; d < 3; ++d) F[d] += -gf[d];
    if (m->hyper)
        for (int d = 0; d < 3; ++d) F[d] += hyp[d];
}
static void ad_source(const void *p, double *S, const double *Q, const double *gf, const double *aux,
                      double t, int dir)
{
    (void)p; (void)S; (void)Q; (void)gf; (void)aux; (void)t; (void)dir;
}
static void ad_gradarg(const void *p, double *G, const double *Q, const double *aux, double t)
{
    const advdiff_t *m = (const advdiff_t *)p;
    (void)aux;
    (void)t;
    if (m->diff || m->hyper) G[0] = Q[0];
}
static void matvec3(double *o, const double *A, const double *v)
{
    for (int i = 0; i < 3; ++i) o[i] = A[i] * v[0] + A[i + 3] * v[1] + A[i + 6] * v[2];
}
static void ad_gradflux(const void *p, double *gf, const double *gradG, const double *Q,
                        const double *aux, double t)
{
    const advdiff_t *m = (const advdiff_t *)p;
    (void)Q;
    (void)t;
    if (m->diff) matvec3(gf, aux + m->oD, gradG);
}
static void ad_postlap(const void *p, double *hyp, const double *gradlap, const double *Q,
                       const double *aux, double t)
{
    const advdiff_t *m = (const advdiff_t *)p;
    (void)Q;
    (void)t;
    if (m->hyper) matvec3(hyp, aux + m->oH, gradlap);
}
static void ad_wavespeed(const void *p, double *ws, const double *n, const double *Q,
                         const double *aux, double t, int facedir)
{
    const advdiff_t *m = (const advdiff_t *)p;
    (void)Q; (void)t; (void)facedir;
    ws[0] = m->adv ? fabs(n[0] * aux[m->ou] + n[1] * aux[m->ou + 1] + n[2] * aux[m->ou + 2]) : 0.0;
}
/* boundary_state!(nf, bcs, m, stateP, auxP, nM, stateM, auxM, t, _...)  :402-428 */
static void ad_bstate(const void *p, int kind, int bctag, double *QP, double *auxP, const double *n,
                      const double *QM, const double *auxM, double t, const double *Q1,
                      const double *aux1)
{
    const advdiff_t *m = (const advdiff_t *)p;
    (void)n; (void)Q1; (void)aux1;
    const int bc = m->bc[bctag - 1];
    if (bc & (1 << 8)) { /* NoFlowBC: boundary_state!(::RusanovNumericalFlux, ...) only */
        if (kind == ORC_BS_FIRST && m->adv)
            for (int d = 0; d < 3; ++d) auxP[m->ou + d] = -auxM[m->ou + d];
        return;
    }
    if (bc & BC_INHOM(0))
        QP[0] = problem_rho(m, auxP, t);
    else if (bc & BC_ANY(1))
        QP[0] = QM[0];
    else if (bc & BC_HOM(0))
        QP[0] = 0.0;
}
/* boundary_state!(nf::CentralNumericalFluxSecondOrder, ...) :430-517 followed by
   flux_second_order! (NumericalFluxes.jl:921-967), or the flux_bc method :519-567 */
static void ad_bflux2(const void *p, int bctag, double *F, double *QP, double *gfP, double *hypP,
                      double *auxP, const double *n, const double *QM, const double *gfM,
                      const double *hypM, const double *auxM, double t, const double *Q1,
                      const double *gf1, const double *aux1)
{
    const advdiff_t *m = (const advdiff_t *)p;
    (void)n; (void)hypM; (void)Q1; (void)gf1; (void)aux1;
    const int bc = m->bc[bctag - 1];
    double g[3];
    if (!m->diff && !m->hyper) {
        if (!m->flux_bc) ad_flux2(p, F, QP, gfP, hypP, auxP, t);
        return;
    }
    if (m->flux_bc) {
        if (bc & BC_ANY(0)) {
            ad_flux2(p, F, QM, gfM, hypM, auxM, t);
        } else if (bc & BC_INHOM(1)) {
            problem_grad(m, g, auxM, t);
            const double *D = auxM + m->oD;
            for (int i = 0; i < 3; ++i) F[i] = -D[i] * g[0] + -D[i + 3] * g[1] + -D[i + 6] * g[2];
        } else if (bc & BC_HOM(1)) {
            F[0] = F[1] = F[2] = 0.0;
        }
        return;
    }
    if (m->diff) {
        if (bc & BC_ANY(0)) {
            for (int d = 0; d < 3; ++d) gfP[d] = gfM[d];
        } else if (bc & BC_INHOM(1)) {
            problem_grad(m, g, auxM, t);
            matvec3(gfP, auxM + m->oD, g);
        } else if (bc & BC_HOM(1)) {
            g[0] = g[1] = g[2] = 0.0;
            matvec3(gfP, auxM + m->oD, g);
        }
    }
    if (m->hyper) {
        if (bc & BC_INHOM(3)) {
            problem_gradlap(m, g, auxM, t);
            matvec3(hypP, auxM + m->oH, g);
        } else if (bc & BC_HOM(3)) {
            g[0] = g[1] = g[2] = 0.0;
            matvec3(hypP, auxM + m->oH, g);
        }
    }
    ad_flux2(p, F, QP, gfP, hypP, auxP, t);
}
/* boundary_state!(::CentralNumericalFluxDivergence, ...) :569-591 */
static void ad_bdiv(const void *p, int bctag, double *gradP, double *auxP, const double *n,
                    const double *gradM, const double *auxM, double t)
{
    const advdiff_t *m = (const advdiff_t *)p;
    (void)auxP; (void)n; (void)gradM;
    if (!m->hyper) return;
    const int bc = m->bc[bctag - 1];
    if (bc & BC_INHOM(1))
        problem_grad(m, gradP, auxM, t);
    else if (bc & BC_HOM(1))
        gradP[0] = gradP[1] = gradP[2] = 0.0;
}
/* boundary_state!(::CentralNumericalFluxHigherOrder, ...) :593-617 */
static void ad_bhigher(const void *p, int bctag, double *QP, double *auxP, double *lapP,
                       const double *n, const double *QM, const double *auxM, const double *lapM,
                       double t)
{
    const advdiff_t *m = (const advdiff_t *)p;
    (void)QP; (void)auxP; (void)n; (void)QM; (void)lapM;
    if (!m->hyper) return;
    const int bc = m->bc[bctag - 1];
    if (bc & BC_INHOM(2))
        lapP[0] = problem_lap(m, auxM, t);
    else if (bc & BC_HOM(2))
        lapP[0] = 0.0;
}

/* update_velocity_diffusion!(::ReversingDeformationalFlow, ...)  advection_sphere.jl:76-101 */
static void ad_update_aux(const void *p, const double *Q, double *aux, double t)
{
    const advdiff_t *m = (const advdiff_t *)p;
    (void)Q;
    const double x = aux[0], y = aux[1], z = aux[2];
    const double r = sqrt(x * x + y * y + z * z);
    const double lam = atan2(y, x), phi = asin(z / r);
    const double T = 5.0;
    const double lamp = lam - 2 * M_PI * t / T;
    const double sl = sin(lamp);
    const double ul = 10 * r / T * (sl * sl) * sin(2 * phi) * cos(M_PI * t / T) +
                      2 * M_PI * r / T * cos(phi);
    const double up = 10 * r / T * sin(2 * lamp) * cos(phi) * cos(M_PI * t / T);
    aux[m->ou + 0] = -ul * sin(lam) - up * cos(lam) * sin(phi);
    aux[m->ou + 1] = +ul * cos(lam) - up * sin(lam) * sin(phi);
    aux[m->ou + 2] = +up * cos(phi);
}

orc_physics *orc_advdiff_new(const int *iparam, const double *dparam, int nf_first)
{
    orc_physics *ph = (orc_physics *)calloc(1, sizeof(orc_physics));
    advdiff_t *m = (advdiff_t *)calloc(1, sizeof(advdiff_t));
    m->adv = iparam[1];
    m->diff = iparam[2];
    m->hyper = iparam[3];
    m->flux_bc = iparam[4];
    m->problem = iparam[5];
    m->nbc = iparam[6];
    for (int i = 0; i < 7; ++i) m->bc[i] = iparam[7 + i];
    memcpy(m->d, dparam, sizeof(m->d));
    int o = 3;
    m->ou = o;
    if (m->adv) o += 3;
    m->oD = o;
    if (m->diff) o += 9;
    m->oH = o;
    if (m->hyper) o += 9;
    ph->ns = 1;
    ph->naux = o;
    ph->ngrad = (m->diff || m->hyper) ? 1 : 0;
    ph->ngf = m->diff ? 3 : 0;
    ph->ngl = m->hyper ? 1 : 0;
    ph->nhyp = m->hyper ? 3 : 0;
    ph->hv_indexmap[0] = 0;
    ph->nf_first = nf_first;
    ph->p = m;
    ph->flux_first_order = ad_flux1;
    ph->flux_second_order = ad_flux2;
    ph->source = ad_source;
    ph->gradient_argument = ad_gradarg;
    ph->gradient_flux = ad_gradflux;
    ph->post_gradient_laplacian = ad_postlap;
    ph->wavespeed = ad_wavespeed;
    ph->boundary_state = ad_bstate;
    ph->boundary_flux_second_order = ad_bflux2;
    ph->boundary_state_divergence = ad_bdiv;
    ph->boundary_state_higher_order = ad_bhigher;
    ph->update_aux = (m->adv && m->problem == 7) ? ad_update_aux : NULL;
    return ph;
}

void orc_physics_free(orc_physics *ph)
{
    if (!ph) return;
    free((void *)ph->p);
    free(ph);
}

/* pointwise evaluation helpers for the python side of the tests */
double orc_advdiff_initial(const orc_physics *ph, const double *x, double t)
{
    return problem_rho((const advdiff_t *)ph->p, x, t);
}
int orc_physics_counts(const orc_physics *ph, int *out)
{
    out[0] = ph->ns; out[1] = ph->naux; out[2] = ph->ngrad; out[3] = ph->ngf;
    out[4] = ph->ngl; out[5] = ph->nhyp;
    return 0;
}
