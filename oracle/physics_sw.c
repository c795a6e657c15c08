/* ORACLE -- TEST INFRASTRUCTURE ONLY (see dg_oracle.h).
 *
 * ShallowWaterModel (src/Ocean/ShallowWater/ShallowWaterModel.jl): state eta, U[2]; auxiliary
 * y, G_U[2], Delta_u[2]; gradient U[2]; gradient flux nu grad U (3 x 2).  Restated: :87-93
 * (state), :107-113 (aux), :146-156, :178-191 (gradient argument / flux, ConstantViscosity),
 * :193-233 (first-order flux, advective flux), :246-258 (second-order flux), :260 (wavespeed),
 * :262-285 (source: Coriolis, forcing_term!, linear drag), and the Coupled forcing
 * src/Ocean/SplitExplicit/ShallowWaterCoupling.jl:3-7.  The barotropic model of the
 * split-explicit ocean runs on a one-layer extrusion of the 2-D grid (see oracle.py).
 * Pinned by test/Ocean/SplitExplicit/test_spindown_long.jl + hydrostatic_spindown_refvals.jl.
 *
 * iparam[0]=advection [1]=turbulence (0 ConstantViscosity, 1 LinearDrag) [2]=Coriolis kind
 * (0 fixed box, 1 rotating, 2 beta plane) [3]=coupling; dparam[0..6] = grav H c nu_or_lambda f_o
 * beta (kinematic stress of the uncoupled SimpleBox problem is -0)
 */
#include <math.h>
#include <stdlib.h>

#include "dg_oracle.h"

typedef struct {
    int adv, drag, cor, coupled;
    double grav, H, c, nu, fo, beta;
} sw_t;
enum { ETA = 0, U1 = 1, U2 = 2 };
enum { AY = 0, AG = 1, ADU = 3 };

static void sw_flux1(const void *p_, double *F, const double *Q, const double *aux, double t, int dir)
{
    const sw_t *m = (const sw_t *)p_;
    (void)aux; (void)t; (void)dir;
    const double Uv[3] = {Q[U1], Q[U2], -0.0};
    static const double Ih[3][2] = {{1, -0.0}, {-0.0, 1}, {-0.0, -0.0}};
    for (int d = 0; d < 3; ++d) F[d + 3 * ETA] += Uv[d];
    const double ghe = m->grav * m->H * Q[ETA];
    for (int c = 0; c < 2; ++c)
        for (int d = 0; d < 3; ++d) F[d + 3 * (U1 + c)] += ghe * Ih[d][c];
    if (m->adv) {
        const double Hinv = 1 / m->H;
        for (int c = 0; c < 2; ++c)
            for (int d = 0; d < 3; ++d) F[d + 3 * (U1 + c)] += Hinv * Uv[d] * Q[U1 + c];
    }
}
static void sw_flux2(const void *p_, double *F, const double *Q, const double *gf, const double *hyp,
                     const double *aux, double t)
{
    const sw_t *m = (const sw_t *)p_;
    (void)Q; (void)hyp; (void)aux; (void)t;
    if (m->drag) return;
    for (int q = 0; q < 6; ++q) F[3 * U1 + q] += gf[q];
}
static void sw_source(const void *p_, double *S, const double *Q, const double *gf, const double *aux,
                      double t, int dir)
{
    const sw_t *m = (const sw_t *)p_;
    (void)gf; (void)t; (void)dir;
    const double f = m->cor == 0 ? -0.0 : (m->cor == 1 ? m->fo : m->fo + m->beta * aux[AY]);
    S[U1] -= -f * Q[U2];
    S[U2] -= f * Q[U1];
    if (m->coupled) { /* forcing_term!(::Coupled): S.U += A.G_U */
        S[U1] += aux[AG];
        S[U2] += aux[AG + 1];
    } else { /* kinematic_stress(::SimpleBox, y) = [-0, -0] */
        S[U1] += -0.0;
        S[U2] += -0.0;
    }
    if (m->drag) {
        S[U1] -= m->nu * Q[U1];
        S[U2] -= m->nu * Q[U2];
    }
}
static void sw_garg(const void *p_, double *G, const double *Q, const double *aux, double t)
{
    const sw_t *m = (const sw_t *)p_;
    (void)aux; (void)t;
    if (m->drag) return;
    G[0] = Q[U1];
    G[1] = Q[U2];
}
static void sw_gflux(const void *p_, double *D, const double *g, const double *Q, const double *aux,
                     double t)
{
    const sw_t *m = (const sw_t *)p_;
    (void)Q; (void)aux; (void)t;
    if (m->drag) return;
    const double nu[3] = {m->nu, m->nu, -0.0};
    for (int c = 0; c < 2; ++c)
        for (int d = 0; d < 3; ++d) D[d + 3 * c] = -nu[d] * g[d + 3 * c];
}
static void sw_none6(const void *p, double *h, const double *gl, const double *Q, const double *aux,
                     double t)
{
    (void)p; (void)h; (void)gl; (void)Q; (void)aux; (void)t;
}
static void sw_ws(const void *p_, double *ws, const double *n, const double *Q, const double *aux,
                  double t, int fd)
{
    const sw_t *m = (const sw_t *)p_;
    (void)n; (void)Q; (void)aux; (void)t; (void)fd;
    ws[0] = ws[1] = ws[2] = m->c;
}
static void sw_bstate(const void *p, int kind, int bctag, double *QP, double *auxP, const double *n,
                      const double *QM, const double *auxM, double t, const double *Q1,
                      const double *aux1)
{ /* periodic boxes only: no boundary tags in scope */
    (void)p; (void)kind; (void)bctag; (void)QP; (void)auxP; (void)n; (void)QM; (void)auxM; (void)t;
    (void)Q1; (void)aux1;
}
static void sw_bflux2(const void *p, int bctag, double *F, double *QP, double *gfP, double *hypP,
                      double *auxP, const double *n, const double *QM, const double *gfM,
                      const double *hypM, const double *auxM, double t, const double *Q1,
                      const double *gf1, const double *aux1)
{
    (void)p; (void)bctag; (void)F; (void)QP; (void)gfP; (void)hypP; (void)auxP; (void)n; (void)QM;
    (void)gfM; (void)hypM; (void)auxM; (void)t; (void)Q1; (void)gf1; (void)aux1;
}
static void sw_bdiv(const void *p, int b, double *gP, double *aP, const double *n, const double *gM,
                    const double *aM, double t)
{
    (void)p; (void)b; (void)gP; (void)aP; (void)n; (void)gM; (void)aM; (void)t;
}
static void sw_bhigher(const void *p, int b, double *QP, double *aP, double *lP, const double *n,
                       const double *QM, const double *aM, const double *lM, double t)
{
    (void)p; (void)b; (void)QP; (void)aP; (void)lP; (void)n; (void)QM; (void)aM; (void)lM; (void)t;
}

orc_physics *orc_sw_new(const int *ip, const double *dp, int nf_first)
{
    orc_physics *ph = (orc_physics *)calloc(1, sizeof(orc_physics));
    sw_t *m = (sw_t *)calloc(1, sizeof(sw_t));
    m->adv = ip[0]; m->drag = ip[1]; m->cor = ip[2]; m->coupled = ip[3];
    m->grav = dp[0]; m->H = dp[1]; m->c = dp[2]; m->nu = dp[3]; m->fo = dp[4]; m->beta = dp[5];
    ph->ns = 3;
    ph->naux = 5;
    ph->ngrad = m->drag ? 0 : 2;
    ph->ngf = m->drag ? 0 : 6;
    ph->nf_first = nf_first;
    ph->p = m;
    ph->flux_first_order = sw_flux1;
    ph->flux_second_order = sw_flux2;
    ph->source = sw_source;
    ph->gradient_argument = sw_garg;
    ph->gradient_flux = sw_gflux;
    ph->post_gradient_laplacian = sw_none6;
    ph->wavespeed = sw_ws;
    ph->boundary_state = sw_bstate;
    ph->boundary_flux_second_order = sw_bflux2;
    ph->boundary_state_divergence = sw_bdiv;
    ph->boundary_state_higher_order = sw_bhigher;
    return ph;
}
