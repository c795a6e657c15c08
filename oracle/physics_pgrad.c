/* ORACLE -- TEST INFRASTRUCTURE ONLY (see dg_oracle.h).
 *
 * PressureGradientModel (src/Atmos/Model/ref_state.jl:196-233): state grad p (3), auxiliary
 * p (1), flux_first_order!: F.grad_p -= p I; everything else empty; boundary_state! nothing.
 * Used by grad reference_pressure (:235-262) for the discrete hydrostatic balance of the
 * reference density (:150-175); pinned by test/Atmos/Model/discrete_hydrostatic_balance.jl.
 */
#include <stdlib.h>

#include "dg_oracle.h"

static void pg_flux1(const void *p, double *F, const double *Q, const double *aux, double t, int dir)
{
    (void)p; (void)Q; (void)t; (void)dir;
    for (int c = 0; c < 3; ++c)
        for (int d = 0; d < 3; ++d) F[d + 3 * c] -= aux[0] * (d == c ? 1.0 : 0.0);
}
static void pg_flux2(const void *p, double *F, const double *Q, const double *gf, const double *hyp,
                     const double *aux, double t)
{
    (void)p; (void)F; (void)Q; (void)gf; (void)hyp; (void)aux; (void)t;
}
static void pg_source(const void *p, double *S, const double *Q, const double *gf, const double *aux,
                      double t, int dir)
{
    (void)p; (void)S; (void)Q; (void)gf; (void)aux; (void)t; (void)dir;
}
static void pg_garg(const void *p, double *G, const double *Q, const double *aux, double t)
{
    (void)p; (void)G; (void)Q; (void)aux; (void)t;
}
static void pg_gflux(const void *p, double *gf, const double *g, const double *Q, const double *aux,
                     double t)
{
    (void)p; (void)gf; (void)g; (void)Q; (void)aux; (void)t;
}
static void pg_ws(const void *p, double *ws, const double *n, const double *Q, const double *aux,
                  double t, int fd)
{
    (void)p; (void)n; (void)Q; (void)aux; (void)t; (void)fd;
    ws[0] = ws[1] = ws[2] = 0.0;
}
static void pg_bstate(const void *p, int kind, int bctag, double *QP, double *auxP, const double *n,
                      const double *QM, const double *auxM, double t, const double *Q1,
                      const double *aux1)
{
    (void)p; (void)kind; (void)bctag; (void)QP; (void)auxP; (void)n; (void)QM; (void)auxM; (void)t;
    (void)Q1; (void)aux1;
}
static void pg_bflux2(const void *p, int bctag, double *F, double *QP, double *gfP, double *hypP,
                      double *auxP, const double *n, const double *QM, const double *gfM,
                      const double *hypM, const double *auxM, double t, const double *Q1,
                      const double *gf1, const double *aux1)
{
    (void)p; (void)bctag; (void)F; (void)QP; (void)gfP; (void)hypP; (void)auxP; (void)n; (void)QM;
    (void)gfM; (void)hypM; (void)auxM; (void)t; (void)Q1; (void)gf1; (void)aux1;
}
static void pg_bdiv(const void *p, int b, double *gP, double *aP, const double *n, const double *gM,
                    const double *aM, double t)
{
    (void)p; (void)b; (void)gP; (void)aP; (void)n; (void)gM; (void)aM; (void)t;
}
static void pg_bhigher(const void *p, int b, double *QP, double *aP, double *lP, const double *n,
                       const double *QM, const double *aM, const double *lM, double t)
{
    (void)p; (void)b; (void)QP; (void)aP; (void)lP; (void)n; (void)QM; (void)aM; (void)lM; (void)t;
}

orc_physics *orc_pgrad_new(const int *ip, const double *dp, int nf_first)
{
    (void)ip; (void)dp;
    orc_physics *ph = (orc_physics *)calloc(1, sizeof(orc_physics));
    ph->ns = 3;
    ph->naux = 1;
    ph->nf_first = nf_first;
    ph->p = calloc(1, 8);
    ph->flux_first_order = pg_flux1;
    ph->flux_second_order = pg_flux2;
    ph->source = pg_source;
    ph->gradient_argument = pg_garg;
    ph->gradient_flux = pg_gflux;
    ph->post_gradient_laplacian = pg_gflux;
    ph->wavespeed = pg_ws;
    ph->boundary_state = pg_bstate;
    ph->boundary_flux_second_order = pg_bflux2;
    ph->boundary_state_divergence = pg_bdiv;
    ph->boundary_state_higher_order = pg_bhigher;
    return ph;
}
