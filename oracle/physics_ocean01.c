/* ORACLE -- TEST INFRASTRUCTURE ONLY (see dg_oracle.h).
 *
 * Pointwise physics of the older split-explicit ocean of the reference,
 * src/Ocean/SplitExplicit01, restated from
 *   OceanModel.jl:176-195 (state u[2], eta, theta; aux w, pkin, wz0, u_d[2], dG_u[2], y),
 *       :212-291 (gradient argument / flux, viscosity and diffusivity), :368-449 (fluxes and
 *       source), :543-561 (wavespeed, update_penalty!)
 *   Continuity3dModel.jl:1-76 (the operator whose theta tendency is -grad_h . u)
 *   BarotropicModel.jl:12-196 (state U[2], eta; aux G_U, U_c, eta_c, U_s, eta_s, Delta_u,
 *       eta_diag, Delta_eta, y)
 *   OceanBoundaryConditions.jl:1-642
 * Pinned by test/Ocean/SplitExplicit/simple_box_2dt.jl with
 * test/Ocean/refvals/simple_box_2dt_refvals.jl (StateCheck of 28 fields after five days).
 *
 * Parameter block: iparam[0] numImplSteps > 0, [6] nbc, [7..13] boundary condition of tag
 * 1..7 (1 CoastlineFreeSlip 2 CoastlineNoSlip 3 OceanFloorFreeSlip 4 OceanFloorNoSlip
 * 6..9 OceanSurface {NoStressNoForcing, StressNoForcing, NoStressForcing, StressForcing});
 * dparam[0..10] = grav c_h c_z alpha_T nu_h nu_z kappa_h kappa_z kappa_c f_o beta,
 * [11..16] = tau_o rho_o L_y lambda_r theta_E H.
 */
#include <math.h>
#include <stdlib.h>

#include "dg_oracle.h"

typedef struct {
    int impl, nbc, bc[8];
    double grav, ch, cz, aT, nuh, nuz, kh, kz, kc, fo, beta, tau_o, rho_o, Ly, lam_r, thE, H;
} se01_t;
enum { CF = 1, CN = 2, FF = 3, FN = 4, SNN = 6, SSN = 7, SNF = 8, SSF = 9 };
enum { U = 0, V = 1, ETA = 2, TH = 3 };
enum { AW = 0, APKIN = 1, AWZ0 = 2, AUD = 3, ADGU = 5, AY = 7 };
enum { GNU = 0, GKAPPA = 6 };
enum { BU1 = 0, BU2 = 1, BETA_ = 2, BAGU = 0, BAY = 12 };
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static se01_t *se01_params(const int *ip, const double *dp)
{
    se01_t *m = (se01_t *)calloc(1, sizeof(se01_t));
    m->impl = ip[0];
    m->nbc = ip[6];
    for (int i = 0; i < 7; ++i) m->bc[i] = ip[7 + i];
    m->grav = dp[0]; m->ch = dp[1]; m->cz = dp[2]; m->aT = dp[3]; m->nuh = dp[4]; m->nuz = dp[5];
    m->kh = dp[6]; m->kz = dp[7]; m->kc = dp[8]; m->fo = dp[9]; m->beta = dp[10];
    m->tau_o = dp[11]; m->rho_o = dp[12]; m->Ly = dp[13]; m->lam_r = dp[14]; m->thE = dp[15];
    m->H = dp[16];
    return m;
}
static double gw_speed(const se01_t *m, const double *n)
{
    return fabs(m->ch * n[0] + m->ch * n[1] + m->cz * n[2]);
}
static void none_postlap(const void *p, double *h, const double *gl, const double *Q, const double *aux, double t)
{ (void)p; (void)h; (void)gl; (void)Q; (void)aux; (void)t; }
static void none_bdiv(const void *p, int b, double *gP, double *aP, const double *n, const double *gM,
                      const double *aM, double t)
{ (void)p; (void)b; (void)gP; (void)aP; (void)n; (void)gM; (void)aM; (void)t; }
static void none_bhigher(const void *p, int b, double *QP, double *aP, double *lP, const double *n,
                         const double *QM, const double *aM, const double *lM, double t)
{ (void)p; (void)b; (void)QP; (void)aP; (void)lP; (void)n; (void)QM; (void)aM; (void)lM; (void)t; }

/* ---- OceanModel -------------------------------------------------------------------- */
static void oc_flux1(const void *p_, double *F, const double *Q, const double *aux, double t, int dir)
{
    (void)p_; (void)t; (void)dir;
    static const double Ih[3][2] = {{1, -0.0}, {-0.0, 1}, {-0.0, -0.0}};
    const double v[3] = {Q[U], Q[V], aux[AW]};
    for (int d = 0; d < 3; ++d) F[d + 3 * TH] += v[d] * Q[TH];      /* div (u theta) */
    for (int c = 0; c < 2; ++c)
        for (int d = 0; d < 3; ++d) F[d + 3 * c] += aux[APKIN] * Ih[d][c]; /* grad_h pkin */
}
static void oc_flux2(const void *p_, double *F, const double *Q, const double *gf, const double *hyp,
                     const double *aux, double t)
{
    (void)p_; (void)Q; (void)hyp; (void)aux; (void)t;
    for (int q = 0; q < 6; ++q) F[q] += gf[GNU + q];
    for (int d = 0; d < 3; ++d) F[d + 3 * TH] += gf[GKAPPA + d];
}
static void oc_source(const void *p_, double *S, const double *Q, const double *gf, const double *aux,
                      double t, int dir)
{
    const se01_t *m = (const se01_t *)p_;
    (void)Q; (void)gf; (void)t; (void)dir;
    const double f = m->fo + m->beta * aux[AY];
    S[U] -= -f * aux[AUD + 1];
    S[V] -= f * aux[AUD];
    S[U] += aux[ADGU];
    S[V] += aux[ADGU + 1];
    S[ETA] += aux[AWZ0];
}
static void oc_gradarg(const void *p_, double *G, const double *Q, const double *aux, double t)
{
    (void)p_; (void)t;
    G[0] = Q[U]; G[1] = Q[V]; G[2] = aux[AUD]; G[3] = aux[AUD + 1]; G[4] = Q[TH];
}
static void oc_gradflux(const void *p_, double *D, const double *g, const double *Q, const double *aux, double t)
{
    const se01_t *m = (const se01_t *)p_;
    (void)Q; (void)aux; (void)t;
    for (int c = 0; c < 2; ++c) {
        D[GNU + 0 + 3 * c] = -(m->nuh * g[0 + 3 * (2 + c)]);
        D[GNU + 1 + 3 * c] = -(m->nuh * g[1 + 3 * (2 + c)]);
        D[GNU + 2 + 3 * c] = -(m->nuz * g[2 + 3 * c]);
    }
    const double dthdz = g[2 + 3 * 4];
    const double kv = m->impl ? m->kz * 0.5 : (dthdz < 0 ? m->kc : m->kz);
    const double kap[3] = {m->kh, m->kh, kv};
    for (int d = 0; d < 3; ++d) D[GKAPPA + d] = -kap[d] * g[d + 3 * 4];
}
static void oc_wavespeed(const void *p_, double *ws, const double *n, const double *Q, const double *aux,
                         double t, int facedir)
{
    (void)Q; (void)aux; (void)t; (void)facedir;
    const double w = gw_speed((const se01_t *)p_, n);
    for (int s = 0; s < 4; ++s) ws[s] = w;
}
static void oc_penalty(const void *p, double *pen, const double *n, const double *QM, const double *QP)
{ (void)p; (void)n; (void)QM; (void)QP; pen[ETA] = -0.0; }
static void oc_bstate(const void *p_, int kind, int bctag, double *QP, double *auxP, const double *n,
                      const double *QM, const double *auxM, double t, const double *Q1, const double *aux1)
{
    const se01_t *m = (const se01_t *)p_;
    (void)t; (void)Q1; (void)aux1;
    const int bc = m->bc[bctag - 1];
    if (bc == CN || bc == FN) {
        if (kind == ORC_BS_FIRST) { QP[U] = -QM[U]; QP[V] = -QM[V]; }
        else {
            QP[U] = -0.0; QP[V] = -0.0;
            if (bc == CN) auxP[AUD] = auxP[AUD + 1] = -0.0;
        }
    } else if (bc == CF) {
        const double f = kind == ORC_BS_FIRST ? 2.0 : 1.0;
        const double dn = n[0] * QM[U] + n[1] * QM[V];
        QP[U] = QM[U] - f * dn * n[0];
        QP[V] = QM[V] - f * dn * n[1];
        if (kind != ORC_BS_FIRST) {
            const double dd = n[0] * auxM[AUD] + n[1] * auxM[AUD + 1];
            auxP[AUD] = auxM[AUD] - dd * n[0];
            auxP[AUD + 1] = auxM[AUD + 1] - dd * n[1];
        }
    }
    if (bc == FN || bc == FF) auxP[AW] = kind == ORC_BS_FIRST ? -auxM[AW] : -0.0;
}
static void oc_bflux2(const void *p_, int bctag, double *F, double *QP, double *gfP, double *hypP,
                      double *auxP, const double *n, const double *QM, const double *gfM,
                      const double *hypM, const double *auxM, double t, const double *Q1,
                      const double *gf1, const double *aux1)
{
    const se01_t *m = (const se01_t *)p_;
    (void)gfM; (void)hypM; (void)Q1; (void)gf1; (void)aux1;
    const int bc = m->bc[bctag - 1];
    double st[2] = {-0.0, -0.0}, fl = -0.0;
    const int keep = bc == CN || bc == FN;      /* D+.nu grad u = D-.nu grad u */
    if (bc == SSN || bc == SSF) {
        const double tauz = -(m->tau_o / m->rho_o) * cos(auxM[AY] * M_PI / m->Ly);
        st[0] = -tauz;
    }
    if (bc == SNF || bc == SSF) {
        const double thr = m->thE * (1 - auxM[AY] / m->Ly);
        fl = -(m->lam_r * (thr - QM[TH]));
    }
    if (!keep)
        for (int c = 0; c < 2; ++c)
            for (int d = 0; d < 3; ++d) gfP[GNU + d + 3 * c] = n[d] * st[c];
    for (int d = 0; d < 3; ++d) gfP[GKAPPA + d] = n[d] * fl;
    oc_flux2(p_, F, QP, gfP, hypP, auxP, t);
}

orc_physics *orc_ocean_se01_new(const int *ip, const double *dp, int nf_first)
{
    orc_physics *ph = (orc_physics *)calloc(1, sizeof(orc_physics));
    ph->ns = 4; ph->naux = 8; ph->ngrad = 5; ph->ngf = 9;
    ph->nf_first = nf_first;
    ph->p = se01_params(ip, dp);
    ph->flux_first_order = oc_flux1;
    ph->flux_second_order = oc_flux2;
    ph->source = oc_source;
    ph->gradient_argument = oc_gradarg;
    ph->gradient_flux = oc_gradflux;
    ph->post_gradient_laplacian = none_postlap;
    ph->wavespeed = oc_wavespeed;
    ph->boundary_state = oc_bstate;
    ph->boundary_flux_second_order = oc_bflux2;
    ph->boundary_state_divergence = none_bdiv;
    ph->boundary_state_higher_order = none_bhigher;
    ph->update_penalty = oc_penalty;
    return ph;
}

/* ---- Continuity3dModel ---------------------------------------------------------------- */
static void ct_flux1(const void *p_, double *F, const double *Q, const double *aux, double t, int dir)
{
    (void)p_; (void)aux; (void)t; (void)dir;
    const double v[3] = {Q[U], Q[V], -0.0};
    for (int d = 0; d < 3; ++d) F[d + 3 * TH] += v[d];
}
static void ct_flux2(const void *p, double *F, const double *Q, const double *gf, const double *hyp,
                     const double *aux, double t)
{ (void)p; (void)F; (void)Q; (void)gf; (void)hyp; (void)aux; (void)t; }
static void ct_source(const void *p, double *S, const double *Q, const double *gf, const double *aux, double t, int dir)
{ (void)p; (void)S; (void)Q; (void)gf; (void)aux; (void)t; (void)dir; }
static void ct_none5(const void *p, double *G, const double *Q, const double *aux, double t)
{ (void)p; (void)G; (void)Q; (void)aux; (void)t; }
static void ct_none6(const void *p, double *D, const double *g, const double *Q, const double *aux, double t)
{ (void)p; (void)D; (void)g; (void)Q; (void)aux; (void)t; }
static void ct_wavespeed(const void *p, double *ws, const double *n, const double *Q, const double *aux,
                         double t, int facedir)
{
    (void)p; (void)n; (void)Q; (void)aux; (void)t; (void)facedir;
    for (int s = 0; s < 4; ++s) ws[s] = -0.0;
}
static void ct_bstate(const void *p_, int kind, int bctag, double *QP, double *auxP, const double *n,
                      const double *QM, const double *auxM, double t, const double *Q1, const double *aux1)
{
    const se01_t *m = (const se01_t *)p_;
    (void)bctag; (void)auxP; (void)auxM; (void)t; (void)Q1; (void)aux1;
    if (kind != ORC_BS_FIRST) return;
    if (m->bc[0] == CN) { QP[U] = -QM[U]; QP[V] = -QM[V]; }
    else if (m->bc[0] == CF) {
        const double dn = n[0] * QM[U] + n[1] * QM[V];
        QP[U] = QM[U] - 2 * dn * n[0];
        QP[V] = QM[V] - 2 * dn * n[1];
    }
}
static void ct_bflux2(const void *p, int bctag, double *F, double *QP, double *gfP, double *hypP,
                      double *auxP, const double *n, const double *QM, const double *gfM,
                      const double *hypM, const double *auxM, double t, const double *Q1,
                      const double *gf1, const double *aux1)
{
    (void)p; (void)bctag; (void)F; (void)QP; (void)gfP; (void)hypP; (void)auxP; (void)n; (void)QM;
    (void)gfM; (void)hypM; (void)auxM; (void)t; (void)Q1; (void)gf1; (void)aux1;
}

orc_physics *orc_conti3d_se01_new(const int *ip, const double *dp, int nf_first)
{
    orc_physics *ph = (orc_physics *)calloc(1, sizeof(orc_physics));
    ph->ns = 4; ph->naux = 0; ph->ngrad = 0; ph->ngf = 0;
    ph->nf_first = nf_first;
    ph->p = se01_params(ip, dp);
    ph->flux_first_order = ct_flux1;
    ph->flux_second_order = ct_flux2;
    ph->source = ct_source;
    ph->gradient_argument = ct_none5;
    ph->gradient_flux = ct_none6;
    ph->post_gradient_laplacian = none_postlap;
    ph->wavespeed = ct_wavespeed;
    ph->boundary_state = ct_bstate;
    ph->boundary_flux_second_order = ct_bflux2;
    ph->boundary_state_divergence = none_bdiv;
    ph->boundary_state_higher_order = none_bhigher;
    return ph;
}

/* ---- BarotropicModel ------------------------------------------------------------------ */
static void bt_flux1(const void *p_, double *F, const double *Q, const double *aux, double t, int dir)
{
    const se01_t *m = (const se01_t *)p_;
    (void)aux; (void)t; (void)dir;
    static const double Ih[3][2] = {{1, 0}, {0, 1}, {0, 0}};
    const double Uv[3] = {Q[BU1], Q[BU2], 0.0};
    for (int d = 0; d < 3; ++d) F[d + 3 * BETA_] += Uv[d];
    const double ghe = m->grav * m->H * Q[BETA_];
    for (int c = 0; c < 2; ++c)
        for (int d = 0; d < 3; ++d) F[d + 3 * c] += ghe * Ih[d][c];
}
static void bt_flux2(const void *p_, double *F, const double *Q, const double *gf, const double *hyp,
                     const double *aux, double t)
{
    (void)p_; (void)Q; (void)hyp; (void)aux; (void)t;
    for (int q = 0; q < 6; ++q) F[q] += gf[q];
}
static void bt_source(const void *p_, double *S, const double *Q, const double *gf, const double *aux,
                      double t, int dir)
{
    const se01_t *m = (const se01_t *)p_;
    (void)gf; (void)t; (void)dir;
    const double f = m->fo + m->beta * aux[BAY];
    S[BU1] -= -f * Q[BU2];
    S[BU2] -= f * Q[BU1];
    S[BU1] += aux[BAGU];
    S[BU2] += aux[BAGU + 1];
}
static void bt_gradarg(const void *p_, double *G, const double *Q, const double *aux, double t)
{ (void)p_; (void)aux; (void)t; G[0] = Q[BU1]; G[1] = Q[BU2]; }
static void bt_gradflux(const void *p_, double *D, const double *g, const double *Q, const double *aux, double t)
{
    const se01_t *m = (const se01_t *)p_;
    (void)Q; (void)aux; (void)t;
    const double nu[3] = {m->nuh, m->nuh, 0.0};
    for (int c = 0; c < 2; ++c)
        for (int d = 0; d < 3; ++d) D[d + 3 * c] = -nu[d] * g[d + 3 * c];
}
static void bt_wavespeed(const void *p_, double *ws, const double *n, const double *Q, const double *aux,
                         double t, int facedir)
{
    (void)Q; (void)aux; (void)t; (void)facedir;
    ws[0] = ws[1] = ws[2] = gw_speed((const se01_t *)p_, n);
}
static void bt_penalty(const void *p, double *pen, const double *n, const double *QM, const double *QP)
{ (void)p; (void)n; (void)QM; (void)QP; pen[BETA_] = -0.0; }
static void bt_bstate(const void *p_, int kind, int bctag, double *QP, double *auxP, const double *n,
                      const double *QM, const double *auxM, double t, const double *Q1, const double *aux1)
{
    const se01_t *m = (const se01_t *)p_;
    (void)bctag; (void)auxP; (void)auxM; (void)t; (void)Q1; (void)aux1;
    if (m->bc[0] == CN) {
        QP[BU1] = kind == ORC_BS_FIRST ? -QM[BU1] : -0.0;
        QP[BU2] = kind == ORC_BS_FIRST ? -QM[BU2] : -0.0;
    } else if (m->bc[0] == CF) {
        const double f = kind == ORC_BS_FIRST ? 2.0 : 1.0;
        const double dn = n[0] * QM[BU1] + n[1] * QM[BU2];
        QP[BU1] = QM[BU1] - f * dn * n[0];
        QP[BU2] = QM[BU2] - f * dn * n[1];
    }
}
static void bt_bflux2(const void *p_, int bctag, double *F, double *QP, double *gfP, double *hypP,
                      double *auxP, const double *n, const double *QM, const double *gfM,
                      const double *hypM, const double *auxM, double t, const double *Q1,
                      const double *gf1, const double *aux1)
{
    const se01_t *m = (const se01_t *)p_;
    (void)bctag; (void)QM; (void)gfM; (void)hypM; (void)auxM; (void)Q1; (void)gf1; (void)aux1;
    if (m->bc[0] == CF)
        for (int c = 0; c < 2; ++c)
            for (int d = 0; d < 3; ++d) gfP[d + 3 * c] = n[d] * -0.0;
    bt_flux2(p_, F, QP, gfP, hypP, auxP, t);
}

orc_physics *orc_baro_se01_new(const int *ip, const double *dp, int nf_first)
{
    orc_physics *ph = (orc_physics *)calloc(1, sizeof(orc_physics));
    ph->ns = 3; ph->naux = 13; ph->ngrad = 2; ph->ngf = 6;
    ph->nf_first = nf_first;
    ph->p = se01_params(ip, dp);
    ph->flux_first_order = bt_flux1;
    ph->flux_second_order = bt_flux2;
    ph->source = bt_source;
    ph->gradient_argument = bt_gradarg;
    ph->gradient_flux = bt_gradflux;
    ph->post_gradient_laplacian = none_postlap;
    ph->wavespeed = bt_wavespeed;
    ph->boundary_state = bt_bstate;
    ph->boundary_flux_second_order = bt_bflux2;
    ph->boundary_state_divergence = none_bdiv;
    ph->boundary_state_higher_order = none_bhigher;
    ph->update_penalty = bt_penalty;
    return ph;
}
