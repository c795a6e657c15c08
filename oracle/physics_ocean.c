/* ORACLE -- TEST INFRASTRUCTURE ONLY (see dg_oracle.h).
 *
 * Pointwise physics of the hydrostatic Boussinesq ocean model (uncoupled), restated from
 *   src/Ocean/HydrostaticBoussinesq/hydrostatic_boussinesq_model.jl:107-112 (state u[2], eta,
 *       theta), :135-144 (aux y, w, pkin, wz0, u_d[2], dG_u[2]), :175-290 (gradient argument /
 *       flux, viscosity and diffusivity tensors), :419-520 (first-order fluxes), :539-552
 *       (second-order flux), :571-606 (source), :613 (wavespeed), :621-635 (update_penalty!)
 *   src/Ocean/HydrostaticBoussinesq/bc_velocity.jl, bc_temperature.jl (OceanBC)
 *   src/Ocean/OceanProblems/simple_box_problem.jl:56-127 (Coriolis parameter of SimpleBox)
 * Pinned by test/Ocean/HydrostaticBoussinesq/test_3D_spindown.jl with
 * test/Ocean/refvals/3D_hydrostatic_spindown_refvals.jl (StateCheck, 12 digits).
 *
 * Parameter block:
 *   iparam[3]=coupling (0 Uncoupled, 1 Coupled: src/Ocean/SplitExplicit/HydrostaticBoussinesqCoupling.jl)
 *   iparam[0]=momentum advection (NonLinearAdvectionTerm) [1]=tracer advection
 *   [2]=Coriolis: 0 SimpleBox{Fixed} (f = -0), 1 SimpleBox{Rotating} (f = f_o), 2 beta plane
 *   [6]=nbc [7..13]=bc of tag 1..7: velocity kind + 8 * temperature kind, velocity kinds
 *   1 Impenetrable(NoSlip) 2 Impenetrable(FreeSlip) 3 Penetrable(FreeSlip)
 *   4 Impenetrable(KinematicStress) 5 Penetrable(KinematicStress); temperature 0 Insulating,
 *   1 TemperatureFlux (stress and flux of the OceanGyre problem, ocean_gyre.jl:84-115)
 *   dparam[0..10] = grav c_h c_z alpha_T nu_h nu_z kappa_h kappa_z kappa_c f_o beta
 *   dparam[11..15] = tau_o rho_o L_y lambda_r theta_E
 */
#include <math.h>
#include <stdlib.h>

#include "dg_oracle.h"

typedef struct {
    int madv, tadv, cor, coupled, nbc, bc[8];
    double grav, ch, cz, aT, nuh, nuz, kh, kz, kc, fo, beta, tau_o, rho_o, Ly, lam_r, thE;
} ocean_t;
enum { U = 0, V = 1, ETA = 2, TH = 3 };                 /* prognostic */
enum { AY = 0, AW = 1, APKIN = 2, AWZ0 = 3, AUD = 4 };  /* auxiliary (u_d at 4, 5) */
enum { GDIVH = 0, GNU = 1, GKAPPA = 7 };                /* gradient flux */
enum { BV_NOSLIP = 1, BV_FREESLIP = 2, BV_PENETRABLE = 3, BV_STRESS = 4, BV_PEN_STRESS = 5 };

static void oc_flux1(const void *p_, double *F, const double *Q, const double *aux, double t, int dir)
{
    const ocean_t *m = (const ocean_t *)p_;
    (void)t; (void)dir;
    static const double Ih[3][2] = {{1, -0.0}, {-0.0, 1}, {-0.0, -0.0}};
    const double ge = m->grav * Q[ETA], gp = m->grav * aux[APKIN];
    for (int c = 0; c < 2; ++c)
        for (int d = 0; d < 3; ++d) {
            if (!m->coupled) F[d + 3 * c] += ge * Ih[d][c]; /* hydrostatic_pressure! (Uncoupled) */
            F[d + 3 * c] += gp * Ih[d][c];                  /* kinematic_pressure! */
        }
    const double v[3] = {Q[U], Q[V], aux[AW]};
    if (m->madv)
        for (int c = 0; c < 2; ++c)
            for (int d = 0; d < 3; ++d) F[d + 3 * c] += v[d] * Q[c];
    if (m->tadv)
        for (int d = 0; d < 3; ++d) F[d + 3 * TH] += v[d] * Q[TH];
}

static void oc_flux2(const void *p_, double *F, const double *Q, const double *gf, const double *hyp,
                     const double *aux, double t)
{
    (void)p_; (void)Q; (void)hyp; (void)aux; (void)t;
    for (int c = 0; c < 2; ++c)
        for (int d = 0; d < 3; ++d) F[d + 3 * c] += gf[GNU + d + 3 * c];
    for (int d = 0; d < 3; ++d) F[d + 3 * TH] += gf[GKAPPA + d];
}

static double coriolis(const ocean_t *m, double y)
{
    return m->cor == 0 ? -0.0 : (m->cor == 1 ? m->fo : m->fo + m->beta * y);
}

static void oc_source(const void *p_, double *S, const double *Q, const double *gf, const double *aux,
                      double t, int dir)
{
    const ocean_t *m = (const ocean_t *)p_;
    (void)gf; (void)t; (void)dir;
    S[ETA] += aux[AWZ0];
    const double f = coriolis(m, aux[AY]);
    if (m->coupled) { /* coriolis_force!(::Coupled): the velocity deviation from the vertical mean */
        S[U] -= -f * aux[AUD + 1];
        S[V] -= f * aux[AUD];
    } else {
        S[U] -= -f * Q[V];
        S[V] -= f * Q[U];
    }
    /* forcing: noforcing(args...) = 0 for every variable */
    S[U] += 0;
    S[V] += 0;
    S[ETA] += 0;
    S[TH] += 0;
}

static void oc_gradarg(const void *p_, double *G, const double *Q, const double *aux, double t)
{
    const ocean_t *m = (const ocean_t *)p_;
    (void)t;
    G[4] = Q[TH]; /* Gradient vars: grad u[2], grad u_d[2] (untouched when uncoupled), grad theta */
    G[0] = Q[U];
    G[1] = Q[V];
    if (m->coupled) { /* velocity_gradient_argument!(::Coupled) */
        G[2] = aux[AUD];
        G[3] = aux[AUD + 1];
    }
}

static void oc_gradflux(const void *p_, double *D, const double *g, const double *Q, const double *aux,
                        double t)
{
    const ocean_t *m = (const ocean_t *)p_;
    (void)Q; (void)aux; (void)t;
    D[GDIVH] = g[0 + 3 * 0] + g[1 + 3 * 1];
    const double nu[3] = {m->nuh, m->nuh, m->nuz};
    for (int c = 0; c < 2; ++c)
        for (int d = 0; d < 3; ++d) {
            /* Coupled: horizontal derivatives of u_d, vertical derivative of u */
            const double gu = (m->coupled && d < 2) ? g[d + 3 * (2 + c)] : g[d + 3 * c];
            D[GNU + d + 3 * c] = -nu[d] * gu;
        }
    const double dthdz = g[2 + 3 * 4];
    const double kap[3] = {m->kh, m->kh, dthdz < 0 ? m->kc : m->kz};
    for (int d = 0; d < 3; ++d) D[GKAPPA + d] = -kap[d] * g[d + 3 * 4];
}

static void oc_postlap(const void *p, double *h, const double *gl, const double *Q, const double *aux,
                       double t)
{
    (void)p; (void)h; (void)gl; (void)Q; (void)aux; (void)t;
}

static void oc_wavespeed(const void *p_, double *ws, const double *n, const double *Q, const double *aux,
                         double t, int facedir)
{
    const ocean_t *m = (const ocean_t *)p_;
    (void)Q; (void)aux; (void)t; (void)facedir;
    const double w = fabs(m->ch * n[0] + m->ch * n[1] + m->cz * n[2]);
    for (int s = 0; s < 4; ++s) ws[s] = w;
}

static void oc_penalty(const void *p_, double *pen, const double *n, const double *QM, const double *QP)
{
    (void)p_; (void)n; (void)QM; (void)QP;
    pen[ETA] = -0.0;
}

static void oc_bstate(const void *p_, int kind, int bctag, double *QP, double *auxP, const double *n,
                      const double *QM, const double *auxM, double t, const double *Q1,
                      const double *aux1)
{
    const ocean_t *m = (const ocean_t *)p_;
    (void)t; (void)Q1; (void)aux1;
    const int bv = m->bc[bctag - 1] & 7;
    if (bv == BV_NOSLIP) {
        if (kind == ORC_BS_FIRST) {
            QP[U] = -QM[U];
            QP[V] = -QM[V];
            auxP[AW] = -auxM[AW];
        } else {
            QP[U] = -0.0;
            QP[V] = -0.0;
            auxP[AW] = -0.0;
        }
    } else if (bv == BV_FREESLIP || bv == BV_STRESS) { /* KinematicStress -> FreeSlip here */
        const double v[3] = {QM[U], QM[V], auxM[AW]};
        double vp[3];
        if (kind == ORC_BS_FIRST) { /* v - ((2 n) . v) n */
            const double dn = (2 * n[0]) * v[0] + (2 * n[1]) * v[1] + (2 * n[2]) * v[2];
            for (int d = 0; d < 3; ++d) vp[d] = v[d] - dn * n[d];
        } else { /* v - (n . v) n */
            const double dn = n[0] * v[0] + n[1] * v[1] + n[2] * v[2];
            for (int d = 0; d < 3; ++d) vp[d] = v[d] - dn * n[d];
        }
        QP[U] = vp[0];
        QP[V] = vp[1];
        auxP[AW] = vp[2];
    }
    QP[TH] = QM[TH]; /* Insulating */
}

/* boundary_state!(::NumericalFluxSecondOrder, ...) then flux_second_order! on the plus side */
static void oc_bflux2(const void *p_, int bctag, double *F, double *QP, double *gfP, double *hypP,
                      double *auxP, const double *n, const double *QM, const double *gfM,
                      const double *hypM, const double *auxM, double t, const double *Q1,
                      const double *gf1, const double *aux1)
{
    const ocean_t *m = (const ocean_t *)p_;
    (void)hypM; (void)Q1; (void)gf1; (void)aux1;
    const int bv = m->bc[bctag - 1] & 7;
    if (bv == BV_NOSLIP) {
        QP[U] = -QM[U];
        QP[V] = -QM[V];
        auxP[AW] = -auxM[AW];
        for (int q = 0; q < 6; ++q) gfP[GNU + q] = gfM[GNU + q];
    } else if (bv == BV_STRESS || bv == BV_PEN_STRESS) {
        /* kinematic_stress(p::OceanGyre, y, rho) = [(tau_o / rho) cos(y pi / L_y), -0] */
        const double st[2] = {(m->tau_o / m->rho_o) * cos(auxM[AY] * M_PI / m->Ly), -0.0};
        QP[U] = QM[U];
        QP[V] = QM[V];
        for (int c = 0; c < 2; ++c)
            for (int d = 0; d < 3; ++d) gfP[GNU + d + 3 * c] = n[d] * st[c];
    } else {
        QP[U] = QM[U];
        QP[V] = QM[V];
        auxP[AW] = auxM[AW];
        for (int c = 0; c < 2; ++c)
            for (int d = 0; d < 3; ++d) gfP[GNU + d + 3 * c] = n[d] * -0.0;
    }
    QP[TH] = QM[TH];
    if ((m->bc[bctag - 1] >> 3) == 1) { /* TemperatureFlux: surface_flux(p, y, theta) */
        const double thr = m->thE * (1 - auxM[AY] / m->Ly);
        const double fl = m->lam_r * (QM[TH] - thr);
        for (int d = 0; d < 3; ++d) gfP[GKAPPA + d] = n[d] * fl;
    } else {
        for (int d = 0; d < 3; ++d) gfP[GKAPPA + d] = n[d] * -0.0;
    }
    oc_flux2(p_, F, QP, gfP, hypP, auxP, t);
}
static void oc_bdiv(const void *p, int b, double *gP, double *aP, const double *n, const double *gM,
                    const double *aM, double t)
{
    (void)p; (void)b; (void)gP; (void)aP; (void)n; (void)gM; (void)aM; (void)t;
}
static void oc_bhigher(const void *p, int b, double *QP, double *aP, double *lP, const double *n,
                       const double *QM, const double *aM, const double *lM, double t)
{
    (void)p; (void)b; (void)QP; (void)aP; (void)lP; (void)n; (void)QM; (void)aM; (void)lM; (void)t;
}

/* src/Ocean/HydrostaticBoussinesq/Courant.jl:13-111: kind 0 advective, 1 nondiffusive (gravity
   waves), 2 diffusive (1000 kappa_z for convective adjustment), 3 viscous */
static double oc_courant(const void *p_, int kind, const double *Q, const double *aux,
                         const double *gf, double dx, double dt, double t, int direction)
{
    const ocean_t *m = (const ocean_t *)p_;
    (void)gf; (void)t;
    if (kind == 0) {
        double ub;
        if (direction == ORC_VERTICAL) ub = fabs(aux[AW]);
        else if (direction == ORC_HORIZONTAL) ub = sqrt(Q[U] * Q[U] + Q[V] * Q[V]);
        else ub = sqrt(Q[U] * Q[U] + Q[V] * Q[V] + aux[AW] * aux[AW]);
        return dt * ub / dx;
    }
    if (kind == 1) return dt * m->ch / dx;
    const double h = kind == 3 ? m->nuh : m->kh;
    const double z = kind == 3 ? m->nuz : 1000 * m->kz;
    const double nb = direction == ORC_VERTICAL ? z
                    : direction == ORC_HORIZONTAL ? sqrt(2.0) * h
                    : sqrt(2 * (h * h) + z * z);
    return dt * nb / (dx * dx);
}

orc_physics *orc_ocean_new(const int *ip, const double *dp, int nf_first)
{
    orc_physics *ph = (orc_physics *)calloc(1, sizeof(orc_physics));
    ocean_t *m = (ocean_t *)calloc(1, sizeof(ocean_t));
    m->madv = ip[0]; m->tadv = ip[1]; m->cor = ip[2]; m->coupled = ip[3]; m->nbc = ip[6];
    for (int i = 0; i < 7; ++i) m->bc[i] = ip[7 + i];
    m->grav = dp[0]; m->ch = dp[1]; m->cz = dp[2]; m->aT = dp[3]; m->nuh = dp[4]; m->nuz = dp[5];
    m->kh = dp[6]; m->kz = dp[7]; m->kc = dp[8]; m->fo = dp[9]; m->beta = dp[10];
    m->tau_o = dp[11]; m->rho_o = dp[12]; m->Ly = dp[13]; m->lam_r = dp[14]; m->thE = dp[15];
    ph->ns = 4;
    ph->naux = 8;
    ph->ngrad = 5;
    ph->ngf = 10;
    ph->nf_first = nf_first;
    ph->p = m;
    ph->flux_first_order = oc_flux1;
    ph->flux_second_order = oc_flux2;
    ph->source = oc_source;
    ph->gradient_argument = oc_gradarg;
    ph->gradient_flux = oc_gradflux;
    ph->post_gradient_laplacian = oc_postlap;
    ph->wavespeed = oc_wavespeed;
    ph->boundary_state = oc_bstate;
    ph->boundary_flux_second_order = oc_bflux2;
    ph->boundary_state_divergence = oc_bdiv;
    ph->boundary_state_higher_order = oc_bhigher;
    ph->update_penalty = oc_penalty;
    ph->courant = oc_courant;
    return ph;
}
