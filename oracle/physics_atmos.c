/* ORACLE -- TEST INFRASTRUCTURE ONLY (see dg_oracle.h).
 *
 * Pointwise physics of the dry AtmosModel configurations in scope, restated from
 *   src/Atmos/Model/AtmosModel.jl:625-690 (gradient argument), :808-828 (wavespeed)
 *   src/Atmos/Model/tendencies_mass.jl:5-7, tendencies_momentum.jl:13-29,52-55,62-84,
 *       tendencies_energy.jl:7-21,37-56      (fluxes and sources, summed in the order of
 *       atmos_tendencies.jl / BalanceLaws/sum_tendencies.jl)
 *   src/Atmos/Model/energy.jl:17-28,48-57, moisture.jl:47-62 (DryModel aux update)
 *   src/Atmos/Model/bc_momentum.jl:25-52, bc_energy.jl:10-20, boundaryconditions.jl:60-100
 *   src/Common/TurbulenceClosures/TurbulenceClosures.jl:354-420 (constant viscosity),
 *       :411-497 (SmagorinskyLilly), :877-912 (DryBiharmonic)
 *   experiments/AtmosGCM/heldsuarez.jl:106-172 (Held-Suarez forcing)
 * Thermodynamics.jl 0.3.2 / CLIMAParameters.jl 0.1.11 (not vendored in the reference):
 * dry closed forms, pinned by test/Numerics/DGMethods/Euler/isentropicvortex.jl:105.
 *
 * Parameter block (shared data contract with the product's descriptor):
 *   iparam[0]=orientation (0 none, 1 flat, 2 spherical) [1]=hydrostatic ref state
 *   [2]=subtract_off [3]=viscosity kind (0 dynamic rho*nu, 1 kinematic nu)
 *   [4]=DryBiharmonic [5]=source bits (1 gravity, 2 coriolis, 4 Held-Suarez)
 *   [6]=nbc [7..13]=bc kind of tag 1..7 (1 = AtmosBC default: Impenetrable(FreeSlip), Insulating)
 *   dparam[0]=viscosity [1]=tau_hyper [2..12]=R_d cp_d cv_d T_0 grav Omega MSLP day
 *                                          planet_radius inv_Pr_turb kappa_d  [13]=C_smag
 *   iparam[14]=turbulence closure (0 constant viscosity, 1 SmagorinskyLilly)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "dg_oracle.h"

typedef struct {
    int orient, ref, subtract, kinematic, hyper, src, nbc, bc[8], smag, withdiv, zero_h;
    double visc, tau, R_d, cp_d, cv_d, T_0, grav, Omega, MSLP, day, a, invPr, kappa, C_smag;
    int oPhi, oRef, oTurb, oDelta, oMoist;
} atmos_t;

/* ---- dry thermodynamics ------------------------------------------------------------ */
static inline double e_pot_of(const atmos_t *m, const double *aux) { return m->orient ? aux[m->oPhi] : -0.0; }
static inline double internal_energy(const atmos_t *m, const double *Q, const double *aux)
{
    const double rho = Q[0];
    const double rhoinv = 1 / rho;
    const double rhoe_kin = rhoinv * (Q[1] * Q[1] + Q[2] * Q[2] + Q[3] * Q[3]) / 2;
    const double rhoe_pot = rho * e_pot_of(m, aux);
    const double rhoe_int = Q[4] - rhoe_kin - rhoe_pot;
    return rhoinv * rhoe_int;
}
static inline double air_T(const atmos_t *m, double e_int) { return m->T_0 + e_int / m->cv_d; }
static inline double air_p(const atmos_t *m, double T, double rho) { return m->R_d * rho * T; }
static inline double soundspeed(const atmos_t *m, double T)
{
    const double gamma = m->cp_d / m->cv_d;
    return sqrt(gamma * m->R_d * T);
}

/* ---- first-order fluxes: Mass Advect; Momentum Advect + PressureGradient; Energy Advect + Pressure */
static void at_flux1(const void *p_, double *F, const double *Q, const double *aux, double t, int dir)
{
    const atmos_t *m = (const atmos_t *)p_;
    (void)t; (void)dir;
    const double rho = Q[0];
    const double T = air_T(m, internal_energy(m, Q, aux));
    const double p = air_p(m, T, rho);
    double u[3];
    for (int d = 0; d < 3; ++d) u[d] = Q[1 + d] / rho;
    for (int d = 0; d < 3; ++d) F[d] = Q[1 + d];
    const double pp = (m->ref && m->subtract) ? p - aux[m->oRef + 1] : p;
    for (int c = 0; c < 3; ++c)
        for (int d = 0; d < 3; ++d)
            F[d + 3 * (1 + c)] = Q[1 + d] * u[c] + (0.0 + (d == c ? pp : 0.0));
    for (int d = 0; d < 3; ++d) F[d + 12] = u[d] * Q[4] + u[d] * p;
}

static inline double sym(const double *c, int i, int j)
{ /* SHermitianCompact{3}: (1,1),(2,1),(3,1),(2,2),(3,2),(3,3) */
    static const int idx[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    return c[idx[i][j]];
}

/* turbulence_tensors: nu (diagonal), tau = (-2 nu) S as a full 3x3 (row d scaled by nu_d),
 * tau[d + 3 c].  Constant viscosity: TurbulenceClosures.jl:372-408 (WithoutDivergence);
 * SmagorinskyLilly: :476-497 */
static void turbulence_tensors(const atmos_t *m, const double *Q, const double *gf,
                               const double *aux, double *nu, double *tau)
{
    const double *S = gf + 3;
    if (!m->smag) {
        const double v = m->kinematic ? m->visc : m->visc / Q[0];
        nu[0] = nu[1] = nu[2] = v;
    } else {
        /* strain_rate_magnitude(S) = sqrt(2 norm2(S)) (:268-285) */
        const double norm2 = S[0] * S[0] + 2 * (S[1] * S[1]) + 2 * (S[2] * S[2]) + S[3] * S[3] +
                             2 * (S[4] * S[4]) + S[5] * S[5];
        const double normS = sqrt(2 * norm2);
        double k[3];
        for (int d = 0; d < 3; ++d) k[d] = aux[m->oPhi + 1 + d] / m->grav;
        const double epsn = nextafter(fabs(normS), INFINITY) - fabs(normS); /* eps(normS) */
        const double Ri = gf[9] / (normS * normS + epsn);
        double c = 1.0 - Ri * m->invPr; /* clamp(.., 0, 1) */
        c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
        const double fb2 = sqrt(c);
        const double cd = m->C_smag * aux[m->oTurb];
        const double nu0 = normS * (cd * cd) + 1e-5;
        const double dk = nu0 * k[0] + nu0 * k[1] + nu0 * k[2]; /* dot(nu, k) */
        for (int d = 0; d < 3; ++d) {
            const double nv = k[d] * dk, nh = nu0 - nv;
            nu[d] = nh + nv * fb2;
        }
    }
    for (int d = 0; d < 3; ++d)
        for (int c = 0; c < 3; ++c) tau[d + 3 * c] = (-2 * nu[d]) * sym(S, d, c);
    if (!m->smag && m->withdiv) { /* (-2 nu) S + (2 nu / 3) tr(S) I */
        const double trS = S[0] + S[3] + S[5];
        for (int d = 0; d < 3; ++d) tau[d + 3 * d] += (2 * nu[d] / 3) * trS;
    }
}

static void at_flux2(const void *p_, double *F, const double *Q, const double *gf, const double *hyp,
                     const double *aux, double t)
{
    const atmos_t *m = (const atmos_t *)p_;
    (void)t;
    double nu[3], tau[9];
    turbulence_tensors(m, Q, gf, aux, nu, tau);
    const double rho = Q[0];
    /* Mass: no second-order tendencies for DryModel -> SVector(0,0,0) */
    for (int d = 0; d < 3; ++d) F[d] = 0.0;
    /* Momentum: ViscousStress (pad + tau*rho) [+ HyperdiffViscousFlux rho * nu grad^3 u_h] */
    for (int c = 0; c < 3; ++c)
        for (int d = 0; d < 3; ++d) {
            double v = 0.0 + tau[d + 3 * c] * rho;
            if (m->hyper) v = v + rho * hyp[d + 3 * c];
            F[d + 3 * (1 + c)] = v;
        }
    /* Energy: ViscousFlux tau*rho u, DiffEnthalpyFlux (-D_t .* grad h_tot) rho
       [+ HyperdiffEnthalpyFlux nu grad^3 h_tot * rho + HyperdiffViscousFlux nu grad^3 u_h * rho u] */
    for (int d = 0; d < 3; ++d) {
        const double Dt = nu[d] * m->invPr;
        double v = (tau[d] * Q[1] + tau[d + 3] * Q[2] + tau[d + 6] * Q[3]) + (-Dt * gf[d]) * rho;
        if (m->hyper) {
            v = v + hyp[9 + d] * rho;
            v = v + (hyp[d + 0] * Q[1] + hyp[d + 3] * Q[2] + hyp[d + 6] * Q[3]);
        }
        F[d + 12] = v;
    }
}

/* Held-Suarez forcing coefficients (heldsuarez.jl:116-155) */
static void hs_coeffs(const atmos_t *m, const double *Q, const double *aux, double T, double *k_v,
                      double *k_T, double *T_equil)
{
    const double day = m->day;
    const double k_a = 1 / (40 * day), k_f = 1 / day, k_s = 1 / (4 * day);
    const double dTy = 60, dthz = 10, T_eq = 315, T_min = 200, sig_b = 7.0 / 10;
    const double *x = aux;
    const double phi = asin(x[2] / sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]));
    const double p = air_p(m, T, Q[0]);
    const double sig = p / m->MSLP;
    const double exner = pow(sig, m->R_d / m->cp_d);
    const double dsig = (sig - sig_b) / (1 - sig_b);
    const double hf = dsig > 0 ? dsig : 0;
    const double s = sin(phi), c = cos(phi);
    double Te = (T_eq - dTy * (s * s) - dthz * log(sig) * (c * c)) * exner;
    Te = Te > T_min ? Te : T_min;
    *T_equil = Te;
    *k_T = k_a + (k_s - k_a) * hf * ((c * c) * (c * c));
    *k_v = k_f * hf;
}

/* ---- manufactured solution of test/Numerics/DGMethods/compressible_Navier_Stokes/
   mms_bc_atmos.jl, dim = 3 (generated by mms_solution.jl:10-100):
     rho = c g + 3,  u = v = c g,  w = c h,  E = c g + 100,
     c = cos(pi t), g = sin(pi x) cos(pi y) cos(pi z), h = sin(pi x) cos(pi y) sin(pi z),
     P = (gamma - 1)(E - rho |u|^2 / 2), tau = 2 mu (eps - tr(eps)/3 I), no heat conduction.
   The source S = dq/dt + div F is assembled from the analytic derivatives of g and h. */
static void mms_state(double t, const double *x, double *Q)
{
    const double c = cos(M_PI * t);
    const double g = sin(M_PI * x[0]) * cos(M_PI * x[1]) * cos(M_PI * x[2]);
    const double h = sin(M_PI * x[0]) * cos(M_PI * x[1]) * sin(M_PI * x[2]);
    const double rho = g * c + 3;
    Q[0] = rho;
    Q[1] = rho * g * c;
    Q[2] = rho * g * c;
    Q[3] = rho * h * c;
    Q[4] = g * c + 100;
}
static void mms_source(const atmos_t *m, double t, const double *x, double *S)
{
    const double gam = m->cp_d / m->cv_d, mu = m->visc, P2 = M_PI * M_PI;
    const double ct = cos(M_PI * t), st = sin(M_PI * t);
    const double sx = sin(M_PI * x[0]), cx = cos(M_PI * x[0]);
    const double sy = sin(M_PI * x[1]), cy = cos(M_PI * x[1]);
    const double sz = sin(M_PI * x[2]), cz = cos(M_PI * x[2]);
    const double g = sx * cy * cz, h = sx * cy * sz;
    /* f[0] = g (for u, v, rho, E), f[1] = h (for w): value, gradient, Hessian */
    const double df[2][3] = {{M_PI * cx * cy * cz, -M_PI * sx * sy * cz, -M_PI * sx * cy * sz},
                             {M_PI * cx * cy * sz, -M_PI * sx * sy * sz, M_PI * sx * cy * cz}};
    const double Hf[2][3][3] = {
        {{-P2 * g, -P2 * cx * sy * cz, -P2 * cx * cy * sz},
         {-P2 * cx * sy * cz, -P2 * g, P2 * sx * sy * sz},
         {-P2 * cx * cy * sz, P2 * sx * sy * sz, -P2 * g}},
        {{-P2 * h, -P2 * cx * sy * sz, P2 * cx * cy * cz},
         {-P2 * cx * sy * sz, -P2 * h, -P2 * sx * sy * cz},
         {P2 * cx * cy * cz, -P2 * sx * sy * cz, -P2 * h}}};
    const int which[3] = {0, 0, 1};
    const double fv[2] = {g, h};
    const double rho = ct * g + 3, rho_t = -M_PI * st * g, E = ct * g + 100, E_t = -M_PI * st * g;
    double u[3], u_t[3], du[3][3], lap[3], ddiv[3], drho[3], dE[3];
    for (int i = 0; i < 3; ++i) {
        u[i] = ct * fv[which[i]];
        u_t[i] = -M_PI * st * fv[which[i]];
        for (int j = 0; j < 3; ++j) du[i][j] = ct * df[which[i]][j];
        lap[i] = ct * (Hf[which[i]][0][0] + Hf[which[i]][1][1] + Hf[which[i]][2][2]);
        drho[i] = ct * df[0][i];
        dE[i] = ct * df[0][i];
    }
    for (int j = 0; j < 3; ++j) { /* d_j (d_i u_i) */
        ddiv[j] = 0;
        for (int i = 0; i < 3; ++i) ddiv[j] += ct * Hf[which[i]][i][j];
    }
    const double divu = du[0][0] + du[1][1] + du[2][2];
    const double ke = (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) / 2;
    double dke[3], dP[3];
    for (int j = 0; j < 3; ++j) dke[j] = u[0] * du[0][j] + u[1] * du[1][j] + u[2] * du[2][j];
    const double P = (gam - 1) * (E - rho * ke);
    for (int j = 0; j < 3; ++j) dP[j] = (gam - 1) * (dE[j] - drho[j] * ke - rho * dke[j]);
    const double divm = (u[0] * drho[0] + u[1] * drho[1] + u[2] * drho[2]) + rho * divu;
    S[0] = rho_t + divm;
    double dtau[3], work = 0;
    for (int i = 0; i < 3; ++i) {
        dtau[i] = mu * (lap[i] + ddiv[i] / 3);
        const double adv = u[0] * du[i][0] + u[1] * du[i][1] + u[2] * du[i][2];
        S[1 + i] = rho_t * u[i] + rho * u_t[i] + u[i] * divm + rho * adv + dP[i] - dtau[i];
        work += u[i] * dtau[i];
        for (int j = 0; j < 3; ++j)
            work += du[i][j] * (mu * (du[i][j] + du[j][i]) - (i == j ? 2 * mu / 3 * divu : 0.0));
    }
    S[4] = E_t + (E + P) * divu +
           (u[0] * (dE[0] + dP[0]) + u[1] * (dE[1] + dP[1]) + u[2] * (dE[2] + dP[2])) - work;
}

static void at_source(const void *p_, double *S, const double *Q, const double *gf, const double *aux,
                      double t, int dir)
{
    const atmos_t *m = (const atmos_t *)p_;
    (void)gf; (void)dir;
    const double rho = Q[0];
    double Sm[3] = {0, 0, 0}, Se = 0;
    int first = 1, firste = 1;
    double T = 0, k_v = 0, k_T = 0, Te = 0;
    if (m->src & 4) {
        T = air_T(m, internal_energy(m, Q, aux));
        hs_coeffs(m, Q, aux, T, &k_v, &k_T, &Te);
    }
    if (m->src & 1) { /* Gravity */
        const double r = (m->ref && m->subtract) ? rho - aux[m->oRef] : rho;
        for (int d = 0; d < 3; ++d) {
            const double v = -r * aux[m->oPhi + 1 + d];
            Sm[d] = first ? v : Sm[d] + v;
        }
        first = 0;
    }
    if (m->src & 2) { /* Coriolis: -(0,0,2 Omega) x rho u */
        const double w = 2 * m->Omega;
        const double c[3] = {-(0 * Q[3] - w * Q[2]), -(w * Q[1] - 0 * Q[3]), -(0 * Q[2] - 0 * Q[1])};
        for (int d = 0; d < 3; ++d) Sm[d] = first ? c[d] : Sm[d] + c[d];
        first = 0;
    }
    if (m->src & 4) { /* HeldSuarezForcing */
        double k[3];
        for (int d = 0; d < 3; ++d) k[d] = aux[m->oPhi + 1 + d] / m->grav;
        const double kn = k[0] * Q[1] + k[1] * Q[2] + k[2] * Q[3];
        for (int d = 0; d < 3; ++d) {
            const double v = -k_v * (Q[1 + d] - k[d] * kn);
            Sm[d] = first ? v : Sm[d] + v;
        }
        first = 0;
        const double ve = -k_T * rho * m->cv_d * (T - Te);
        Se = firste ? ve : Se + ve;
        firste = 0;
    }
    S[0] = 0;
    S[1] = Sm[0];
    S[2] = Sm[1];
    S[3] = Sm[2];
    S[4] = Se;
    if (m->src & 8) { /* MMSSource{3} (mms_bc_atmos.jl:65-86) */
        double Sx[5];
        mms_source(m, t, aux, Sx);
        S[0] = Sx[0];
        for (int q = 1; q < 4; ++q) S[q] = first ? Sx[q] : S[q] + Sx[q];
        S[4] = firste ? Sx[4] : S[4] + Sx[4];
    }
}

static void at_gradarg(const void *p_, double *G, const double *Q, const double *aux, double t)
{
    const atmos_t *m = (const atmos_t *)p_;
    (void)t;
    const double rhoinv = 1 / Q[0];
    for (int d = 0; d < 3; ++d) G[d] = rhoinv * Q[1 + d];
    const double T = air_T(m, internal_energy(m, Q, aux));
    const double e_tot = Q[4] * (1 / Q[0]);
    G[3] = m->zero_h ? 0.0 : e_tot + m->R_d * T;
    if (m->smag) G[4] = aux[m->oMoist]; /* transform.turbulence.theta_v = aux.moisture.theta_v */
    if (m->hyper) {
        double u[3], k[3];
        for (int d = 0; d < 3; ++d) u[d] = Q[1 + d] * rhoinv;
        for (int d = 0; d < 3; ++d) k[d] = aux[m->oPhi + 1 + d] / m->grav;
        for (int i = 0; i < 3; ++i) { /* (SDiagonal(1,1,1) - k k') * u */
            double acc = 0;
            for (int j = 0; j < 3; ++j) {
                const double Pij = (i == j ? 1.0 : 0.0) - k[i] * k[j];
                acc = j == 0 ? Pij * u[j] : acc + Pij * u[j];
            }
            G[4 + i] = acc;
        }
        G[7] = G[3];
    }
}

static void at_gradflux(const void *p_, double *gf, const double *g, const double *Q, const double *aux,
                        double t)
{
    const atmos_t *m = (const atmos_t *)p_;
    (void)Q; (void)t;
    if (m->smag) /* N^2 = dot(grad theta_v, grad Phi) / aux.moisture.theta_v  (:451-466) */
        gf[9] = (g[0 + 3 * 4] * aux[m->oPhi + 1] + g[1 + 3 * 4] * aux[m->oPhi + 2] +
                 g[2 + 3 * 4] * aux[m->oPhi + 3]) / aux[m->oMoist];
    /* energy: grad h_tot ; turbulence: S = symmetrize(grad u) */
    for (int d = 0; d < 3; ++d) gf[d] = g[d + 3 * 3];
    /* grad u is 3x3 with g[d + 3*c] = d u_c / d x_d ; symmetrize(A) = (A + A')/2 lower */
    gf[3] = g[0 + 3 * 0];
    gf[4] = (g[1 + 3 * 0] + g[0 + 3 * 1]) / 2;
    gf[5] = (g[2 + 3 * 0] + g[0 + 3 * 2]) / 2;
    gf[6] = g[1 + 3 * 1];
    gf[7] = (g[2 + 3 * 1] + g[1 + 3 * 2]) / 2;
    gf[8] = g[2 + 3 * 2];
}

static void at_postlap(const void *p_, double *hyp, const double *gl, const double *Q, const double *aux,
                       double t)
{
    const atmos_t *m = (const atmos_t *)p_;
    (void)Q; (void)t;
    if (!m->hyper) return;
    const double h = aux[m->oDelta] / 2;
    const double nu4 = (h * h) * (h * h) / 2 / m->tau;
    for (int q = 0; q < 9; ++q) hyp[q] = nu4 * gl[q];           /* nu grad^3 u_h (3x3) */
    for (int d = 0; d < 3; ++d) hyp[9 + d] = nu4 * gl[d + 9];    /* nu grad^3 h_tot */
}

static void at_wavespeed(const void *p_, double *ws, const double *n, const double *Q, const double *aux,
                         double t, int facedir)
{
    const atmos_t *m = (const atmos_t *)p_;
    (void)t; (void)facedir;
    const double rhoinv = 1 / Q[0];
    const double uN = fabs(n[0] * (rhoinv * Q[1]) + n[1] * (rhoinv * Q[2]) + n[2] * (rhoinv * Q[3]));
    const double ss = soundspeed(m, air_T(m, internal_energy(m, Q, aux)));
    for (int s = 0; s < 5; ++s) ws[s] = uN + ss;
}

static void moist_update(const atmos_t *m, const double *Q, double *aux)
{ /* DryModel atmos_nodal_update_auxiliary_state! (moisture.jl:53-62) */
    const double T = air_T(m, internal_energy(m, Q, aux));
    const double p = air_p(m, T, Q[0]);
    const double exner = pow(p / m->MSLP, m->R_d / m->cp_d);
    aux[m->oMoist] = m->R_d / m->R_d * (T / exner);
    aux[m->oMoist + 1] = T;
}

static void at_bstate(const void *p_, int kind, int bctag, double *QP, double *auxP, const double *n,
                      const double *QM, const double *auxM, double t, const double *Q1,
                      const double *aux1)
{
    const atmos_t *m = (const atmos_t *)p_;
    (void)auxM; (void)Q1; (void)aux1;
    if (m->bc[bctag - 1] == 2) { /* InitStateBC (bc_initstate.jl:12-26) */
        mms_state(t, auxP, QP);
        return;
    }
    if (m->bc[bctag - 1] == 1) {
        const double dn = QM[1] * n[0] + QM[2] * n[1] + QM[3] * n[2];
        const double f = kind == ORC_BS_FIRST ? 2 * dn : dn;
        for (int d = 0; d < 3; ++d) QP[1 + d] -= f * n[d];
    }
    moist_update(m, QP, auxP);
}
/* normal_boundary_flux_second_order! for AtmosBC (boundaryconditions.jl:101-131): FreeSlip and
   Insulating add nothing */
static void at_bflux2(const void *p_, int bctag, double *F, double *QP, double *gfP, double *hypP,
                      double *auxP, const double *n, const double *QM, const double *gfM,
                      const double *hypM, const double *auxM, double t, const double *Q1,
                      const double *gf1, const double *aux1)
{
    const atmos_t *m = (const atmos_t *)p_;
    (void)n; (void)Q1; (void)gf1; (void)aux1;
    if (m->bc[bctag - 1] != 2) return;
    /* InitStateBC: generic boundary_flux_second_order! (NumericalFluxes.jl:925-967) --
       boundary_state! puts the exact solution on the plus side (bc_initstate.jl:28-46), the
       gradient flux there is the copy of the minus side, then flux_second_order! of the plus side */
    (void)QM; (void)gfM; (void)hypM; (void)auxM;
    mms_state(t, auxP, QP);
    double FP[15];
    for (int q = 0; q < 15; ++q) FP[q] = -0.0;
    at_flux2(p_, FP, QP, gfP, hypP, auxP, t);
    for (int q = 0; q < 15; ++q) F[q] += FP[q];
}
static void at_bdiv(const void *p_, int bctag, double *gradP, double *auxP, const double *n,
                    const double *gradM, const double *auxM, double t)
{
    (void)p_; (void)bctag; (void)gradP; (void)auxP; (void)n; (void)gradM; (void)auxM; (void)t;
}
static void at_bhigher(const void *p_, int bctag, double *QP, double *auxP, double *lapP, const double *n,
                       const double *QM, const double *auxM, const double *lapM, double t)
{
    (void)p_; (void)bctag; (void)QP; (void)auxP; (void)lapP; (void)n; (void)QM; (void)auxM; (void)lapM; (void)t;
}
static void at_update_aux(const void *p_, const double *Q, double *aux, double t)
{
    (void)t;
    moist_update((const atmos_t *)p_, Q, aux);
}

/* src/Atmos/Model/courant.jl:12-83 */
static double at_courant(const void *p_, int kind, const double *Q, const double *aux,
                         const double *gf, double dx, double dt, double t, int direction)
{
    const atmos_t *m = (const atmos_t *)p_;
    (void)t;
    double k[3] = {0, 0, 0};
    if (m->orient)
        for (int d = 0; d < 3; ++d) k[d] = aux[m->oPhi + 1 + d] / m->grav;
    if (kind == 2) { /* diffusive_courant: norm_nu (courant.jl:19-24) */
        double nu[3], tau[9], normnu;
        turbulence_tensors(m, Q, gf, aux, nu, tau);
        if (!m->smag) {
            normnu = nu[0]; /* nu::Real */
        } else {
            const double dk = nu[0] * k[0] + nu[1] * k[1] + nu[2] * k[2];
            if (direction == ORC_VERTICAL) {
                normnu = dk;
            } else {
                double v[3];
                for (int d = 0; d < 3; ++d)
                    v[d] = direction == ORC_HORIZONTAL ? nu[d] - dk * k[d] : nu[d];
                normnu = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            }
        }
        return dt * normnu / (dx * dx);
    }
    double normu;
    const double dotk = Q[1] * k[0] + Q[2] * k[1] + Q[3] * k[2];
    if (direction == ORC_VERTICAL) {
        normu = fabs(dotk) / Q[0];
    } else if (direction == ORC_HORIZONTAL) {
        double v[3];
        for (int d = 0; d < 3; ++d) v[d] = (Q[1 + d] - dotk * k[d]) / Q[0];
        normu = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    } else {
        double v[3];
        for (int d = 0; d < 3; ++d) v[d] = Q[1 + d] / Q[0];
        normu = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    }
    if (kind == 0) return dt * normu / dx;
    const double ss = soundspeed(m, air_T(m, internal_energy(m, Q, aux)));
    return dt * (normu + ss) / dx;
}

/* RoeNumericalFlux (AtmosModel.jl:1003-1130, DryModel only), HLLCNumericalFlux
 * (:1154-1276) and LMARSNumericalFlux (:1515-1600) */
static inline double roe_average(double sM, double sP, double vM, double vP)
{
    return (sM * vM + sP * vP) / (sM + sP);
}
static void at_nf_law(const void *p_, int nf, double *fluxn, const double *n, const double *QM,
                      const double *auxM, const double *QP, const double *auxP, double t, int dir)
{
    const atmos_t *m = (const atmos_t *)p_;
    double FM[15], FP[15], fnM[5], fnP[5];
    for (int i = 0; i < 15; ++i) FM[i] = FP[i] = -0.0;
    at_flux1(p_, FM, QM, auxM, t, dir);
    at_flux1(p_, FP, QP, auxP, t, dir);
    const double rM = QM[0], rP = QP[0];
    double uM[3], uP[3];
    for (int d = 0; d < 3; ++d) { uM[d] = QM[1 + d] / rM; uP[d] = QP[1 + d] / rP; }
    const double TM = air_T(m, internal_energy(m, QM, auxM)), TP = air_T(m, internal_energy(m, QP, auxP));
    const double pM = air_p(m, TM, rM), pP = air_p(m, TP, rP);
    const double cM = soundspeed(m, TM), cP = soundspeed(m, TP);
    const double unM = uM[0] * n[0] + uM[1] * n[1] + uM[2] * n[2];
    const double unP = uP[0] * n[0] + uP[1] * n[1] + uP[2] * n[2];
    if (nf == ORC_NF_LMARS) { /* AtmosModel.jl:1515-1600, beta = 1 */
        double ppM = pM, ppP = pP;
        if (m->ref && m->subtract) { ppM -= auxM[m->oRef + 1]; ppP -= auxP[m->oRef + 1]; }
        const double hM = m->zero_h ? 0.0 : QM[4] / rM + m->R_d * TM;
        const double hP = m->zero_h ? 0.0 : QP[4] / rP + m->R_d * TP;
        const double beta = 1.0;
        const double u_half = 1.0 / 2 * (unP + unM) - beta * 1 / (rM + rP) / cM * (ppP - ppM);
        const double p_half = 1.0 / 2 * (ppP + ppM) - beta * ((rM + rP) * cM) / 4 * (unP - unM);
        const int up = u_half > 0;
        fluxn[0] += (up ? rM : rP) * u_half;
        for (int d = 0; d < 3; ++d) fluxn[1 + d] += (up ? QM[1 + d] : QP[1 + d]) * u_half + p_half * n[d];
        fluxn[4] += (up ? rM * hM : rP * hP) * u_half;
        return;
    }
    if (nf == ORC_NF_ROE) {
        /* central part (NumericalFluxes.jl:300-340) */
        const double nh[3] = {n[0] / 2, n[1] / 2, n[2] / 2};
        for (int s = 0; s < 5; ++s)
            fluxn[s] += (FM[3 * s] + FP[3 * s]) * nh[0] + (FM[3 * s + 1] + FP[3 * s + 1]) * nh[1] +
                        (FM[3 * s + 2] + FP[3 * s + 2]) * nh[2];
        const double Phi = m->orient ? auxM[m->oPhi] : 0.0;
        const double eM = QM[4] / rM, eP = QP[4] / rP;
        /* total_specific_enthalpy(ts, e_tot) = e_tot + R_m T */
        const double hM = m->zero_h ? 0.0 : eM + m->R_d * TM;
        const double hP = m->zero_h ? 0.0 : eP + m->R_d * TP;
        const double sM = sqrt(rM), sP = sqrt(rP);
        const double rt = sqrt(rM * rP);
        double ut[3], du[3];
        for (int d = 0; d < 3; ++d) { ut[d] = roe_average(sM, sP, uM[d], uP[d]); du[d] = uP[d] - uM[d]; }
        const double ht = roe_average(sM, sP, hM, hP);
        const double ct = sqrt(roe_average(sM, sP, cM * cM, cP * cP));
        const double utn = ut[0] * n[0] + ut[1] * n[1] + ut[2] * n[2];
        const double dr = rP - rM, dp = pP - pM;
        const double dun = du[0] * n[0] + du[1] * n[1] + du[2] * n[2];
        const double w1 = fabs(utn - ct) * (dp - rt * ct * dun) / (2 * (ct * ct));
        const double w2 = fabs(utn + ct) * (dp + rt * ct * dun) / (2 * (ct * ct));
        const double w3 = fabs(utn) * (dr - dp / (ct * ct));
        const double w4 = fabs(utn) * rt;
        fluxn[0] -= (w1 + w2 + w3) / 2;
        for (int d = 0; d < 3; ++d)
            fluxn[1 + d] -= (w1 * (ut[d] - ct * n[d]) + w2 * (ut[d] + ct * n[d]) + w3 * ut[d] +
                             w4 * (du[d] - dun * n[d])) / 2;
        const double utut = ut[0] * ut[0] + ut[1] * ut[1] + ut[2] * ut[2];
        const double utdu = ut[0] * du[0] + ut[1] * du[1] + ut[2] * du[2];
        fluxn[4] -= (w1 * (ht - ct * utn) + w2 * (ht + ct * utn) +
                     w3 * (utut / 2 + Phi - m->T_0 * m->cv_d) + w4 * (utdu - utn * dun)) / 2;
        return;
    }
    /* HLLC: flux' * n */
    for (int s = 0; s < 5; ++s) {
        fnM[s] = FM[3 * s] * n[0] + FM[3 * s + 1] * n[1] + FM[3 * s + 2] * n[2];
        fnP[s] = FP[3 * s] * n[0] + FP[3 * s + 1] * n[1] + FP[3 * s + 2] * n[2];
    }
    const double SM = fmin(unM - cM, unP - cP), SP = fmax(unM + cM, unP + cP);
    const double S0 = (pP - pM + rM * unM * (SM - unM) - rP * unP * (SP - unP)) /
                      (rM * (SM - unM) - rP * (SP - unP));
    const double p0 = (pP + pM + rM * (SM - unM) * (S0 - unM) + rP * (SP - unP) * (S0 - unP)) / 2;
    double mp = p0;
    if (m->ref && m->subtract) mp = p0 - (auxM[m->oRef + 1] + auxP[m->oRef + 1]) / 2;
    const double pD[5] = {0.0, mp * n[0], mp * n[1], mp * n[2], p0 * S0};
    if (0 <= SM) {
        for (int s = 0; s < 5; ++s) fluxn[s] += fnM[s];
    } else if (SM < 0 && 0 <= S0) {
        for (int s = 0; s < 5; ++s) fluxn[s] += (S0 * (SM * QM[s] - fnM[s]) + SM * pD[s]) / (SM - S0);
    } else if (S0 < 0 && 0 <= SP) {
        for (int s = 0; s < 5; ++s) fluxn[s] += (S0 * (SP * QP[s] - fnP[s]) + SP * pD[s]) / (SP - S0);
    } else {
        for (int s = 0; s < 5; ++s) fluxn[s] += fnP[s];
    }
}

orc_physics *orc_atmos_new(const int *ip, const double *dp, int nf_first)
{
    orc_physics *ph = (orc_physics *)calloc(1, sizeof(orc_physics));
    atmos_t *m = (atmos_t *)calloc(1, sizeof(atmos_t));
    m->orient = ip[0]; m->ref = ip[1]; m->subtract = ip[2]; m->kinematic = ip[3];
    m->hyper = ip[4]; m->src = ip[5]; m->nbc = ip[6];
    for (int i = 0; i < 7; ++i) m->bc[i] = ip[7 + i];
    m->visc = dp[0]; m->tau = dp[1];
    m->R_d = dp[2]; m->cp_d = dp[3]; m->cv_d = dp[4]; m->T_0 = dp[5]; m->grav = dp[6];
    m->Omega = dp[7]; m->MSLP = dp[8]; m->day = dp[9]; m->a = dp[10]; m->invPr = dp[11];
    m->kappa = dp[12];
    m->C_smag = dp[13];
    m->smag = ip[14] == 1;
    m->withdiv = ip[15] & 1;        /* WithDivergence (TurbulenceClosures.jl:369-370) */
    m->zero_h = (ip[15] >> 1) & 1;  /* total_specific_enthalpy == 0 (mms_bc_atmos.jl:50-51) */
    int o = 3;
    m->oPhi = o;   o += m->orient ? 4 : 0;
    m->oRef = o;   o += m->ref ? 7 : 0;
    m->oTurb = o;  o += m->smag ? 1 : 0;
    m->oDelta = o; o += m->hyper ? 1 : 0;
    m->oMoist = o; o += 2;
    ph->ns = 5;
    ph->naux = o;
    ph->ngrad = 4 + (m->smag ? 1 : 0) + (m->hyper ? 4 : 0);
    ph->ngf = 9 + (m->smag ? 1 : 0);
    ph->ngl = m->hyper ? 4 : 0;
    ph->nhyp = m->hyper ? 12 : 0;
    for (int s = 0; s < 4; ++s) ph->hv_indexmap[s] = 4 + (m->smag ? 1 : 0) + s;
    ph->nf_first = nf_first;
    ph->p = m;
    ph->flux_first_order = at_flux1;
    ph->flux_second_order = at_flux2;
    ph->source = at_source;
    ph->gradient_argument = at_gradarg;
    ph->gradient_flux = at_gradflux;
    ph->post_gradient_laplacian = at_postlap;
    ph->wavespeed = at_wavespeed;
    ph->boundary_state = at_bstate;
    ph->boundary_flux_second_order = at_bflux2;
    ph->boundary_state_divergence = at_bdiv;
    ph->boundary_state_higher_order = at_bhigher;
    ph->update_aux = at_update_aux;
    ph->courant = at_courant;
    ph->numerical_flux_law = at_nf_law;
    return ph;
}
