/* TEST INFRASTRUCTURE -- CPU restatement, not shipped.
 *
 * Moist LES configuration of the AtmosModel: TotalEnergyModel + EquilMoist, FlatOrientation,
 * HydrostaticState (subtract_off), closure ConstantKinematic/DynamicViscosity (0),
 * SmagorinskyLilly (1) or AnisoMinDiss (2), source Gravity, default AtmosBC.  Restated from
 *   src/Atmos/Model/AtmosModel.jl:397-520 (layouts), :625-690, :808-828,
 *   tendencies_{mass,momentum,energy,moisture}.jl, atmos_tendencies.jl (term order),
 *   moisture.jl:70-115 (EquilMoist), thermo_states.jl (new_thermo_state: PhaseEquil from
 *   (e_int, rho, q_tot) -- in this snapshot recover_thermo_state calls it too, :40-60),
 *   src/Common/TurbulenceClosures/TurbulenceClosures.jl:411-497 (SmagorinskyLilly), :600-690
 *   (AnisoMinDiss).
 * Thermodynamics.jl 0.3.2 is not in the reference tree: the moist formulas (gas constants and
 * heat capacities of the mixture, internal energy, saturation vapour pressure over liquid / ice,
 * liquid fraction, saturation adjustment by Newton's method on e_int_sat(T) - e_int) restate its
 * published formulation.  PARITY UNPINNED for saturated states: no file in the reference holds a
 * number that exercises them.  With q_tot = 0 every formula reduces, operation by operation, to
 * the dry ones of physics_atmos.c (checked bit for bit in tests/test_moist_oracle.py).
 *
 * State rho, rho u[3], rho e, rho q_tot.  Auxiliary coord[3], Phi, grad Phi[3], ref_state[7],
 * Delta, moisture (temperature, theta_v, q_liq, q_ice).  Gradient u[3], h_tot, theta_v, q_tot.
 * Gradient flux grad h_tot[3], S[6] | grad u[9], N^2, grad q_tot[3].
 *
 * iparam[0] closure, [1] subtract_off, [2] kinematic viscosity (closure 0), [3] maxiter,
 * [5] sources (1 Gravity, 2 BomexTendencies, 4 BomexSponge, 8 BomexGeostrophic), [6] nbc,
 * [7..13] bc kinds (1 = AtmosBC(); 2 = the BOMEX surface: Impenetrable(DragLaw(u_star)),
 * PrescribedEnergyFlux, PrescribedMoistureFlux).  dparam[32..49]: the BOMEX constants.
 * dparam[0] viscosity | C_smag | C_poincare, [2..9] R_d cp_d cv_d T_0 grav MSLP inv_Pr_turb
 * tolerance, [16..26] R_v cp_v cp_l cp_i LH_v0 LH_s0 T_triple T_freeze T_icenuc press_triple T_min.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "dg_oracle.h"

typedef struct {
    int closure, subtract, kinematic, maxiter, src, nbc, bc[8];
    double visc, R_d, cp_d, cv_d, T_0, grav, MSLP, invPr, tol;
    double R_v, cp_v, cp_l, cp_i, LH_v0, LH_s0, T_triple, T_freeze, T_icenuc, p_triple, T_min;
    int ngt; /* turbulence entries of the gradient flux: 7 (S, N^2) or 10 (grad u, N^2) */
    /* BOMEX (experiments/AtmosLES/bomex_model.jl:76-246, 352-470): surface fluxes and sources */
    double u_star, e_flux, q_flux, f_cor, u_geo, u_slope, v_geo, z_sponge, a_max, gam, z_max;
    double dqt_peak, zl_m, zh_m, dth_peak, zl_sub, zh_sub, w_sub;
} moist_t;
enum { OPHI = 3, OREF = 7, OTURB = 14, OMOIST = 15, NAUXM = 19 };

typedef struct { double T, q_tot, q_liq, q_ice, R_m, cv_m, cp_m, e_int; } thermo_t;

/* ---- Thermodynamics.jl: mixture properties ------------------------------------------- */
static inline double gas_constant_air(const moist_t *m, double qt, double ql, double qi)
{
    const double eps = m->R_v / m->R_d; /* molmass_ratio */
    return m->R_d * (1 + (eps - 1) * qt - eps * (ql + qi));
}
static inline double cv_mix(const moist_t *m, double qt, double ql, double qi)
{
    const double cv_v = m->cp_v - m->R_v;
    return m->cv_d + (cv_v - m->cv_d) * qt + (m->cp_l - cv_v) * ql + (m->cp_i - cv_v) * qi;
}
static inline double cp_mix(const moist_t *m, double qt, double ql, double qi)
{
    return m->cp_d + (m->cp_v - m->cp_d) * qt + (m->cp_l - m->cp_v) * ql + (m->cp_i - m->cp_v) * qi;
}
static inline double e_int_v0(const moist_t *m) { return m->LH_v0 - m->R_v * m->T_0; }
static inline double e_int_i0(const moist_t *m) { return m->LH_s0 - m->LH_v0; }
static inline double internal_energy_T(const moist_t *m, double T, double qt, double ql, double qi)
{
    return cv_mix(m, qt, ql, qi) * (T - m->T_0) + (qt - ql) * e_int_v0(m) -
           qi * (e_int_v0(m) + e_int_i0(m));
}
static inline double air_temperature(const moist_t *m, double e_int, double qt, double ql, double qi)
{
    return m->T_0 + (e_int - (qt - ql) * e_int_v0(m) + qi * (e_int_v0(m) + e_int_i0(m))) /
                        cv_mix(m, qt, ql, qi);
}
static inline double liquid_fraction(const moist_t *m, double T)
{ /* PhaseEquil: power law between T_icenuc and T_freeze (pow_icenuc = 1) */
    if (T > m->T_freeze) return 1.0;
    if (T > m->T_icenuc) return (T - m->T_icenuc) / (m->T_freeze - m->T_icenuc);
    return 0.0;
}
static inline double saturation_vapor_pressure(const moist_t *m, double T, double LH_0, double dcp)
{ /* Clausius-Clapeyron with constant heat capacities, integrated from the triple point */
    return m->p_triple * pow(T / m->T_triple, dcp / m->R_v) *
           exp((LH_0 - dcp * m->T_0) / m->R_v * (1 / m->T_triple - 1 / T));
}
static inline double q_vap_saturation(const moist_t *m, double T, double rho)
{
    const double lam = liquid_fraction(m, T);
    const double LH_0 = lam * m->LH_v0 + (1 - lam) * m->LH_s0;
    const double dcp = lam * (m->cp_v - m->cp_l) + (1 - lam) * (m->cp_v - m->cp_i);
    return saturation_vapor_pressure(m, T, LH_0, dcp) / (rho * m->R_v * T);
}
static inline void phase_partition_equil(const moist_t *m, double T, double rho, double qt, double *ql,
                                         double *qi)
{
    const double qvs = q_vap_saturation(m, T, rho);
    const double qc = qt - qvs > 0 ? qt - qvs : 0.0;
    const double lam = liquid_fraction(m, T);
    *ql = lam * qc;
    *qi = (1 - lam) * qc;
}
static double saturation_adjustment(const moist_t *m, double e_int, double rho, double qt, int *unsat)
{
    double T = air_temperature(m, e_int, qt, 0.0, 0.0); /* all vapour */
    if (T < m->T_min) T = m->T_min;
    *unsat = qt <= q_vap_saturation(m, T, rho) && T > m->T_min;
    if (*unsat) return T;
    for (int it = 0; it < m->maxiter; ++it) { /* Newton on e_int_sat(T) - e_int */
        double ql, qi;
        phase_partition_equil(m, T, rho, qt, &ql, &qi);
        const double f = internal_energy_T(m, T, qt, ql, qi) - e_int;
        const double lam = liquid_fraction(m, T);
        const double qvs = q_vap_saturation(m, T, rho);
        const double L = lam * m->LH_v0 + (1 - lam) * m->LH_s0;
        const double dlam = (T > m->T_icenuc && T < m->T_freeze) ? 1 / (m->T_freeze - m->T_icenuc) : 0.0;
        const double dqvs = qvs * L / (m->R_v * T * T);
        const double cv_v = m->cp_v - m->R_v;
        const double dcvm = cv_v - lam * m->cp_l - (1 - lam) * m->cp_i;
        const double fp = cv_mix(m, qt, ql, qi) +
                          (e_int_v0(m) + (1 - lam) * e_int_i0(m) + (T - m->T_0) * dcvm) * dqvs +
                          (ql + qi) * e_int_i0(m) * dlam;
        const double dT = f / fp;
        T -= dT;
        if (fabs(dT) < m->tol) break;
    }
    return T;
}

static inline double e_pot_of(const double *aux) { return aux[OPHI]; }
static inline double internal_energy(const double *Q, const double *aux)
{
    const double rho = Q[0];
    const double rhoinv = 1 / rho;
    const double rhoe_kin = rhoinv * (Q[1] * Q[1] + Q[2] * Q[2] + Q[3] * Q[3]) / 2;
    const double rhoe_pot = rho * e_pot_of(aux);
    const double rhoe_int = Q[4] - rhoe_kin - rhoe_pot;
    return rhoinv * rhoe_int;
}
/* new_thermo_state(atmos, ::TotalEnergyModel, ::EquilMoist, state, aux) (thermo_states.jl) */
static void thermo_state(const moist_t *m, const double *Q, const double *aux, thermo_t *ts)
{
    ts->e_int = internal_energy(Q, aux);
    ts->q_tot = Q[5] / Q[0];
    int unsat;
    ts->T = saturation_adjustment(m, ts->e_int, Q[0], ts->q_tot, &unsat);
    if (unsat) { /* q_c = max(q_tot - q_vap_sat, 0) = 0 */
        ts->q_liq = 0.0;
        ts->q_ice = 0.0;
    } else {
        phase_partition_equil(m, ts->T, Q[0], ts->q_tot, &ts->q_liq, &ts->q_ice);
    }
    ts->R_m = gas_constant_air(m, ts->q_tot, ts->q_liq, ts->q_ice);
    ts->cv_m = cv_mix(m, ts->q_tot, ts->q_liq, ts->q_ice);
    ts->cp_m = cp_mix(m, ts->q_tot, ts->q_liq, ts->q_ice);
}
static inline double air_pressure(const thermo_t *ts, double rho) { return ts->R_m * rho * ts->T; }
static inline double soundspeed(const thermo_t *ts)
{
    const double gamma = ts->cp_m / ts->cv_m;
    return sqrt(gamma * ts->R_m * ts->T);
}
static inline double virtual_pottemp(const moist_t *m, const thermo_t *ts, double rho)
{
    const double exner = pow(air_pressure(ts, rho) / m->MSLP, ts->R_m / ts->cp_m);
    return ts->R_m / m->R_d * (ts->T / exner);
}

/* ---- first-order fluxes ---------------------------------------------------------------- */
static void mo_flux1(const void *p_, double *F, const double *Q, const double *aux, double t, int dir)
{
    const moist_t *m = (const moist_t *)p_;
    (void)t; (void)dir;
    thermo_t ts;
    thermo_state(m, Q, aux, &ts);
    const double rho = Q[0];
    const double p = air_pressure(&ts, rho);
    double u[3];
    for (int d = 0; d < 3; ++d) u[d] = Q[1 + d] / rho;
    for (int d = 0; d < 3; ++d) F[d] = Q[1 + d];
    const double pp = m->subtract ? p - aux[OREF + 1] : p;
    for (int c = 0; c < 3; ++c)
        for (int d = 0; d < 3; ++d)
            F[d + 3 * (1 + c)] = Q[1 + d] * u[c] + (0.0 + (d == c ? pp : 0.0));
    for (int d = 0; d < 3; ++d) F[d + 12] = u[d] * Q[4] + u[d] * p;
    for (int d = 0; d < 3; ++d) F[d + 15] = u[d] * Q[5]; /* TotalMoisture Advect */
}

static inline double sym(const double *c, int i, int j)
{
    static const int idx[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    return c[idx[i][j]];
}
/* nu (diagonal) and tau = (-2 nu) S, tau[d + 3 c] */
static void turbulence_tensors(const moist_t *m, const double *Q, const double *gf, const double *aux,
                               double *nu, double *tau)
{
    double S[6], k[3];
    for (int d = 0; d < 3; ++d) k[d] = aux[OPHI + 1 + d] / m->grav;
    const double *T = gf + 3; /* turbulence block */
    if (m->closure == 2) { /* AnisoMinDiss: gradient flux holds grad u (g[d + 3 c] = d u_c / d x_d) */
        S[0] = T[0];
        S[1] = (T[1] + T[3]) / 2;
        S[2] = (T[2] + T[6]) / 2;
        S[3] = T[4];
        S[4] = (T[5] + T[7]) / 2;
        S[5] = T[8];
    } else {
        for (int q = 0; q < 6; ++q) S[q] = T[q];
    }
    if (m->closure == 0) {
        const double v = m->kinematic ? m->visc : m->visc / Q[0];
        nu[0] = nu[1] = nu[2] = v;
    } else {
        const double N2 = T[m->ngt - 1];
        const double norm2 = S[0] * S[0] + 2 * (S[1] * S[1]) + 2 * (S[2] * S[2]) + S[3] * S[3] +
                             2 * (S[4] * S[4]) + S[5] * S[5];
        const double normS = sqrt(2 * norm2);
        const double epsn = nextafter(fabs(normS), INFINITY) - fabs(normS);
        const double Ri = N2 / (normS * normS + epsn);
        double c = 1.0 - Ri * m->invPr;
        c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
        const double fb2 = sqrt(c);
        double nu0[3];
        const double delta = aux[OTURB];
        if (m->closure == 1) {
            const double cd = m->visc * delta; /* C_smag * Delta */
            nu0[0] = nu0[1] = nu0[2] = normS * (cd * cd) + 1e-5;
        } else {
            /* delta_m = 1 (isotropic lengthscale): grad u_hat = grad u, S_hat = S.
               nu0 = (C delta)^2 max(1e-5, -dot(gu' gu, S_hat) / (dot(gu, gu) + eps(normS))) */
            double num = 0, den = 0;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    double gg = 0; /* (gu' * gu)[i][j] with gu[r][c] = T[r + 3 c] */
                    for (int r = 0; r < 3; ++r) gg += T[r + 3 * i] * T[r + 3 * j];
                    num += gg * sym(S, i, j);
                    den += T[i + 3 * j] * T[i + 3 * j];
                }
            double r = -num / (den + epsn);
            if (r < 1e-5) r = 1e-5;
            const double cd = m->visc * delta;
            nu0[0] = nu0[1] = nu0[2] = (cd * cd) * r;
        }
        const double dk = nu0[0] * k[0] + nu0[1] * k[1] + nu0[2] * k[2];
        for (int d = 0; d < 3; ++d) {
            const double nv = k[d] * dk, nh = nu0[d] - nv;
            nu[d] = nh + nv * fb2;
        }
    }
    for (int d = 0; d < 3; ++d)
        for (int c = 0; c < 3; ++c) tau[d + 3 * c] = (-2 * nu[d]) * sym(S, d, c);
}

static void mo_flux2(const void *p_, double *F, const double *Q, const double *gf, const double *hyp,
                     const double *aux, double t)
{
    const moist_t *m = (const moist_t *)p_;
    (void)t; (void)hyp;
    double nu[3], tau[9], dq[3];
    turbulence_tensors(m, Q, gf, aux, nu, tau);
    const double rho = Q[0];
    const double *gq = gf + 3 + m->ngt; /* grad q_tot */
    for (int d = 0; d < 3; ++d) dq[d] = (-(nu[d] * m->invPr)) * gq[d]; /* (-D_t) .* grad q_tot */
    for (int d = 0; d < 3; ++d) F[d] = dq[d] * rho; /* Mass: MoistureDiffusion */
    for (int c = 0; c < 3; ++c) /* Momentum: ViscousStress + MoistureDiffusion d_q .* rho u' */
        for (int d = 0; d < 3; ++d) F[d + 3 * (1 + c)] = (0.0 + tau[d + 3 * c] * rho) + dq[d] * Q[1 + c];
    for (int d = 0; d < 3; ++d) { /* Energy: ViscousFlux + DiffEnthalpyFlux */
        const double Dt = nu[d] * m->invPr;
        F[d + 12] = (tau[d] * Q[1] + tau[d + 3] * Q[2] + tau[d + 6] * Q[3]) + (-Dt * gf[d]) * rho;
    }
    for (int d = 0; d < 3; ++d) F[d + 15] = dq[d] * rho; /* TotalMoisture: MoistureDiffusion */
}

static inline double exner_of(const moist_t *m, const thermo_t *ts, double rho)
{
    return pow(air_pressure(ts, rho) / m->MSLP, ts->R_m / ts->cp_m);
}
/* sources in the order of the model's tuple: Gravity, BomexTendencies, BomexSponge,
   BomexGeostrophic (bomex_model.jl:396-420); each prognostic variable sums its terms in it */
static void mo_source(const void *p_, double *S, const double *Q, const double *gf, const double *aux,
                      double t, int dir)
{
    const moist_t *m = (const moist_t *)p_;
    (void)t; (void)dir;
    for (int q = 0; q < 6; ++q) S[q] = 0;
    const double rho = Q[0];
    double k[3];
    for (int d = 0; d < 3; ++d) k[d] = aux[OPHI + 1 + d] / m->grav;
    const double z = aux[OPHI] / m->grav; /* altitude */
    int first = 1;
    if (m->src & 1) { /* Gravity */
        const double r = m->subtract ? rho - aux[OREF] : rho;
        for (int d = 0; d < 3; ++d) S[1 + d] = -r * aux[OPHI + 1 + d];
        first = 0;
    }
    if (m->src & 2) { /* BomexTendencies: moisture, radiative cooling, subsidence (:141-246) */
        double rdqt, rdth, w_s = -0.0;
        const double lm = (z - m->zl_m) / (m->zh_m - m->zl_m);
        if (z <= m->zl_m) rdqt = rho * m->dqt_peak;
        else if (z <= m->zh_m) rdqt = rho * (m->dqt_peak - m->dqt_peak * lm);
        else rdqt = -0.0;
        const double lt = (z - m->zl_sub) / (m->z_max - m->zl_sub);
        if (z <= m->zl_sub) rdth = rho * m->dth_peak;
        else if (z <= m->z_max) rdth = rho * (m->dth_peak - m->dth_peak * lt);
        else rdth = -0.0;
        const double ls = (z - m->zl_sub) / (m->zh_sub - m->zl_sub);
        if (z <= m->zl_sub) w_s = -0.0 + z * (m->w_sub) / (m->zl_sub);
        else if (z <= m->zh_sub) w_s = m->w_sub - (m->w_sub) * ls;
        const double *gq = gf + 3 + m->ngt;
        const double kq = k[0] * gq[0] + k[1] * gq[1] + k[2] * gq[2];
        const double kh = k[0] * gf[0] + k[1] * gf[1] + k[2] * gf[2];
        S[0] = rdqt - rho * w_s * kq;
        S[5] = rdqt - rho * w_s * kq;
        /* precomputed.ts: the state the nodal refresh of this evaluation left in aux */
        thermo_t ts;
        ts.T = aux[OMOIST];
        ts.q_tot = Q[5] / Q[0];
        ts.q_liq = aux[OMOIST + 2];
        ts.q_ice = aux[OMOIST + 3];
        ts.R_m = gas_constant_air(m, ts.q_tot, ts.q_liq, ts.q_ice);
        ts.cv_m = cv_mix(m, ts.q_tot, ts.q_liq, ts.q_ice);
        ts.cp_m = cp_mix(m, ts.q_tot, ts.q_liq, ts.q_ice);
        const double term1 = ts.cv_m * rdth * exner_of(m, &ts, rho) + e_int_v0(m) * rdqt;
        const double term2 = rho * w_s * kh;
        S[4] = term1 - term2;
    }
    const double ug[3] = {m->u_geo + m->u_slope * z, m->v_geo, 0.0};
    if (m->src & 4) { /* BomexSponge (:106-133) */
        double v[3] = {0, 0, 0};
        if (m->z_sponge <= z) {
            const double r = (z - m->z_sponge) / (m->z_max - m->z_sponge);
            const double sp = sin(M_PI * (r / 2));
            const double beta = m->a_max * (m->gam == 2 ? sp * sp : pow(sp, m->gam));
            for (int d = 0; d < 3; ++d) v[d] = -beta * (Q[1 + d] - rho * ug[d]);
        }
        for (int d = 0; d < 3; ++d) S[1 + d] = first ? v[d] : S[1 + d] + v[d];
        first = 0;
    }
    if (m->src & 8) { /* BomexGeostrophic: -(f k) x (rho u - rho u_geo) (:79-104) */
        double a[3], b[3];
        for (int d = 0; d < 3; ++d) {
            a[d] = m->f_cor * k[d];
            b[d] = Q[1 + d] - rho * ug[d];
        }
        const double c[3] = {-(a[1] * b[2] - a[2] * b[1]), -(a[2] * b[0] - a[0] * b[2]),
                             -(a[0] * b[1] - a[1] * b[0])};
        for (int d = 0; d < 3; ++d) S[1 + d] = first ? c[d] : S[1 + d] + c[d];
        first = 0;
    }
}

static void mo_gradarg(const void *p_, double *G, const double *Q, const double *aux, double t)
{
    const moist_t *m = (const moist_t *)p_;
    (void)t;
    thermo_t ts;
    thermo_state(m, Q, aux, &ts);
    const double rhoinv = 1 / Q[0];
    for (int d = 0; d < 3; ++d) G[d] = rhoinv * Q[1 + d];
    const double e_tot = Q[4] * (1 / Q[0]);
    G[3] = e_tot + ts.R_m * ts.T;     /* total_specific_enthalpy */
    G[4] = aux[OMOIST + 1];           /* transform.turbulence.theta_v = aux.moisture.theta_v */
    G[5] = Q[5] * rhoinv;             /* q_tot */
}

static void mo_gradflux(const void *p_, double *gf, const double *g, const double *Q, const double *aux,
                        double t)
{
    const moist_t *m = (const moist_t *)p_;
    (void)Q; (void)t;
    for (int d = 0; d < 3; ++d) gf[d] = g[d + 3 * 3]; /* grad h_tot */
    double *T = gf + 3;
    if (m->closure == 2) {
        for (int q = 0; q < 9; ++q) T[q] = g[q]; /* grad u */
    } else {
        T[0] = g[0 + 3 * 0];
        T[1] = (g[1 + 3 * 0] + g[0 + 3 * 1]) / 2;
        T[2] = (g[2 + 3 * 0] + g[0 + 3 * 2]) / 2;
        T[3] = g[1 + 3 * 1];
        T[4] = (g[2 + 3 * 1] + g[1 + 3 * 2]) / 2;
        T[5] = g[2 + 3 * 2];
    }
    T[m->ngt - 1] = (g[0 + 3 * 4] * aux[OPHI + 1] + g[1 + 3 * 4] * aux[OPHI + 2] +
                     g[2 + 3 * 4] * aux[OPHI + 3]) / aux[OMOIST + 1];
    for (int d = 0; d < 3; ++d) gf[3 + m->ngt + d] = g[d + 3 * 5]; /* grad q_tot */
}

static void mo_postlap(const void *p_, double *hyp, const double *gl, const double *Q, const double *aux,
                       double t)
{
    (void)p_; (void)hyp; (void)gl; (void)Q; (void)aux; (void)t;
}

static void mo_wavespeed(const void *p_, double *ws, const double *n, const double *Q, const double *aux,
                         double t, int facedir)
{
    const moist_t *m = (const moist_t *)p_;
    (void)t; (void)facedir;
    thermo_t ts;
    thermo_state(m, Q, aux, &ts);
    const double rhoinv = 1 / Q[0];
    const double uN = fabs(n[0] * (rhoinv * Q[1]) + n[1] * (rhoinv * Q[2]) + n[2] * (rhoinv * Q[3]));
    const double ss = soundspeed(&ts);
    for (int s = 0; s < 6; ++s) ws[s] = uN + ss;
}

/* EquilMoist atmos_nodal_update_auxiliary_state! (moisture.jl:85-98) */
static void moist_update(const moist_t *m, const double *Q, double *aux)
{
    thermo_t ts;
    thermo_state(m, Q, aux, &ts);
    aux[OMOIST] = ts.T;
    aux[OMOIST + 1] = virtual_pottemp(m, &ts, Q[0]);
    aux[OMOIST + 2] = ts.q_liq;
    aux[OMOIST + 3] = ts.q_ice;
}
static void mo_bstate(const void *p_, int kind, int bctag, double *QP, double *auxP, const double *n,
                      const double *QM, const double *auxM, double t, const double *Q1,
                      const double *aux1)
{
    const moist_t *m = (const moist_t *)p_;
    (void)auxM; (void)t; (void)Q1; (void)aux1;
    const int bc = m->bc[bctag - 1];
    if (bc == 1 || bc == 2) { /* Impenetrable: FreeSlip, and DragLaw reflects the same way
                                 (bc_momentum.jl:21-40, 88-102) */
        const double dn = QM[1] * n[0] + QM[2] * n[1] + QM[3] * n[2];
        const double f = kind == ORC_BS_FIRST ? 2 * dn : dn;
        for (int d = 0; d < 3; ++d) QP[1 + d] -= f * n[d];
    }
    moist_update(m, QP, auxP);
}
/* normal_boundary_flux_second_order! of AtmosBC (boundaryconditions.jl:101-140): the default
   conditions add nothing; the BOMEX surface adds the drag-law stress (bc_momentum.jl:103-118),
   the prescribed energy flux (bc_energy.jl:87-99) and the prescribed moisture flux
   (bc_moisture.jl:38-52).  They are normal fluxes X; the caller forms F . n, so F = X n. */
static void mo_bflux2(const void *p_, int bctag, double *F, double *QP, double *gfP, double *hypP,
                      double *auxP, const double *n, const double *QM, const double *gfM,
                      const double *hypM, const double *auxM, double t, const double *Q1,
                      const double *gf1, const double *aux1)
{
    const moist_t *m = (const moist_t *)p_;
    (void)QP; (void)gfP; (void)hypP; (void)auxP; (void)gfM; (void)hypM; (void)auxM; (void)t; (void)gf1;
    (void)aux1;
    if (m->bc[bctag - 1] != 2) return;
    double X[6] = {0, 0, 0, 0, 0, 0};
    /* momentum: tau_n = C |u_tan| u_tan with C = (u_star / |u_tan|)^2, u of the first interior node */
    double u1[3], ut[3];
    for (int d = 0; d < 3; ++d) u1[d] = Q1[1 + d] / Q1[0];
    const double un = u1[0] * n[0] + u1[1] * n[1] + u1[2] * n[2];
    for (int d = 0; d < 3; ++d) ut[d] = u1[d] - un * n[d];
    const double nut = sqrt(ut[0] * ut[0] + ut[1] * ut[1] + ut[2] * ut[2]);
    const double Cd = (m->u_star / nut) * (m->u_star / nut);
    for (int d = 0; d < 3; ++d) X[1 + d] += QM[0] * (Cd * nut * ut[d]);
    X[4] -= m->e_flux; /* inward energy flux LHF + SHF */
    const double nrd = -m->q_flux;
    X[0] += nrd;
    for (int d = 0; d < 3; ++d) X[1 + d] += nrd / QM[0] * QM[1 + d];
    X[5] += nrd;
    for (int s = 0; s < 6; ++s)
        for (int d = 0; d < 3; ++d) F[d + 3 * s] += X[s] * n[d];
}
static void mo_bdiv(const void *p_, int bctag, double *gradP, double *auxP, const double *n,
                    const double *gradM, const double *auxM, double t)
{
    (void)p_; (void)bctag; (void)gradP; (void)auxP; (void)n; (void)gradM; (void)auxM; (void)t;
}
static void mo_bhigher(const void *p_, int bctag, double *QP, double *auxP, double *lapP, const double *n,
                       const double *QM, const double *auxM, const double *lapM, double t)
{
    (void)p_; (void)bctag; (void)QP; (void)auxP; (void)lapP; (void)n; (void)QM; (void)auxM; (void)lapM; (void)t;
}
static void mo_update_aux(const void *p_, const double *Q, double *aux, double t)
{
    (void)t;
    moist_update((const moist_t *)p_, Q, aux);
}

/* src/Atmos/Model/courant.jl:12-83 */
static double mo_courant(const void *p_, int kind, const double *Q, const double *aux, const double *gf,
                         double dx, double dt, double t, int direction)
{
    const moist_t *m = (const moist_t *)p_;
    (void)t;
    double k[3];
    for (int d = 0; d < 3; ++d) k[d] = aux[OPHI + 1 + d] / m->grav;
    if (kind == 2) {
        double nu[3], tau[9], normnu;
        turbulence_tensors(m, Q, gf, aux, nu, tau);
        if (m->closure == 0) {
            normnu = nu[0];
        } else {
            const double dk = nu[0] * k[0] + nu[1] * k[1] + nu[2] * k[2];
            if (direction == ORC_VERTICAL) {
                normnu = dk;
            } else {
                double v[3];
                for (int d = 0; d < 3; ++d) v[d] = direction == ORC_HORIZONTAL ? nu[d] - dk * k[d] : nu[d];
                normnu = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            }
        }
        return dt * normnu / (dx * dx);
    }
    const double dotk = Q[1] * k[0] + Q[2] * k[1] + Q[3] * k[2];
    double normu;
    if (direction == ORC_VERTICAL) {
        normu = fabs(dotk) / Q[0];
    } else {
        double v[3];
        for (int d = 0; d < 3; ++d)
            v[d] = direction == ORC_HORIZONTAL ? (Q[1 + d] - dotk * k[d]) / Q[0] : Q[1 + d] / Q[0];
        normu = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    }
    if (kind == 0) return dt * normu / dx;
    thermo_t ts;
    thermo_state(m, Q, aux, &ts);
    return dt * (normu + soundspeed(&ts)) / dx;
}

/* the saturation adjustment on its own, for the thermodynamic tests */
double orc_moist_saturation_adjustment(const orc_physics *ph, double e_int, double rho, double q_tot,
                                       double *q_liq, double *q_ice, double *e_int_back)
{
    const moist_t *m = (const moist_t *)ph->p;
    int unsat;
    const double T = saturation_adjustment(m, e_int, rho, q_tot, &unsat);
    phase_partition_equil(m, T, rho, q_tot, q_liq, q_ice);
    *e_int_back = internal_energy_T(m, T, q_tot, *q_liq, *q_ice);
    return T;
}

/* numerical_flux_first_order!(::RoeNumericalFluxMoist, ::AtmosModel, ...)
 * src/Atmos/Model/AtmosModel.jl:1276-1513; nf: 5 plain, 6 LM, 7 HH, 8 LV, 9 LVPP */
static double roe_avg(double sM, double sP, double vM, double vP) { return (sM * vM + sP * vP) / (sM + sP); }
static void solve6(double A[6][6], double *b)
{ /* LU with partial pivoting */
    for (int k = 0; k < 6; ++k) {
        int piv = k;
        double best = fabs(A[k][k]);
        for (int i = k + 1; i < 6; ++i)
            if (fabs(A[i][k]) > best) { best = fabs(A[i][k]); piv = i; }
        if (piv != k) {
            for (int j = 0; j < 6; ++j) { double t = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = t; }
            double t = b[k]; b[k] = b[piv]; b[piv] = t;
        }
        const double inv = 1 / A[k][k];
        for (int i = k + 1; i < 6; ++i) {
            const double l = A[i][k] * inv;
            A[i][k] = l;
            for (int j = k + 1; j < 6; ++j) A[i][j] -= l * A[k][j];
            b[i] -= l * b[k];
        }
    }
    for (int i = 5; i >= 0; --i) {
        double x = b[i];
        for (int j = i + 1; j < 6; ++j) x -= A[i][j] * b[j];
        b[i] = x / A[i][i];
    }
}
static void mo_nf_law(const void *p_, int nf, double *fluxn, const double *n, const double *QM,
                      const double *auxM, const double *QP, const double *auxP, double t, int facedir)
{
    const moist_t *m = (const moist_t *)p_;
    double FM[18], FP[18];
    for (int i = 0; i < 18; ++i) FM[i] = FP[i] = -0.0;
    mo_flux1(p_, FM, QM, auxM, t, facedir);
    mo_flux1(p_, FP, QP, auxP, t, facedir);
    const double nh[3] = {n[0] / 2, n[1] / 2, n[2] / 2};
    for (int s = 0; s < 6; ++s)
        fluxn[s] += (FM[3 * s] + FP[3 * s]) * nh[0] + (FM[3 * s + 1] + FP[3 * s + 1]) * nh[1] +
                    (FM[3 * s + 2] + FP[3 * s + 2]) * nh[2];
    const double gam = m->cp_d / m->cv_d, eiv0 = e_int_v0(m), Phi = auxM[OPHI];
    thermo_t tM, tP, ts;
    thermo_state(m, QM, auxM, &tM);
    thermo_state(m, QP, auxP, &tP);
    const double rM = QM[0], rP = QP[0];
    double uM[3], uP[3];
    for (int d = 0; d < 3; ++d) { uM[d] = QM[1 + d] / rM; uP[d] = QP[1 + d] / rP; }
    const double hM = QM[4] / rM + tM.R_m * tM.T, hP = QP[4] / rP + tP.R_m * tP.T;
    const double qM = QM[5] / rM, qP = QP[5] / rP;
    const double cM = soundspeed(&tM), cP = soundspeed(&tP);
    const double sM = sqrt(rM), sP = sqrt(rP);
    double ut[3];
    for (int d = 0; d < 3; ++d) ut[d] = roe_avg(sM, sP, uM[d], uP[d]);
    const double ht = roe_avg(sM, sP, hM, hP), qt = roe_avg(sM, sP, qM, qP);
    const double rho = sqrt(rM * rP);
    const double ei = roe_avg(sM, sP, tM.e_int, tP.e_int);
    int unsat;
    ts.q_tot = qt;
    ts.T = saturation_adjustment(m, ei, rho, qt, &unsat);
    if (unsat) ts.q_liq = ts.q_ice = 0.0;
    else phase_partition_equil(m, ts.T, rho, qt, &ts.q_liq, &ts.q_ice);
    ts.cv_m = cv_mix(m, ts.q_tot, ts.q_liq, ts.q_ice);
    const double utut = ut[0] * ut[0] + ut[1] * ut[1] + ut[2] * ut[2];
    const double ct = sqrt((gam - 1) * (ht - utut / 2));
    const double om = M_PI / 3, de = M_PI / 5;
    const double rv[3] = {sin(om) * cos(de), cos(om) * cos(de), sin(de)};
    const double t1[3] = {rv[1] * n[2] - rv[2] * n[1], rv[2] * n[0] - rv[0] * n[2], rv[0] * n[1] - rv[1] * n[0]};
    const double t2[3] = {t1[1] * n[2] - t1[2] * n[1], t1[2] * n[0] - t1[0] * n[2], t1[0] * n[1] - t1[1] * n[0]};
    const double unM = uM[0] * n[0] + uM[1] * n[1] + uM[2] * n[2];
    const double unP = uP[0] * n[0] + uP[1] * n[1] + uP[2] * n[2];
    const double un = ut[0] * n[0] + ut[1] * n[1] + ut[2] * n[2];
    double ucm[3], ucp[3];
    for (int d = 0; d < 3; ++d) { ucm[d] = ut[d] + ct * n[d]; ucp[d] = ut[d] - ct * n[d]; }
    double cLM = ct;
    if (nf == 6) {
        const double MaP = sqrt(uP[0] * uP[0] + uP[1] * uP[1] + uP[2] * uP[2]) / cP;
        const double MaM = sqrt(uM[0] * uM[0] + uM[1] * uM[1] + uM[2] * uM[2]) / cM;
        const double Ma = (MaP + MaM) / 2, w = 1 - Ma * Ma;
        cLM = ct * fmin(Ma * sqrt(4 + w * w) / (1 + Ma * Ma), 1.0);
    }
    double L[6] = {fabs(un - cLM), fabs(un), fabs(un), fabs(un), fabs(un + cLM), fabs(un)};
    if (nf == 7) {
        const double a = fmax(fabs(un), fmax(0.0, fmax(un - unM, unP - un)));
        L[0] = fmax(fabs(un - cLM), fmax(0.0, fmax(un - cLM - (unM - cM), unP - cP - (un - cLM))));
        L[1] = L[2] = L[3] = L[5] = a;
        L[4] = fmax(fabs(un + cLM), fmax(0.0, fmax(un + cLM - (unM + cM), unP + cP - (un + cLM))));
    }
    if (nf == 8 || nf == 9) {
        const int pp = nf == 9;
        double dL1, dL2, dL3, dR1, dR2, dR3;
        if (!pp) {
            dL1 = fmax(0.0, un - unM); dL2 = fmax(0.0, un - cLM - (unM - cM)); dL3 = fmax(0.0, un + cLM - (unM + cM));
            dR1 = fmax(0.0, unP - un); dR2 = fmax(0.0, unP - cP - (un - cLM)); dR3 = fmax(0.0, unP + cP - (un + cLM));
        } else {
            const double bL = fmin(un - cLM, unM - cM), bR = fmax(un + cLM, unP + cP);
            const double bm = fmin(0.0, bL), bp = fmax(0.0, bR);
            dL1 = fmax(0.0, un - bm); dL2 = fmax(0.0, un - cLM - bm); dL3 = fmax(0.0, un + cLM - bm);
            dR1 = fmax(0.0, bp - un); dR2 = fmax(0.0, bp - (un - cLM)); dR3 = fmax(0.0, bp - (un + cLM));
        }
        double qa1, qa2, qa3;
        if (un < dL1 && un > -dR1) qa1 = ((dL1 - dR1) * un + 2 * dL1 * dR1) / (dL1 + dR1);
        else qa1 = fabs(un);
        if ((pp ? un - cLM : un - ct) < dL2 && un - cLM > -dR2)
            qa2 = ((dL2 - dR2) * (pp ? un - ct : un - cLM) + 2 * dL2 * dR2) / (dL2 + dR2);
        else qa2 = fabs(un - cLM);
        if (un + cLM < dL3 && (pp ? un + cLM : un + ct) > -dR3)
            qa3 = ((dL3 - dR3) * (un + cLM) + 2 * dR3 * dR3) / (dL3 + dR3);
        else qa3 = fabs(un + cLM);
        L[0] = qa2; L[1] = L[2] = L[3] = L[5] = qa1; L[4] = qa3;
    }
    const double col[6][6] = {
        {1, ucp[0], ucp[1], ucp[2], ht - ct * un, qt},
        {0, t1[0], t1[1], t1[2], t1[0] * ut[0] + t1[1] * ut[1] + t1[2] * ut[2], 0},
        {0, t2[0], t2[1], t2[2], t2[0] * ut[0] + t2[1] * ut[1] + t2[2] * ut[2], 0},
        {1, ut[0], ut[1], ut[2], utut / 2 + Phi - m->T_0 * ts.cv_m, 0},
        {1, ucm[0], ucm[1], ucm[2], ht + ct * un, qt},
        {0, 0, 0, 0, eiv0, 1}};
    double A[6][6], x[6] = {rP - rM, QP[1] - QM[1], QP[2] - QM[2], QP[3] - QM[3], QP[4] - QM[4], QP[5] - QM[5]};
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) A[i][j] = col[j][i];
    solve6(A, x);
    for (int j = 0; j < 6; ++j) x[j] *= L[j];
    for (int i = 0; i < 6; ++i) {
        double acc = 0.0;
        for (int j = 0; j < 6; ++j) acc += col[j][i] * x[j];
        fluxn[i] -= acc / 2;
    }
}

orc_physics *orc_moist_new(const int *ip, const double *dp, int nf_first)
{
    orc_physics *ph = (orc_physics *)calloc(1, sizeof(orc_physics));
    moist_t *m = (moist_t *)calloc(1, sizeof(moist_t));
    m->closure = ip[0]; m->subtract = ip[1]; m->kinematic = ip[2]; m->maxiter = ip[3];
    m->src = ip[5]; m->nbc = ip[6];
    for (int i = 0; i < 7; ++i) m->bc[i] = ip[7 + i];
    m->visc = dp[0];
    m->R_d = dp[2]; m->cp_d = dp[3]; m->cv_d = dp[4]; m->T_0 = dp[5]; m->grav = dp[6];
    m->MSLP = dp[7]; m->invPr = dp[8]; m->tol = dp[9];
    m->R_v = dp[16]; m->cp_v = dp[17]; m->cp_l = dp[18]; m->cp_i = dp[19]; m->LH_v0 = dp[20];
    m->LH_s0 = dp[21]; m->T_triple = dp[22]; m->T_freeze = dp[23]; m->T_icenuc = dp[24];
    m->p_triple = dp[25]; m->T_min = dp[26];
    m->ngt = m->closure == 2 ? 10 : 7;
    m->u_star = dp[32]; m->e_flux = dp[33]; m->q_flux = dp[34]; m->f_cor = dp[35]; m->u_geo = dp[36];
    m->u_slope = dp[37]; m->v_geo = dp[38]; m->z_sponge = dp[39]; m->a_max = dp[40]; m->gam = dp[41];
    m->z_max = dp[42]; m->dqt_peak = dp[43]; m->zl_m = dp[44]; m->zh_m = dp[45]; m->dth_peak = dp[46];
    m->zl_sub = dp[47]; m->zh_sub = dp[48]; m->w_sub = dp[49];
    ph->ns = 6;
    ph->naux = NAUXM;
    ph->ngrad = 6;
    ph->ngf = 3 + m->ngt + 3;
    ph->ngl = 0;
    ph->nhyp = 0;
    ph->nf_first = nf_first;
    ph->p = m;
    ph->flux_first_order = mo_flux1;
    ph->flux_second_order = mo_flux2;
    ph->source = mo_source;
    ph->gradient_argument = mo_gradarg;
    ph->gradient_flux = mo_gradflux;
    ph->post_gradient_laplacian = mo_postlap;
    ph->wavespeed = mo_wavespeed;
    ph->boundary_state = mo_bstate;
    ph->boundary_flux_second_order = mo_bflux2;
    ph->boundary_state_divergence = mo_bdiv;
    ph->boundary_state_higher_order = mo_bhigher;
    ph->update_aux = mo_update_aux;
    ph->courant = mo_courant;
    ph->numerical_flux_law = mo_nf_law;
    return ph;
}
