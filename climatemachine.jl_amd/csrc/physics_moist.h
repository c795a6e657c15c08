// Device functor for the moist LES configuration of AtmosModel: TotalEnergyModel + EquilMoist,
// FlatOrientation, HydrostaticState (subtract_off), closure constant viscosity (0) /
// SmagorinskyLilly (1) / AnisoMinDiss (2), source Gravity, default AtmosBC.  Restates
//   src/Atmos/Model/AtmosModel.jl:397-520 (layouts), :625-690, :808-828,
//   tendencies_{mass,momentum,energy,moisture}.jl + atmos_tendencies.jl (term order),
//   moisture.jl:70-115 (EquilMoist), thermo_states.jl (PhaseEquil from (e_int, rho, q_tot); in
//   this snapshot every flux evaluation re-runs the saturation adjustment, :40-60),
//   src/Common/TurbulenceClosures/TurbulenceClosures.jl:411-497, :600-690.
// Thermodynamics.jl 0.3.2 is not in the reference tree: mixture gas constant / heat capacities,
// internal energy, saturation vapour pressure over liquid and ice, liquid fraction and the
// saturation adjustment (Newton on e_int_sat(T) - e_int) restate its published formulation;
// parity of saturated states is unpinned.  With q_tot = 0 every formula reduces operation by
// operation to physics_atmos.h.
//
// State rho, rho u[3], rho e, rho q_tot; auxiliary coord[3], Phi, grad Phi[3], ref_state[7],
// Delta, moisture (temperature, theta_v, q_liq, q_ice); gradient u[3], h_tot, theta_v, q_tot;
// gradient flux grad h_tot[3], S[6] | grad u[9], N^2, grad q_tot[3].
// Parameter block: see climatemachine.jl_amd/moist.py.
#pragma once
#include <math.h>

#include "cmdg_common.h"

namespace cmdg {

struct MoistParams {
    int subtract, kinematic, maxiter, src, nbc;
    int bc[7];
    double visc, R_d, cp_d, cv_d, T_0, grav, MSLP, invPr, tol;
    double R_v, cp_v, cp_l, cp_i, LH_v0, LH_s0, T_triple, T_freeze, T_icenuc, p_triple, T_min;
    // BOMEX (experiments/AtmosLES/bomex_model.jl:76-246, 352-470): surface fluxes and sources
    double u_star, e_flux, q_flux, f_cor, u_geo, u_slope, v_geo, z_sponge, a_max, gam, z_max;
    double dqt_peak, zl_m, zh_m, dth_peak, zl_sub, zh_sub, w_sub;
};

template <int CLOSURE, bool LAWNF = false>
struct MoistAtmos {
    using Params = MoistParams;
    // RoeNumericalFluxMoist lives in a variant of its own (a 6 x 6 characteristic solve per
    // face node) so that the kernels of every other configuration keep their register budget
    static constexpr bool LAW_NF = LAWNF;
    static constexpr int OPHI = 3, OREF = 7, OTURB = 14, OMOIST = 15;
    static constexpr int NGT = CLOSURE == 2 ? 10 : 7;  // turbulence block of the gradient flux
    static constexpr int NS = 6, NAUX = 19, NGRAD = 6, NGF = 3 + NGT + 3, NGL = 0, NHYP = 0;
    // (state_gradient_flux stays in the reference layout: node-major, as the dry law keeps it, takes 88 B/node
    // off the N = 6 tendency pass and not one microsecond -- profiles/r04_ab_node_major.txt)
    // the nodal refresh (one saturation adjustment per node) runs as its own pass before the
    // gradient kernel, as in the reference; the kernels then read temperature, theta_v and the
    // condensate from the auxiliary state instead of repeating the adjustment where the
    // reference reads them too (gradient argument / flux) or where the values are the same by
    // construction (source)
    static constexpr bool HAS_UPDATE_AUX = true, FUSE_UPDATE_AUX = false, HAS_SOURCE = true;
    static constexpr bool HAS_COURANT = true, HAS_PENALTY = false;
    static constexpr int NUPD = 4;
    __host__ __device__ static constexpr int upd_aux(int i) { return OMOIST + i; }
    static constexpr int NDER = 0;
    // auxiliary fields the interface kernels read on both sides: Phi, grad Phi, ref p, ref rho,
    // Delta and the equilibrium state of the nodal refresh (temperature, q_liq, q_ice)
    static constexpr int NFAUX = 10;
    __host__ __device__ static constexpr int face_aux(int i)
    {
        return i < 4 ? OPHI + i
                     : (i == 4 ? OREF
                               : (i == 5 ? OREF + 1
                                         : (i == 6 ? OTURB : (i == 7 ? OMOIST : OMOIST + i - 6))));
    }
    __host__ __device__ static constexpr int hv_indexmap(int) { return 0; }
    // Byte accounting (cmdg_query, bench.py): columns the volume code of a pass reads.  Gradients
    // (pass 0): the refreshed equilibrium state (temperature, theta_v, q_liq, q_ice) and grad Phi
    // (N^2); tendency (pass 3): the ten columns of face_aux (Phi, grad Phi, ref rho, ref p, Delta,
    // temperature, q_liq, q_ice) -- coord, the rest of ref_state and theta_v are never loaded.
    __host__ __device__ static constexpr int state_read(int) { return NS; }
    __host__ __device__ static constexpr int aux_read(int pass) { return pass == 0 ? 7 : (pass == 3 ? 10 : 0); }
    __host__ __device__ static bool needs_gradflux(const Params &) { return true; }
    __host__ __device__ static bool update_aux_active(const Params &) { return true; }

    static void make_params(Params &p, const int32_t *ip, const double *dp)
    {
        p.subtract = ip[1];
        p.kinematic = ip[2];
        p.maxiter = ip[3];
        p.src = ip[5];
        p.nbc = ip[6];
        for (int i = 0; i < 7; ++i) p.bc[i] = ip[7 + i];
        p.visc = dp[0];
        p.R_d = dp[2];
        p.cp_d = dp[3];
        p.cv_d = dp[4];
        p.T_0 = dp[5];
        p.grav = dp[6];
        p.MSLP = dp[7];
        p.invPr = dp[8];
        p.tol = dp[9];
        p.R_v = dp[16];
        p.cp_v = dp[17];
        p.cp_l = dp[18];
        p.cp_i = dp[19];
        p.LH_v0 = dp[20];
        p.LH_s0 = dp[21];
        p.T_triple = dp[22];
        p.T_freeze = dp[23];
        p.T_icenuc = dp[24];
        p.p_triple = dp[25];
        p.T_min = dp[26];
        p.u_star = dp[32];
        p.e_flux = dp[33];
        p.q_flux = dp[34];
        p.f_cor = dp[35];
        p.u_geo = dp[36];
        p.u_slope = dp[37];
        p.v_geo = dp[38];
        p.z_sponge = dp[39];
        p.a_max = dp[40];
        p.gam = dp[41];
        p.z_max = dp[42];
        p.dqt_peak = dp[43];
        p.zl_m = dp[44];
        p.zh_m = dp[45];
        p.dth_peak = dp[46];
        p.zl_sub = dp[47];
        p.zh_sub = dp[48];
        p.w_sub = dp[49];
    }

    // ---- Thermodynamics.jl: mixture properties ---------------------------------------
    struct Thermo {
        double T, q_tot, q_liq, q_ice, R_m, cv_m, cp_m;
    };
    __device__ static double cv_mix(const Params &m, double qt, double ql, double qi)
    {
        const double cv_v = m.cp_v - m.R_v;
        return m.cv_d + (cv_v - m.cv_d) * qt + (m.cp_l - cv_v) * ql + (m.cp_i - cv_v) * qi;
    }
    __device__ static double e_int_v0(const Params &m) { return m.LH_v0 - m.R_v * m.T_0; }
    __device__ static double e_int_i0(const Params &m) { return m.LH_s0 - m.LH_v0; }
    __device__ static double internal_energy_T(const Params &m, double T, double qt, double ql,
                                               double qi)
    {
        return cv_mix(m, qt, ql, qi) * (T - m.T_0) + (qt - ql) * e_int_v0(m) -
               qi * (e_int_v0(m) + e_int_i0(m));
    }
    __device__ static double liquid_fraction(const Params &m, double T)
    {
        if (T > m.T_freeze) return 1.0;
        if (T > m.T_icenuc) return (T - m.T_icenuc) / (m.T_freeze - m.T_icenuc);
        return 0.0;
    }
    __device__ static double q_vap_saturation(const Params &m, double T, double rho)
    {
        const double lam = liquid_fraction(m, T);
        const double LH_0 = lam * m.LH_v0 + (1 - lam) * m.LH_s0;
        const double dcp = lam * (m.cp_v - m.cp_l) + (1 - lam) * (m.cp_v - m.cp_i);
        const double pvs = m.p_triple * pow(T / m.T_triple, dcp / m.R_v) *
                           exp((LH_0 - dcp * m.T_0) / m.R_v * (1 / m.T_triple - 1 / T));
        return pvs / (rho * m.R_v * T);
    }
    __device__ static void phase_partition_equil(const Params &m, double T, double rho, double qt,
                                                 double &ql, double &qi)
    {
        const double qvs = q_vap_saturation(m, T, rho);
        const double qc = qt - qvs > 0 ? qt - qvs : 0.0;
        const double lam = liquid_fraction(m, T);
        ql = lam * qc;
        qi = (1 - lam) * qc;
    }
    // unsat is set when the all-vapour temperature is below saturation: no condensate then
    __device__ static double saturation_adjustment(const Params &m, double e_int, double rho,
                                                   double qt, bool &unsat)
    {
        double T = m.T_0 + (e_int - (qt - 0.0) * e_int_v0(m) + 0.0 * (e_int_v0(m) + e_int_i0(m))) /
                               cv_mix(m, qt, 0.0, 0.0);
        if (T < m.T_min) T = m.T_min;
        unsat = qt <= q_vap_saturation(m, T, rho) && T > m.T_min;
        if (unsat) return T;
        for (int it = 0; it < m.maxiter; ++it) {  // Newton on e_int_sat(T) - e_int
            double ql, qi;
            phase_partition_equil(m, T, rho, qt, ql, qi);
            const double f = internal_energy_T(m, T, qt, ql, qi) - e_int;
            const double lam = liquid_fraction(m, T);
            const double qvs = q_vap_saturation(m, T, rho);
            const double L = lam * m.LH_v0 + (1 - lam) * m.LH_s0;
            const double dlam = (T > m.T_icenuc && T < m.T_freeze) ? 1 / (m.T_freeze - m.T_icenuc) : 0.0;
            const double dqvs = qvs * L / (m.R_v * T * T);
            const double cv_v = m.cp_v - m.R_v;
            const double dcvm = cv_v - lam * m.cp_l - (1 - lam) * m.cp_i;
            const double fp = cv_mix(m, qt, ql, qi) +
                              (e_int_v0(m) + (1 - lam) * e_int_i0(m) + (T - m.T_0) * dcvm) * dqvs +
                              (ql + qi) * e_int_i0(m) * dlam;
            const double dT = f / fp;
            T -= dT;
            if (fabs(dT) < m.tol) break;
        }
        return T;
    }
    __device__ static double internal_energy(const double *Q, const double *aux)
    {
        const double rho = Q[0];
        const double rhoinv = 1 / rho;
        const double rhoe_kin = rhoinv * (Q[1] * Q[1] + Q[2] * Q[2] + Q[3] * Q[3]) / 2;
        const double rhoe_pot = rho * aux[OPHI];
        const double rhoe_int = Q[4] - rhoe_kin - rhoe_pot;
        return rhoinv * rhoe_int;
    }
    // new_thermo_state(atmos, ::TotalEnergyModel, ::EquilMoist, state, aux)
    __device__ static void thermo_state(const Params &m, const double *Q, const double *aux,
                                        Thermo &ts)
    {
        const double e_int = internal_energy(Q, aux);
        ts.q_tot = Q[5] / Q[0];
        bool unsat;
        ts.T = saturation_adjustment(m, e_int, Q[0], ts.q_tot, unsat);
        if (unsat) {  // q_c = max(q_tot - q_vap_sat, 0) = 0: the partition needs no second look
            ts.q_liq = 0.0;
            ts.q_ice = 0.0;
        } else {
            phase_partition_equil(m, ts.T, Q[0], ts.q_tot, ts.q_liq, ts.q_ice);
        }
        mixture(m, ts);
    }
    // The thermodynamic state of (Q, aux) as the nodal refresh of this evaluation left it in
    // aux.moisture (temperature, q_liq, q_ice).  The reference's recover_thermo_state repeats
    // the saturation adjustment in every flux, wave-speed and gradient-argument evaluation
    // (thermo_states.jl:40-60 in this snapshot); its result is a function of (Q, aux.Phi) alone,
    // so the value the refresh stored for the same node in the same evaluation is that result,
    // bit for bit -- on both sides of a face: the plus side is the neighbour's node, and a
    // boundary state passes through update_aux (boundary_state below) before it is used.
    __device__ static void thermo_state_refreshed(const Params &m, const double *Q,
                                                  const double *aux, Thermo &ts)
    {
        ts.T = aux[OMOIST];
        ts.q_tot = Q[5] / Q[0];
        ts.q_liq = aux[OMOIST + 2];
        ts.q_ice = aux[OMOIST + 3];
        mixture(m, ts);
    }
    __device__ static void mixture(const Params &m, Thermo &ts)
    {
        const double eps = m.R_v / m.R_d;
        ts.R_m = m.R_d * (1 + (eps - 1) * ts.q_tot - eps * (ts.q_liq + ts.q_ice));
        ts.cv_m = cv_mix(m, ts.q_tot, ts.q_liq, ts.q_ice);
        ts.cp_m = m.cp_d + (m.cp_v - m.cp_d) * ts.q_tot + (m.cp_l - m.cp_v) * ts.q_liq +
                  (m.cp_i - m.cp_v) * ts.q_ice;
    }
    __device__ static double air_pressure(const Thermo &ts, double rho) { return ts.R_m * rho * ts.T; }
    __device__ static double virtual_pottemp(const Params &m, const Thermo &ts, double rho)
    {
        const double exner = pow(air_pressure(ts, rho) / m.MSLP, ts.R_m / ts.cp_m);
        return ts.R_m / m.R_d * (ts.T / exner);
    }

    __device__ static void update_penalty(const Params &, double *, const double *, const double *,
                                          const double *)
    {
    }
    // local Courant numbers of src/Atmos/Model/courant.jl:12-83 (kind 0 advective, 1
    // nondiffusive with the moist sound speed, 2 diffusive with the closure's viscosity)
    __device__ static double courant(const Params &m, int kind, const double *Q, const double *aux,
                                     const double *gf, double dx, double dt, double, int direction)
    {
        double k[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) k[d] = aux[OPHI + 1 + d] / m.grav;
        if (kind == 2) {
            double nu[3], tau[9];
            turbulence_tensors(m, Q, gf, aux, nu, tau);
            double normnu;
            if constexpr (CLOSURE == 0) {
                normnu = nu[0];
            } else {
                const double dk = nu[0] * k[0] + nu[1] * k[1] + nu[2] * k[2];
                if (direction == DIR_VERTICAL) {
                    normnu = dk;
                } else {
                    double v[3];
#pragma unroll
                    for (int d = 0; d < 3; ++d)
                        v[d] = direction == DIR_HORIZONTAL ? nu[d] - dk * k[d] : nu[d];
                    normnu = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                }
            }
            return dt * normnu / (dx * dx);
        }
        const double dotk = Q[1] * k[0] + Q[2] * k[1] + Q[3] * k[2];
        double normu;
        if (direction == DIR_VERTICAL) {
            normu = fabs(dotk) / Q[0];
        } else {
            double v[3];
#pragma unroll
            for (int d = 0; d < 3; ++d)
                v[d] = direction == DIR_HORIZONTAL ? (Q[1 + d] - dotk * k[d]) / Q[0] : Q[1 + d] / Q[0];
            normu = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        }
        if (kind == 0) return dt * normu / dx;
        Thermo ts;
        thermo_state(m, Q, aux, ts);
        const double gamma = ts.cp_m / ts.cv_m;
        return dt * (normu + sqrt(gamma * ts.R_m * ts.T)) / dx;
    }

    // ---- fluxes ----------------------------------------------------------------------
    __device__ static void flux_first_order(const Params &m, double *F, const double *Q,
                                            const double *aux, double, int)
    {
        Thermo ts;
        thermo_state_refreshed(m, Q, aux, ts);
        flux_first_order_ts(m, F, Q, aux, ts);
    }
    // flux_first_order and wavespeed of one state from one thermodynamic state (the Rusanov flux
    // asks for both, kernels.h nf_first_order); same values as the two separate calls
    static constexpr bool HAS_FLUX_WAVESPEED = true;
    __device__ static void flux_wavespeed(const Params &m, double *F, double *ws, const double *n,
                                          const double *Q, const double *aux, double, int)
    {
        Thermo ts;
        thermo_state_refreshed(m, Q, aux, ts);
        flux_first_order_ts(m, F, Q, aux, ts);
        wavespeed_ts(ws, n, Q, ts);
    }
    __device__ static void flux_first_order_ts(const Params &m, double *F, const double *Q,
                                               const double *aux, const Thermo &ts)
    {
        const double rho = Q[0];
        const double p = air_pressure(ts, rho);
        double u[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) u[d] = Q[1 + d] / rho;
#pragma unroll
        for (int d = 0; d < 3; ++d) F[d] = Q[1 + d];
        const double pp = m.subtract ? p - aux[OREF + 1] : p;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d)
                F[d + 3 * (1 + c)] = Q[1 + d] * u[c] + (0.0 + (d == c ? pp : 0.0));
#pragma unroll
        for (int d = 0; d < 3; ++d) F[d + 12] = u[d] * Q[4] + u[d] * p;
#pragma unroll
        for (int d = 0; d < 3; ++d) F[d + 15] = u[d] * Q[5];
    }
    __device__ static double sym(const double *c, int i, int j)
    {
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        return c[lo == 0 ? hi : (lo == 1 ? 2 + hi : 5)];
    }
    __device__ static void turbulence_tensors(const Params &m, const double *Q, const double *gf,
                                              const double *aux, double *nu, double *tau)
    {
        double S[6], k[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) k[d] = aux[OPHI + 1 + d] / m.grav;
        const double *T = gf + 3;
        if constexpr (CLOSURE == 2) {  // gradient flux holds grad u: T[d + 3 c] = d u_c / d x_d
            S[0] = T[0];
            S[1] = (T[1] + T[3]) / 2;
            S[2] = (T[2] + T[6]) / 2;
            S[3] = T[4];
            S[4] = (T[5] + T[7]) / 2;
            S[5] = T[8];
        } else {
#pragma unroll
            for (int q = 0; q < 6; ++q) S[q] = T[q];
        }
        if constexpr (CLOSURE == 0) {
            const double v = m.kinematic ? m.visc : m.visc / Q[0];
            nu[0] = nu[1] = nu[2] = v;
        } else {
            const double N2 = T[NGT - 1];
            const double norm2 = S[0] * S[0] + 2 * (S[1] * S[1]) + 2 * (S[2] * S[2]) + S[3] * S[3] +
                                 2 * (S[4] * S[4]) + S[5] * S[5];
            const double normS = sqrt(2 * norm2);
            const double epsn = nextafter(fabs(normS), INFINITY) - fabs(normS);
            const double Ri = N2 / (normS * normS + epsn);
            double c = 1.0 - Ri * m.invPr;
            c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
            const double fb2 = sqrt(c);
            const double cd = m.visc * aux[OTURB];
            double nu0;
            if constexpr (CLOSURE == 1) {
                nu0 = normS * (cd * cd) + 1e-5;
            } else {  // AnisoMinDiss with an isotropic lengthscale: grad u_hat = grad u
                double num = 0, den = 0;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        double gg = 0;
#pragma unroll
                        for (int r = 0; r < 3; ++r) gg += T[r + 3 * i] * T[r + 3 * j];
                        num += gg * sym(S, i, j);
                        den += T[i + 3 * j] * T[i + 3 * j];
                    }
                double r = -num / (den + epsn);
                if (r < 1e-5) r = 1e-5;
                nu0 = (cd * cd) * r;
            }
            const double dk = nu0 * k[0] + nu0 * k[1] + nu0 * k[2];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const double nv = k[d] * dk, nh = nu0 - nv;
                nu[d] = nh + nv * fb2;
            }
        }
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int c = 0; c < 3; ++c) tau[d + 3 * c] = (-2 * nu[d]) * sym(S, d, c);
    }
    __device__ static void flux_second_order(const Params &m, double *F, const double *Q,
                                             const double *gf, const double *, const double *aux,
                                             double)
    {
        double nu[3], tau[9], dq[3];
        turbulence_tensors(m, Q, gf, aux, nu, tau);
        const double rho = Q[0];
        const double *gq = gf + 3 + NGT;
#pragma unroll
        for (int d = 0; d < 3; ++d) dq[d] = (-(nu[d] * m.invPr)) * gq[d];
#pragma unroll
        for (int d = 0; d < 3; ++d) F[d] = dq[d] * rho;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d)
                F[d + 3 * (1 + c)] = (0.0 + tau[d + 3 * c] * rho) + dq[d] * Q[1 + c];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const double Dt = nu[d] * m.invPr;
            F[d + 12] = (tau[d] * Q[1] + tau[d + 3] * Q[2] + tau[d + 6] * Q[3]) + (-Dt * gf[d]) * rho;
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) F[d + 15] = dq[d] * rho;
    }
    __device__ static void init_derived(const Params &, double *, const double *) {}
    // sources in the order of the model's tuple: Gravity, BomexTendencies, BomexSponge,
    // BomexGeostrophic (bomex_model.jl:396-420)
    __device__ static void source(const Params &m, double *S, const double *Q, const double *gf,
                                  const double *aux, const double *, double, int)
    {
#pragma unroll
        for (int q = 0; q < 6; ++q) S[q] = 0;
        const double rho = Q[0];
        double k[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) k[d] = aux[OPHI + 1 + d] / m.grav;
        const double z = aux[OPHI] / m.grav;
        bool first = true;
        if (m.src & 1) {  // Gravity
            const double r = m.subtract ? rho - aux[OREF] : rho;
#pragma unroll
            for (int d = 0; d < 3; ++d) S[1 + d] = -r * aux[OPHI + 1 + d];
            first = false;
        }
        if (m.src & 2) {  // BomexTendencies (:141-246)
            double rdqt, rdth, w_s = -0.0;
            const double lm = (z - m.zl_m) / (m.zh_m - m.zl_m);
            if (z <= m.zl_m)
                rdqt = rho * m.dqt_peak;
            else if (z <= m.zh_m)
                rdqt = rho * (m.dqt_peak - m.dqt_peak * lm);
            else
                rdqt = -0.0;
            const double lt = (z - m.zl_sub) / (m.z_max - m.zl_sub);
            if (z <= m.zl_sub)
                rdth = rho * m.dth_peak;
            else if (z <= m.z_max)
                rdth = rho * (m.dth_peak - m.dth_peak * lt);
            else
                rdth = -0.0;
            const double ls = (z - m.zl_sub) / (m.zh_sub - m.zl_sub);
            if (z <= m.zl_sub)
                w_s = -0.0 + z * (m.w_sub) / (m.zl_sub);
            else if (z <= m.zh_sub)
                w_s = m.w_sub - (m.w_sub) * ls;
            const double *gq = gf + 3 + NGT;
            const double kq = k[0] * gq[0] + k[1] * gq[1] + k[2] * gq[2];
            const double kh = k[0] * gf[0] + k[1] * gf[1] + k[2] * gf[2];
            S[0] = rdqt - rho * w_s * kq;
            S[5] = rdqt - rho * w_s * kq;
            // the thermodynamic state of this node as the nodal refresh left it
            Thermo ts;
            ts.T = aux[OMOIST];
            ts.q_tot = Q[5] / Q[0];
            ts.q_liq = aux[OMOIST + 2];
            ts.q_ice = aux[OMOIST + 3];
            mixture(m, ts);
            const double exner = pow(air_pressure(ts, rho) / m.MSLP, ts.R_m / ts.cp_m);
            const double term1 = ts.cv_m * rdth * exner + e_int_v0(m) * rdqt;
            const double term2 = rho * w_s * kh;
            S[4] = term1 - term2;
        }
        const double ug[3] = {m.u_geo + m.u_slope * z, m.v_geo, 0.0};
        if (m.src & 4) {  // BomexSponge (:106-133)
            double v[3] = {0, 0, 0};
            if (m.z_sponge <= z) {
                const double r = (z - m.z_sponge) / (m.z_max - m.z_sponge);
                const double sp = sin(3.14159265358979323846 * (r / 2));
                const double beta = m.a_max * (m.gam == 2 ? sp * sp : pow(sp, m.gam));
#pragma unroll
                for (int d = 0; d < 3; ++d) v[d] = -beta * (Q[1 + d] - rho * ug[d]);
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) S[1 + d] = first ? v[d] : S[1 + d] + v[d];
            first = false;
        }
        if (m.src & 8) {  // BomexGeostrophic: -(f k) x (rho u - rho u_geo) (:79-104)
            double a[3], b[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                a[d] = m.f_cor * k[d];
                b[d] = Q[1 + d] - rho * ug[d];
            }
            const double c[3] = {-(a[1] * b[2] - a[2] * b[1]), -(a[2] * b[0] - a[0] * b[2]),
                                 -(a[0] * b[1] - a[1] * b[0])};
#pragma unroll
            for (int d = 0; d < 3; ++d) S[1 + d] = first ? c[d] : S[1 + d] + c[d];
            first = false;
        }
    }
    __device__ static void gradient_argument(const Params &m, double *G, const double *Q,
                                             const double *aux, double)
    {
        Thermo ts;
        thermo_state_refreshed(m, Q, aux, ts);
        const double rhoinv = 1 / Q[0];
#pragma unroll
        for (int d = 0; d < 3; ++d) G[d] = rhoinv * Q[1 + d];
        const double e_tot = Q[4] * (1 / Q[0]);
        G[3] = e_tot + ts.R_m * ts.T;
        G[4] = aux[OMOIST + 1];  // transform.turbulence.theta_v = aux.moisture.theta_v
        G[5] = Q[5] * rhoinv;
    }
    __device__ static void gradient_flux(const Params &m, double *gf, const double *g,
                                         const double *Q, const double *aux, double)
    {
        const double th = aux[OMOIST + 1];
#pragma unroll
        for (int d = 0; d < 3; ++d) gf[d] = g[d + 9];
        double *T = gf + 3;
        if constexpr (CLOSURE == 2) {
#pragma unroll
            for (int q = 0; q < 9; ++q) T[q] = g[q];
        } else {
            T[0] = g[0 + 3 * 0];
            T[1] = (g[1 + 3 * 0] + g[0 + 3 * 1]) / 2;
            T[2] = (g[2 + 3 * 0] + g[0 + 3 * 2]) / 2;
            T[3] = g[1 + 3 * 1];
            T[4] = (g[2 + 3 * 1] + g[1 + 3 * 2]) / 2;
            T[5] = g[2 + 3 * 2];
        }
        T[NGT - 1] = (g[0 + 3 * 4] * aux[OPHI + 1] + g[1 + 3 * 4] * aux[OPHI + 2] +
                      g[2 + 3 * 4] * aux[OPHI + 3]) / th;
#pragma unroll
        for (int d = 0; d < 3; ++d) gf[3 + NGT + d] = g[d + 15];
    }
    __device__ static void post_gradient_laplacian(const Params &, double *, const double *,
                                                   const double *, const double *, double)
    {
    }
    __device__ static void wavespeed(const Params &m, double *ws, const double *n, const double *Q,
                                     const double *aux, double, int)
    {
        Thermo ts;
        thermo_state_refreshed(m, Q, aux, ts);
        wavespeed_ts(ws, n, Q, ts);
    }
    __device__ static void wavespeed_ts(double *ws, const double *n, const double *Q,
                                        const Thermo &ts)
    {
        const double rhoinv = 1 / Q[0];
        const double uN =
            fabs(n[0] * (rhoinv * Q[1]) + n[1] * (rhoinv * Q[2]) + n[2] * (rhoinv * Q[3]));
        const double gamma = ts.cp_m / ts.cv_m;
        const double ss = sqrt(gamma * ts.R_m * ts.T);
#pragma unroll
        for (int s = 0; s < 6; ++s) ws[s] = uN + ss;
    }
    // numerical_flux_first_order!(::RoeNumericalFluxMoist, ::AtmosModel, ...)
    // (src/Atmos/Model/AtmosModel.jl:1276-1513): central flux, then the Roe matrix dissipation
    // M |Lambda| M^-1 (Q+ - Q-) / 2 of the six-variable moist system about the Roe-averaged state
    // (PhaseEquil from the averaged e_int, rho, q_tot), with the four optional entropy /
    // low-Mach fixes of the reference, quirks included.
    __device__ static double roe_average(double sM, double sP, double vM, double vP)
    {
        return (sM * vM + sP * vP) / (sM + sP);
    }
    __device__ static void solve6(double A[6][6], double *b)
    {  // LU with partial pivoting (what `M \ b` of a 6 x 6 static matrix does)
        for (int k = 0; k < 6; ++k) {
            int piv = k;
            double best = fabs(A[k][k]);
            for (int i = k + 1; i < 6; ++i)
                if (fabs(A[i][k]) > best) best = fabs(A[i][k]), piv = i;
            if (piv != k) {
                for (int j = 0; j < 6; ++j) {
                    const double t = A[k][j];
                    A[k][j] = A[piv][j];
                    A[piv][j] = t;
                }
                const double t = b[k];
                b[k] = b[piv];
                b[piv] = t;
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
    __device__ static void numerical_flux_law(const Params &m, int nf, double *fluxn,
                                              const double *n, const double *QM,
                                              const double *auxM, const double *QP,
                                              const double *auxP, double t, int facedir)
    {
        double FM[18], FP[18];
#pragma unroll
        for (int i = 0; i < 18; ++i) FM[i] = FP[i] = -0.0;
        flux_first_order(m, FM, QM, auxM, t, facedir);
        flux_first_order(m, FP, QP, auxP, t, facedir);
        const double nh0 = n[0] / 2, nh1 = n[1] / 2, nh2 = n[2] / 2;
#pragma unroll
        for (int s = 0; s < 6; ++s)
            fluxn[s] += (FM[3 * s] + FP[3 * s]) * nh0 + (FM[3 * s + 1] + FP[3 * s + 1]) * nh1 +
                        (FM[3 * s + 2] + FP[3 * s + 2]) * nh2;
        const double gam = m.cp_d / m.cv_d, eiv0 = e_int_v0(m);
        const double Phi = auxM[OPHI];
        Thermo tM, tP;
        thermo_state_refreshed(m, QM, auxM, tM);
        thermo_state_refreshed(m, QP, auxP, tP);
        const double rM = QM[0], rP = QP[0];
        double uM[3], uP[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) uM[d] = QM[1 + d] / rM, uP[d] = QP[1 + d] / rP;
        const double eM = QM[4] / rM, eP = QP[4] / rP;
        const double hM = eM + tM.R_m * tM.T, hP = eP + tP.R_m * tP.T;
        const double qM = QM[5] / rM, qP = QP[5] / rP;
        const double cM = sqrt(tM.cp_m / tM.cv_m * tM.R_m * tM.T);
        const double cP = sqrt(tP.cp_m / tP.cv_m * tP.R_m * tP.T);
        const double sM = sqrt(rM), sP = sqrt(rP);
        double ut[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) ut[d] = roe_average(sM, sP, uM[d], uP[d]);
        const double ht = roe_average(sM, sP, hM, hP);
        const double qt = roe_average(sM, sP, qM, qP);
        const double rho = sqrt(rM * rP);
        const double ei = roe_average(sM, sP, internal_energy(QM, auxM), internal_energy(QP, auxP));
        Thermo ts;  // PhaseEquil(param_set, e_int, rho, q_tot, maxiter, tolerance)
        ts.q_tot = qt;
        bool unsat;
        ts.T = saturation_adjustment(m, ei, rho, qt, unsat);
        if (unsat)
            ts.q_liq = ts.q_ice = 0.0;
        else
            phase_partition_equil(m, ts.T, rho, qt, ts.q_liq, ts.q_ice);
        mixture(m, ts);
        const double ct = sqrt((gam - 1) * (ht - (ut[0] * ut[0] + ut[1] * ut[1] + ut[2] * ut[2]) / 2));
        const double om = M_PI / 3, de = M_PI / 5;
        const double rv[3] = {sin(om) * cos(de), cos(om) * cos(de), sin(de)};
        const double t1[3] = {rv[1] * n[2] - rv[2] * n[1], rv[2] * n[0] - rv[0] * n[2],
                              rv[0] * n[1] - rv[1] * n[0]};
        const double t2[3] = {t1[1] * n[2] - t1[2] * n[1], t1[2] * n[0] - t1[0] * n[2],
                              t1[0] * n[1] - t1[1] * n[0]};
        const double unM = uM[0] * n[0] + uM[1] * n[1] + uM[2] * n[2];
        const double unP = uP[0] * n[0] + uP[1] * n[1] + uP[2] * n[2];
        const double un = ut[0] * n[0] + ut[1] * n[1] + ut[2] * n[2];
        double ucm[3], ucp[3];  // u + c n ("minus" in the reference's naming), u - c n
#pragma unroll
        for (int d = 0; d < 3; ++d) ucm[d] = ut[d] + ct * n[d], ucp[d] = ut[d] - ct * n[d];
        double cLM = ct;
        if (nf == NF_ROE_MOIST_LM) {
            const double MaP = sqrt(uP[0] * uP[0] + uP[1] * uP[1] + uP[2] * uP[2]) / cP;
            const double MaM = sqrt(uM[0] * uM[0] + uM[1] * uM[1] + uM[2] * uM[2]) / cM;
            const double Ma = (MaP + MaM) / 2;
            const double w = 1 - Ma * Ma;
            cLM = ct * fmin(Ma * sqrt(4 + w * w) / (1 + Ma * Ma), 1.0);
        }
        double L[6] = {fabs(un - cLM), fabs(un), fabs(un), fabs(un), fabs(un + cLM), fabs(un)};
        if (nf == NF_ROE_MOIST_HH) {
            const double a = fmax(fabs(un), fmax(0.0, fmax(un - unM, unP - un)));
            L[0] = fmax(fabs(un - cLM), fmax(0.0, fmax(un - cLM - (unM - cM), unP - cP - (un - cLM))));
            L[1] = L[2] = L[3] = L[5] = a;
            L[4] = fmax(fabs(un + cLM), fmax(0.0, fmax(un + cLM - (unM + cM), unP + cP - (un + cLM))));
        }
        if (nf == NF_ROE_MOIST_LV || nf == NF_ROE_MOIST_LVPP) {
            double dL1, dL2, dL3, dR1, dR2, dR3;
            const bool pp = nf == NF_ROE_MOIST_LVPP;
            if (!pp) {
                dL1 = fmax(0.0, un - unM);
                dL2 = fmax(0.0, un - cLM - (unM - cM));
                dL3 = fmax(0.0, un + cLM - (unM + cM));
                dR1 = fmax(0.0, unP - un);
                dR2 = fmax(0.0, unP - cP - (un - cLM));
                dR3 = fmax(0.0, unP + cP - (un + cLM));
            } else {
                const double bL = fmin(un - cLM, unM - cM), bR = fmax(un + cLM, unP + cP);
                const double bm = fmin(0.0, bL), bp = fmax(0.0, bR);
                dL1 = fmax(0.0, un - bm);
                dL2 = fmax(0.0, un - cLM - bm);
                dL3 = fmax(0.0, un + cLM - bm);
                dR1 = fmax(0.0, bp - un);
                dR2 = fmax(0.0, bp - (un - cLM));
                dR3 = fmax(0.0, bp - (un + cLM));
            }
            double qa1, qa2, qa3;
            if (un < dL1 && un > -dR1)
                qa1 = ((dL1 - dR1) * un + 2 * dL1 * dR1) / (dL1 + dR1);
            else
                qa1 = fabs(un);
            // (the mixed use of c and c_LM, and delta_R_3 squared, are the reference's)
            if ((pp ? un - cLM : un - ct) < dL2 && un - cLM > -dR2)
                qa2 = ((dL2 - dR2) * (pp ? un - ct : un - cLM) + 2 * dL2 * dR2) / (dL2 + dR2);
            else
                qa2 = fabs(un - cLM);
            if (un + cLM < dL3 && (pp ? un + cLM : un + ct) > -dR3)
                qa3 = ((dL3 - dR3) * (un + cLM) + 2 * dR3 * dR3) / (dL3 + dR3);
            else
                qa3 = fabs(un + cLM);
            L[0] = qa2;
            L[1] = L[2] = L[3] = L[5] = qa1;
            L[4] = qa3;
        }
        const double utut = ut[0] * ut[0] + ut[1] * ut[1] + ut[2] * ut[2];
        // columns of M (AtmosModel.jl:1497-1504)
        const double col[6][6] = {
            {1, ucp[0], ucp[1], ucp[2], ht - ct * un, qt},
            {0, t1[0], t1[1], t1[2], t1[0] * ut[0] + t1[1] * ut[1] + t1[2] * ut[2], 0},
            {0, t2[0], t2[1], t2[2], t2[0] * ut[0] + t2[1] * ut[1] + t2[2] * ut[2], 0},
            {1, ut[0], ut[1], ut[2], utut / 2 + Phi - m.T_0 * ts.cv_m, 0},
            {1, ucm[0], ucm[1], ucm[2], ht + ct * un, qt},
            {0, 0, 0, 0, eiv0, 1}};
        double A[6][6], x[6] = {rP - rM, QP[1] - QM[1], QP[2] - QM[2], QP[3] - QM[3], QP[4] - QM[4],
                                QP[5] - QM[5]};
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) A[i][j] = col[j][i];
        solve6(A, x);
#pragma unroll
        for (int j = 0; j < 6; ++j) x[j] *= L[j];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 6; ++j) acc += col[j][i] * x[j];
            fluxn[i] -= acc / 2;
        }
    }
    // EquilMoist atmos_nodal_update_auxiliary_state! (moisture.jl:85-98)
    __device__ static void update_aux(const Params &m, const double *Q, double *aux, double)
    {
        Thermo ts;
        thermo_state(m, Q, aux, ts);
        aux[OMOIST] = ts.T;
        aux[OMOIST + 1] = virtual_pottemp(m, ts, Q[0]);
        aux[OMOIST + 2] = ts.q_liq;
        aux[OMOIST + 3] = ts.q_ice;
    }
    __device__ static void boundary_state(const Params &m, int kind, int bctag, double *QP,
                                          double *auxP, const double *n, const double *QM,
                                          const double *, double t, const double *, const double *)
    {
        const int bc = m.bc[bctag - 1];
        if (bc == 1 || bc == 2) {  // Impenetrable: FreeSlip; DragLaw reflects the same way
            const double dn = QM[1] * n[0] + QM[2] * n[1] + QM[3] * n[2];
            const double f = kind == BS_FIRST ? 2 * dn : dn;
#pragma unroll
            for (int d = 0; d < 3; ++d) QP[1 + d] -= f * n[d];
        }
        update_aux(m, QP, auxP, t);
    }
    // normal_boundary_flux_second_order! of AtmosBC: the default conditions add nothing; the
    // BOMEX surface adds the drag-law stress (bc_momentum.jl:103-118), the prescribed energy flux
    // (bc_energy.jl:87-99) and the prescribed moisture flux (bc_moisture.jl:38-52).  They are
    // normal fluxes X; the kernel forms F . n, so F = X n.
    __device__ static void boundary_flux_second_order(const Params &m, int bctag, double *F,
                                                      double *, double *, double *, double *,
                                                      const double *n, const double *QM,
                                                      const double *, const double *,
                                                      const double *, double, const double *Q1,
                                                      const double *, const double *)
    {
        if (m.bc[bctag - 1] != 2) return;
        double X[6] = {0, 0, 0, 0, 0, 0};
        double u1[3], ut[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) u1[d] = Q1[1 + d] / Q1[0];
        const double un = u1[0] * n[0] + u1[1] * n[1] + u1[2] * n[2];
#pragma unroll
        for (int d = 0; d < 3; ++d) ut[d] = u1[d] - un * n[d];
        const double nut = sqrt(ut[0] * ut[0] + ut[1] * ut[1] + ut[2] * ut[2]);
        const double Cd = (m.u_star / nut) * (m.u_star / nut);
#pragma unroll
        for (int d = 0; d < 3; ++d) X[1 + d] += QM[0] * (Cd * nut * ut[d]);
        X[4] -= m.e_flux;
        const double nrd = -m.q_flux;
        X[0] += nrd;
#pragma unroll
        for (int d = 0; d < 3; ++d) X[1 + d] += nrd / QM[0] * QM[1 + d];
        X[5] += nrd;
#pragma unroll
        for (int s = 0; s < 6; ++s)
#pragma unroll
            for (int d = 0; d < 3; ++d) F[d + 3 * s] += X[s] * n[d];
    }
    __device__ static void boundary_state_divergence(const Params &, int, double *, double *,
                                                     const double *, const double *,
                                                     const double *, double)
    {
    }
    __device__ static void boundary_state_higher_order(const Params &, int, double *, double *,
                                                       double *, const double *, const double *,
                                                       const double *, const double *, double)
    {
    }
};

}  // namespace cmdg
