// Device functor for the hydrostatic Boussinesq ocean model (uncoupled), restating
//   src/Ocean/HydrostaticBoussinesq/hydrostatic_boussinesq_model.jl:107-144 (state, aux),
//       :175-290 (gradient argument / flux), :419-520 (first-order fluxes), :539-552
//       (second-order flux), :571-606 (source), :613 (wavespeed), :621-635 (update_penalty!)
//   src/Ocean/HydrostaticBoussinesq/bc_velocity.jl, bc_temperature.jl (OceanBC)
//   src/Ocean/OceanProblems/simple_box_problem.jl:56-127 (Coriolis parameter)
// The law's update_auxiliary_state! (filters) and update_auxiliary_state_gradient! (column
// integrals) are operator hooks (cmdg_set_rhs_hooks), not part of this functor.
//
// Parameter block: iparam[0] momentum advection, [1] tracer advection, [2] Coriolis (0 fixed
// box f = -0, 1 rotating f = f_o, 2 beta plane), [3] coupling (1 = Coupled(): the baroclinic half
// of the split-explicit pair, src/Ocean/SplitExplicit/HydrostaticBoussinesqCoupling.jl), [6] nbc, [7..13] bc = velocity kind + 8 *
// temperature kind (velocity 1 Impenetrable(NoSlip), 2 Impenetrable(FreeSlip), 3
// Penetrable(FreeSlip), 4 Impenetrable(KinematicStress), 5 Penetrable(KinematicStress);
// temperature 0 Insulating, 1 TemperatureFlux -- stress and flux of the OceanGyre problem,
// ocean_gyre.jl:84-115); dparam[0..10] = grav c_h c_z alpha_T nu_h nu_z kappa_h kappa_z kappa_c
// f_o beta, [11..15] = tau_o rho_o L_y lambda_r theta_E.
#pragma once
#include "cmdg_common.h"

namespace cmdg {

struct OceanParams {
    int madv, tadv, cor, coupled, nbc;
    int bc[8];
    double grav, ch, cz, aT, nuh, nuz, kh, kz, kc, fo, beta, tau_o, rho_o, Ly, lam_r, thE;
};

struct HydroBoussinesq {
    using Params = OceanParams;
    enum { U = 0, V = 1, ETA = 2, TH = 3 };
    enum { AY = 0, AW = 1, APKIN = 2, AWZ0 = 3, AUD = 4 };
    enum { GDIVH = 0, GNU = 1, GKAPPA = 7 };
    static constexpr int NS = 4, NAUX = 8, NGRAD = 5, NGF = 10, NGL = 0, NHYP = 0;
    static constexpr bool HAS_UPDATE_AUX = false, FUSE_UPDATE_AUX = false, HAS_SOURCE = true;
    static constexpr bool HAS_COURANT = true, HAS_PENALTY = true;
    static constexpr int NUPD = 0, NDER = 0;
    __host__ __device__ static constexpr int upd_aux(int) { return 0; }
    __host__ __device__ static constexpr int hv_indexmap(int) { return 0; }
    __host__ __device__ static bool needs_gradflux(const Params &) { return true; }
    // the face fluxes read w and pkin (and y never): aux columns 1, 2
    static constexpr int NFAUX = 2;
    __host__ __device__ static constexpr int face_aux(int i) { return 1 + i; }
    static constexpr int GRAD_MIN_WAVES = 6;  // k_gradients: 98 VGPRs unconstrained

    static void make_params(Params &p, const int32_t *ip, const double *dp)
    {
        p.madv = ip[0];
        p.tadv = ip[1];
        p.cor = ip[2];
        p.coupled = ip[3];
        p.nbc = ip[6];
        for (int i = 0; i < 7; ++i) p.bc[i] = ip[7 + i];
        p.bc[7] = 0;
        p.grav = dp[0];
        p.ch = dp[1];
        p.cz = dp[2];
        p.aT = dp[3];
        p.nuh = dp[4];
        p.nuz = dp[5];
        p.kh = dp[6];
        p.kz = dp[7];
        p.kc = dp[8];
        p.fo = dp[9];
        p.beta = dp[10];
        p.tau_o = dp[11];
        p.rho_o = dp[12];
        p.Ly = dp[13];
        p.lam_r = dp[14];
        p.thE = dp[15];
    }

    __device__ static void flux_first_order(const Params &m, double *F, const double *Q,
                                            const double *aux, double, int)
    {
        const double ge = m.grav * Q[ETA], gp = m.grav * aux[APKIN];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const double I = d == c ? 1.0 : -0.0;  // I^h (3 x 2)
                if (!m.coupled) F[d + 3 * c] += ge * I;  // hydrostatic_pressure! (Uncoupled only)
                F[d + 3 * c] += gp * I;
            }
        const double v[3] = {Q[U], Q[V], aux[AW]};
        if (m.madv) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 3; ++d) F[d + 3 * c] += v[d] * Q[c];
        }
        if (m.tadv) {
#pragma unroll
            for (int d = 0; d < 3; ++d) F[d + 3 * TH] += v[d] * Q[TH];
        }
    }
    __device__ static void flux_second_order(const Params &, double *F, const double *,
                                             const double *gf, const double *, const double *,
                                             double)
    {
#pragma unroll
        for (int q = 0; q < 6; ++q) F[q] += gf[GNU + q];
#pragma unroll
        for (int d = 0; d < 3; ++d) F[d + 3 * TH] += gf[GKAPPA + d];
    }
    __device__ static double coriolis(const Params &m, double y)
    {
        return m.cor == 0 ? -0.0 : (m.cor == 1 ? m.fo : m.fo + m.beta * y);
    }
    __device__ static void source(const Params &m, double *S, const double *Q, const double *,
                                  const double *aux, const double *, double, int)
    {
        S[ETA] += aux[AWZ0];
        const double f = coriolis(m, aux[AY]);
        if (m.coupled) {  // coriolis_force!(::Coupled): deviation from the vertical mean
            S[U] -= -f * aux[AUD + 1];
            S[V] -= f * aux[AUD];
        } else {
            S[U] -= -f * Q[V];
            S[V] -= f * Q[U];
        }
        S[U] += 0;  // forcing: noforcing(args...) = 0
        S[V] += 0;
        S[ETA] += 0;
        S[TH] += 0;
    }
    __device__ static void init_derived(const Params &, double *, const double *) {}
    __device__ static void gradient_argument(const Params &m, double *G, const double *Q,
                                             const double *aux, double)
    {
        G[4] = Q[TH];
        G[0] = Q[U];
        G[1] = Q[V];
        if (m.coupled) {  // velocity_gradient_argument!(::Coupled)
            G[2] = aux[AUD];
            G[3] = aux[AUD + 1];
        }
    }
    __device__ static void gradient_flux(const Params &m, double *D, const double *g,
                                         const double *, const double *, double)
    {
        D[GDIVH] = g[0 + 3 * 0] + g[1 + 3 * 1];
        const double nu[3] = {m.nuh, m.nuh, m.nuz};
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                // Coupled: horizontal derivatives of u_d, vertical derivative of u
                const double gu = (m.coupled && d < 2) ? g[d + 3 * (2 + c)] : g[d + 3 * c];
                D[GNU + d + 3 * c] = -nu[d] * gu;
            }
        const double kap[3] = {m.kh, m.kh, g[2 + 3 * 4] < 0 ? m.kc : m.kz};
#pragma unroll
        for (int d = 0; d < 3; ++d) D[GKAPPA + d] = -kap[d] * g[d + 3 * 4];
    }
    __device__ static void post_gradient_laplacian(const Params &, double *, const double *,
                                                   const double *, const double *, double)
    {
    }
    __device__ static void wavespeed(const Params &m, double *ws, const double *n, const double *,
                                     const double *, double, int)
    {
        const double w = fabs(m.ch * n[0] + m.ch * n[1] + m.cz * n[2]);
#pragma unroll
        for (int s = 0; s < NS; ++s) ws[s] = w;
    }
    // update_penalty!(::RusanovNumericalFlux, ::HBModel, ...): no penalty on eta
    __device__ static void update_penalty(const Params &, double *pen, const double *,
                                          const double *, const double *)
    {
        pen[ETA] = -0.0;
    }
    __device__ static void boundary_state(const Params &m, int kind, int bctag, double *QP,
                                          double *auxP, const double *n, const double *QM,
                                          const double *auxM, double, const double *,
                                          const double *)
    {
        const int bv = m.bc[bctag - 1] & 7;
        if (bv == 1) {  // Impenetrable(NoSlip)
            if (kind == BS_FIRST) {
                QP[U] = -QM[U];
                QP[V] = -QM[V];
                auxP[AW] = -auxM[AW];
            } else {
                QP[U] = -0.0;
                QP[V] = -0.0;
                auxP[AW] = -0.0;
            }
        } else if (bv == 2 || bv == 4) {  // Impenetrable(FreeSlip); KinematicStress -> FreeSlip
            const double v[3] = {QM[U], QM[V], auxM[AW]};
            const double dn = kind == BS_FIRST
                                  ? (2 * n[0]) * v[0] + (2 * n[1]) * v[1] + (2 * n[2]) * v[2]
                                  : n[0] * v[0] + n[1] * v[1] + n[2] * v[2];
            QP[U] = v[0] - dn * n[0];
            QP[V] = v[1] - dn * n[1];
            auxP[AW] = v[2] - dn * n[2];
        }
        QP[TH] = QM[TH];  // Insulating
    }
    __device__ static void boundary_flux_second_order(
        const Params &m, int bctag, double *F, double *QP, double *gfP, double *hypP, double *auxP,
        const double *n, const double *QM, const double *gfM, const double *, const double *auxM,
        double t, const double *, const double *, const double *)
    {
        const int bv = m.bc[bctag - 1] & 7;
        if (bv == 1) {
            QP[U] = -QM[U];
            QP[V] = -QM[V];
            auxP[AW] = -auxM[AW];
#pragma unroll
            for (int q = 0; q < 6; ++q) gfP[GNU + q] = gfM[GNU + q];
        } else if (bv == 4 || bv == 5) {  // KinematicStress (bc_velocity.jl:218-289)
            const double st[2] = {(m.tau_o / m.rho_o) * cos(auxM[AY] * M_PI / m.Ly), -0.0};
            QP[U] = QM[U];
            QP[V] = QM[V];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 3; ++d) gfP[GNU + d + 3 * c] = n[d] * st[c];
        } else {
            QP[U] = QM[U];
            QP[V] = QM[V];
            auxP[AW] = auxM[AW];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 3; ++d) gfP[GNU + d + 3 * c] = n[d] * -0.0;
        }
        QP[TH] = QM[TH];
        if ((m.bc[bctag - 1] >> 3) == 1) {  // TemperatureFlux (bc_temperature.jl:66-88)
            const double thr = m.thE * (1 - auxM[AY] / m.Ly);
            const double fl = m.lam_r * (QM[TH] - thr);
#pragma unroll
            for (int d = 0; d < 3; ++d) gfP[GKAPPA + d] = n[d] * fl;
        } else {
#pragma unroll
            for (int d = 0; d < 3; ++d) gfP[GKAPPA + d] = n[d] * -0.0;
        }
        flux_second_order(m, F, QP, gfP, hypP, auxP, t);
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
    __device__ static void update_aux(const Params &, const double *, double *, double) {}
    __host__ __device__ static bool update_aux_active(const Params &) { return false; }
    // local Courant numbers of src/Ocean/HydrostaticBoussinesq/Courant.jl:13-111:
    // kind 0 advective (|w|, |u| or |(u, v, w)| by direction), 1 nondiffusive (gravity waves,
    // c_h), 2 diffusive (kappa; the factor 1000 on kappa_z stands for convective adjustment),
    // 3 viscous (nu)
    __device__ static double courant(const Params &m, int kind, const double *Q, const double *aux,
                                     const double *, double dx, double dt, double, int direction)
    {
        if (kind == 0) {
            double ub;
            if (direction == DIR_VERTICAL)
                ub = fabs(aux[AW]);
            else if (direction == DIR_HORIZONTAL)
                ub = sqrt(Q[U] * Q[U] + Q[V] * Q[V]);
            else
                ub = sqrt(Q[U] * Q[U] + Q[V] * Q[V] + aux[AW] * aux[AW]);
            return dt * ub / dx;
        }
        if (kind == 1) return dt * m.ch / dx;
        const double h = kind == 3 ? m.nuh : m.kh;
        const double z = kind == 3 ? m.nuz : 1000 * m.kz;
        const double nb = direction == DIR_VERTICAL     ? z
                          : direction == DIR_HORIZONTAL ? sqrt(2.0) * h
                                                        : sqrt(2 * (h * h) + z * z);
        return dt * nb / (dx * dx);
    }
};

}  // namespace cmdg
