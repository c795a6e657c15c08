// Device functors for the older split-explicit ocean of the reference, src/Ocean/SplitExplicit01
// (the variant experiments/OceanSplitExplicit/simple_box.jl runs, with fast-step averaging):
//   OceanSE01        OceanModel            OceanModel.jl:1-603
//   Continuity3dSE01 Continuity3dModel     Continuity3dModel.jl:1-76   (grad_h . u by a DG operator)
//   BarotropicSE01   BarotropicModel       BarotropicModel.jl:1-196    (2-D in the reference; here on
//                                          the one-layer extrusion of the 2-D grid, see physics_sw.h)
// with the boundary conditions of OceanBoundaryConditions.jl:1-642.  Auxiliary and gradient-flux
// layouts are the reference's (vars_state order).  The law's update_auxiliary_state! -- filters,
// the continuity operator, column integrals, w(z = 0), flow deviation (OceanModel.jl:432-541) -- is
// a recorded composition replayed by the operator (cmdg_set_rhs_hooks), not part of the functor.
//
// Parameter block (all three): iparam[0] numImplSteps > 0 (implicit vertical diffusion on:
// kappa_z / 2 instead of the convective-adjustment switch), [6] nbc, [7..13] boundary condition of
// tag 1..7: 1 CoastlineFreeSlip, 2 CoastlineNoSlip, 3 OceanFloorFreeSlip, 4 OceanFloorNoSlip,
// 6 OceanSurfaceNoStressNoForcing, 7 OceanSurfaceStressNoForcing, 8 OceanSurfaceNoStressForcing,
// 9 OceanSurfaceStressForcing; dparam[0..10] = grav c_h c_z alpha_T nu_h nu_z kappa_h kappa_z
// kappa_c f_o beta, [11..16] = tau_o rho_o L_y lambda_r theta_E H.
#pragma once
#include "cmdg_common.h"

namespace cmdg {

struct OceanSE01Params {
    int impl, nbc;
    int bc[8];
    double grav, ch, cz, aT, nuh, nuz, kh, kz, kc, fo, beta, tau_o, rho_o, Ly, lam_r, thE, H;
};

enum {
    SE01_COAST_FREESLIP = 1, SE01_COAST_NOSLIP = 2, SE01_FLOOR_FREESLIP = 3, SE01_FLOOR_NOSLIP = 4,
    SE01_SURF_NONE = 6, SE01_SURF_STRESS = 7, SE01_SURF_FORCING = 8, SE01_SURF_STRESS_FORCING = 9
};

static inline void se01_make_params(OceanSE01Params &p, const int32_t *ip, const double *dp)
{
    p.impl = ip[0];
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
    p.H = dp[16];
}

// the pieces every law of this file leaves at their defaults
struct SE01Defaults {
    static constexpr int NGL = 0, NHYP = 0, NUPD = 0, NDER = 0;
    static constexpr bool HAS_UPDATE_AUX = false, FUSE_UPDATE_AUX = false, HAS_COURANT = false;
    __host__ __device__ static constexpr int upd_aux(int) { return 0; }
    __host__ __device__ static constexpr int hv_indexmap(int) { return 0; }
    __host__ __device__ static bool update_aux_active(const OceanSE01Params &) { return false; }
    __device__ static void init_derived(const OceanSE01Params &, double *, const double *) {}
    __device__ static void post_gradient_laplacian(const OceanSE01Params &, double *, const double *,
                                                   const double *, const double *, double)
    {
    }
    __device__ static void boundary_state_divergence(const OceanSE01Params &, int, double *, double *,
                                                     const double *, const double *, const double *,
                                                     double)
    {
    }
    __device__ static void boundary_state_higher_order(const OceanSE01Params &, int, double *,
                                                       double *, double *, const double *,
                                                       const double *, const double *,
                                                       const double *, double)
    {
    }
    __device__ static void update_aux(const OceanSE01Params &, const double *, double *, double) {}
    __device__ static double courant(const OceanSE01Params &, int, const double *, const double *,
                                     const double *, double, double, double, int)
    {
        return 0.0;
    }
    // abs(SVector(c_h, c_h, c_z)' * n)
    __device__ static double gravity_wavespeed(const OceanSE01Params &m, const double *n)
    {
        return fabs(m.ch * n[0] + m.ch * n[1] + m.cz * n[2]);
    }
};

// ---------------------------------------------------------------------------------------
struct OceanSE01 : SE01Defaults {
    using Params = OceanSE01Params;
    enum { U = 0, V = 1, ETA = 2, TH = 3 };
    enum { AW = 0, APKIN = 1, AWZ0 = 2, AUD = 3, ADGU = 5, AY = 7 };  // OceanModel.jl:186-195
    enum { GNU = 0, GKAPPA = 6 };                                      // :237-242
    static constexpr int NS = 4, NAUX = 8, NGRAD = 5, NGF = 9;
    static constexpr bool HAS_SOURCE = true, HAS_PENALTY = true;
    __host__ __device__ static bool needs_gradflux(const Params &) { return true; }
    static constexpr int NFAUX = 2;  // the face fluxes read w and pkin
    __host__ __device__ static constexpr int face_aux(int i) { return AW + i; }
    static void make_params(Params &p, const int32_t *ip, const double *dp) { se01_make_params(p, ip, dp); }

    // flux_first_order!  OceanModel.jl:368-402: temperature advection by (u, v, w), kinematic
    // pressure; the momentum advection and the surface-height gradient are switched off there
    __device__ static void flux_first_order(const Params &, double *F, const double *Q,
                                            const double *aux, double, int)
    {
        const double v[3] = {Q[U], Q[V], aux[AW]};
#pragma unroll
        for (int d = 0; d < 3; ++d) F[d + 3 * TH] += v[d] * Q[TH];
        const double pk = aux[APKIN];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) F[d + 3 * c] += pk * (d == c ? 1.0 : -0.0);
    }
    __device__ static void flux_second_order(const Params &, double *F, const double *,
                                             const double *gf, const double *, const double *,
                                             double)
    {  // :404-421
#pragma unroll
        for (int q = 0; q < 6; ++q) F[q] += gf[GNU + q];
#pragma unroll
        for (int d = 0; d < 3; ++d) F[d + 3 * TH] += gf[GKAPPA + d];
    }
    __device__ static void source(const Params &m, double *S, const double *, const double *,
                                  const double *aux, const double *, double, int)
    {  // :423-449: Coriolis force on the deviation from the vertical mean, barotropic
       // tendency adjustment, surface w into eta
        const double f = m.fo + m.beta * aux[AY];
        S[U] -= -f * aux[AUD + 1];
        S[V] -= f * aux[AUD];
        S[U] += aux[ADGU];
        S[V] += aux[ADGU + 1];
        S[ETA] += aux[AWZ0];
    }
    __device__ static void gradient_argument(const Params &, double *G, const double *Q,
                                             const double *aux, double)
    {  // :221-235
        G[0] = Q[U];
        G[1] = Q[V];
        G[2] = aux[AUD];
        G[3] = aux[AUD + 1];
        G[4] = Q[TH];
    }
    __device__ static void gradient_flux(const Params &m, double *D, const double *g,
                                         const double *, const double *, double)
    {  // :244-279: horizontal derivatives of u_d, vertical derivative of u
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            D[GNU + 0 + 3 * c] = -(m.nuh * g[0 + 3 * (2 + c)]);
            D[GNU + 1 + 3 * c] = -(m.nuh * g[1 + 3 * (2 + c)]);
            D[GNU + 2 + 3 * c] = -(m.nuz * g[2 + 3 * c]);
        }
        const double dthdz = g[2 + 3 * 4];
        const double kv = m.impl ? m.kz * 0.5 : (dthdz < 0 ? m.kc : m.kz);
        const double kap[3] = {m.kh, m.kh, kv};
#pragma unroll
        for (int d = 0; d < 3; ++d) D[GKAPPA + d] = -kap[d] * g[d + 3 * 4];
    }
    __device__ static void wavespeed(const Params &m, double *ws, const double *n, const double *,
                                     const double *, double, int)
    {
        const double w = gravity_wavespeed(m, n);
#pragma unroll
        for (int s = 0; s < NS; ++s) ws[s] = w;
    }
    __device__ static void update_penalty(const Params &, double *pen, const double *,
                                          const double *, const double *)
    {
        pen[ETA] = -0.0;  // :546-561
    }
    // ocean_boundary_state! for first-order and gradient numerical fluxes
    __device__ static void boundary_state(const Params &m, int kind, int bctag, double *QP,
                                          double *auxP, const double *n, const double *QM,
                                          const double *auxM, double, const double *,
                                          const double *)
    {
        const int bc = m.bc[bctag - 1];
        if (bc == SE01_COAST_NOSLIP || bc == SE01_FLOOR_NOSLIP) {
            if (kind == BS_FIRST) {
                QP[U] = -QM[U];
                QP[V] = -QM[V];
            } else {
                QP[U] = -0.0;
                QP[V] = -0.0;
                if (bc == SE01_COAST_NOSLIP) auxP[AUD] = auxP[AUD + 1] = -0.0;
            }
        } else if (bc == SE01_COAST_FREESLIP) {
            const double f = kind == BS_FIRST ? 2.0 : 1.0;
            const double dn = n[0] * QM[U] + n[1] * QM[V];
            QP[U] = QM[U] - f * dn * n[0];
            QP[V] = QM[V] - f * dn * n[1];
            if (kind != BS_FIRST) {
                const double dd = n[0] * auxM[AUD] + n[1] * auxM[AUD + 1];
                auxP[AUD] = auxM[AUD] - dd * n[0];
                auxP[AUD + 1] = auxM[AUD + 1] - dd * n[1];
            }
        }
        if (bc == SE01_FLOOR_NOSLIP || bc == SE01_FLOOR_FREESLIP)
            auxP[AW] = kind == BS_FIRST ? -auxM[AW] : -0.0;
    }
    // boundary_state!(::NumericalFluxSecondOrder) then flux_second_order! of the plus side
    // (NumericalFluxes.jl:925-967)
    __device__ static void boundary_flux_second_order(
        const Params &m, int bctag, double *F, double *QP, double *gfP, double *hypP, double *auxP,
        const double *n, const double *QM, const double *, const double *, const double *auxM,
        double t, const double *, const double *, const double *)
    {
        const int bc = m.bc[bctag - 1];
        double st[2] = {-0.0, -0.0};  // D+.nu grad u = n (st)' unless it stays the minus side's
        bool keep = bc == SE01_COAST_NOSLIP || bc == SE01_FLOOR_NOSLIP;
        double fl = -0.0;             // D+.kappa grad theta = n fl
        if (bc == SE01_SURF_STRESS || bc == SE01_SURF_STRESS_FORCING) {
            const double tauz = -(m.tau_o / m.rho_o) * cos(auxM[AY] * M_PI / m.Ly);  // velocity_flux
            st[0] = -tauz;
        }
        if (bc == SE01_SURF_FORCING || bc == SE01_SURF_STRESS_FORCING) {
            const double thr = m.thE * (1 - auxM[AY] / m.Ly);
            fl = -(m.lam_r * (thr - QM[TH]));  // -n sigma_z
        }
        if (!keep) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 3; ++d) gfP[GNU + d + 3 * c] = n[d] * st[c];
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) gfP[GKAPPA + d] = n[d] * fl;
        flux_second_order(m, F, QP, gfP, hypP, auxP, t);
    }
};

// ---------------------------------------------------------------------------------------
// Continuity3dModel: the tendency of the theta slot of one evaluation is -grad_h . u
struct Continuity3dSE01 : SE01Defaults {
    using Params = OceanSE01Params;
    enum { U = 0, V = 1, ETA = 2, TH = 3 };
    static constexpr int NS = 4, NAUX = 0, NGRAD = 0, NGF = 0, NFAUX = 0;
    static constexpr bool HAS_SOURCE = false, HAS_PENALTY = false;
    __host__ __device__ static bool needs_gradflux(const Params &) { return false; }
    __host__ __device__ static constexpr int face_aux(int) { return 0; }
    static void make_params(Params &p, const int32_t *ip, const double *dp) { se01_make_params(p, ip, dp); }
    __device__ static void flux_first_order(const Params &, double *F, const double *Q,
                                            const double *, double, int)
    {
        const double v[3] = {Q[U], Q[V], -0.0};
#pragma unroll
        for (int d = 0; d < 3; ++d) F[d + 3 * TH] += v[d];
    }
    __device__ static void flux_second_order(const Params &, double *, const double *, const double *,
                                             const double *, const double *, double)
    {
    }
    __device__ static void source(const Params &, double *, const double *, const double *,
                                  const double *, const double *, double, int)
    {
    }
    __device__ static void gradient_argument(const Params &, double *, const double *, const double *,
                                             double)
    {
    }
    __device__ static void gradient_flux(const Params &, double *, const double *, const double *,
                                         const double *, double)
    {
    }
    __device__ static void wavespeed(const Params &, double *ws, const double *, const double *,
                                     const double *, double, int)
    {  // -zero: the jump of theta must not enter (Continuity3dModel.jl:48-50)
#pragma unroll
        for (int s = 0; s < NS; ++s) ws[s] = -0.0;
    }
    __device__ static void update_penalty(const Params &, double *, const double *, const double *,
                                          const double *)
    {
    }
    // boundary_conditions(cm) = (bc[1], bc[1], bc[1]): the coastline condition on every tag
    __device__ static void boundary_state(const Params &m, int kind, int, double *QP, double *,
                                          const double *n, const double *QM, const double *, double,
                                          const double *, const double *)
    {
        if (kind != BS_FIRST) return;
        const int bc = m.bc[0];
        if (bc == SE01_COAST_NOSLIP) {
            QP[U] = -QM[U];
            QP[V] = -QM[V];
        } else if (bc == SE01_COAST_FREESLIP) {
            const double dn = n[0] * QM[U] + n[1] * QM[V];
            QP[U] = QM[U] - 2 * dn * n[0];
            QP[V] = QM[V] - 2 * dn * n[1];
        }
    }
    __device__ static void boundary_flux_second_order(const Params &, int, double *, double *,
                                                      double *, double *, double *, const double *,
                                                      const double *, const double *,
                                                      const double *, const double *, double,
                                                      const double *, const double *,
                                                      const double *)
    {
    }
};

// ---------------------------------------------------------------------------------------
struct BarotropicSE01 : SE01Defaults {
    using Params = OceanSE01Params;
    enum { U1 = 0, U2 = 1, ETA = 2 };
    // G_U[2], U_c[2], eta_c, U_s[2], eta_s, Delta_u[2], eta_diag, Delta_eta, y  (BarotropicModel.jl:32-44)
    enum { AGU = 0, AUC = 2, AETAC = 4, AUS = 5, AETAS = 7, ADU = 8, AETAD = 10, ADETA = 11, AY = 12 };
    static constexpr int NS = 3, NAUX = 13, NGRAD = 2, NGF = 6, NFAUX = 0;
    static constexpr bool HAS_SOURCE = true, HAS_PENALTY = true;
    __host__ __device__ static bool needs_gradflux(const Params &) { return true; }
    __host__ __device__ static constexpr int face_aux(int) { return 0; }
    static void make_params(Params &p, const int32_t *ip, const double *dp) { se01_make_params(p, ip, dp); }
    __device__ static void flux_first_order(const Params &m, double *F, const double *Q,
                                            const double *, double, int)
    {  // :109-135
        const double Uv[3] = {Q[U1], Q[U2], 0.0};
#pragma unroll
        for (int d = 0; d < 3; ++d) F[d + 3 * ETA] += Uv[d];
        const double ghe = m.grav * m.H * Q[ETA];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) F[d + 3 * c] += ghe * (d == c ? 1.0 : 0.0);
    }
    __device__ static void flux_second_order(const Params &, double *F, const double *,
                                             const double *gf, const double *, const double *,
                                             double)
    {
#pragma unroll
        for (int q = 0; q < 6; ++q) F[q] += gf[q];
    }
    __device__ static void source(const Params &m, double *S, const double *Q, const double *,
                                  const double *aux, const double *, double, int)
    {  // :153-173
        const double f = m.fo + m.beta * aux[AY];
        S[U1] -= -f * Q[U2];
        S[U2] -= f * Q[U1];
        S[U1] += aux[AGU];
        S[U2] += aux[AGU + 1];
    }
    __device__ static void gradient_argument(const Params &, double *G, const double *Q,
                                             const double *, double)
    {
        G[0] = Q[U1];
        G[1] = Q[U2];
    }
    __device__ static void gradient_flux(const Params &m, double *D, const double *g,
                                         const double *, const double *, double)
    {  // -Diagonal(nu_h, nu_h, 0) * G.U
        const double nu[3] = {m.nuh, m.nuh, 0.0};
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) D[d + 3 * c] = -nu[d] * g[d + 3 * c];
    }
    __device__ static void wavespeed(const Params &m, double *ws, const double *n, const double *,
                                     const double *, double, int)
    {
        ws[0] = ws[1] = ws[2] = gravity_wavespeed(m, n);
    }
    __device__ static void update_penalty(const Params &, double *pen, const double *,
                                          const double *, const double *)
    {
        pen[ETA] = -0.0;
    }
    // boundary_conditions(bm) = (bc[1],)
    __device__ static void boundary_state(const Params &m, int kind, int, double *QP, double *,
                                          const double *n, const double *QM, const double *, double,
                                          const double *, const double *)
    {
        const int bc = m.bc[0];
        if (bc == SE01_COAST_NOSLIP) {
            QP[U1] = kind == BS_FIRST ? -QM[U1] : -0.0;
            QP[U2] = kind == BS_FIRST ? -QM[U2] : -0.0;
        } else if (bc == SE01_COAST_FREESLIP) {
            const double f = kind == BS_FIRST ? 2.0 : 1.0;
            const double dn = n[0] * QM[U1] + n[1] * QM[U2];
            QP[U1] = QM[U1] - f * dn * n[0];
            QP[U2] = QM[U2] - f * dn * n[1];
        }
    }
    __device__ static void boundary_flux_second_order(
        const Params &m, int, double *F, double *QP, double *gfP, double *hypP, double *auxP,
        const double *n, const double *, const double *, const double *, const double *, double t,
        const double *, const double *, const double *)
    {
        if (m.bc[0] == SE01_COAST_FREESLIP) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 3; ++d) gfP[d + 3 * c] = n[d] * -0.0;
        }  // CoastlineNoSlip: D+ = D-
        flux_second_order(m, F, QP, gfP, hypP, auxP, t);
    }
};

}  // namespace cmdg
