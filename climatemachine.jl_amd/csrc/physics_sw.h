// Device functor for the ShallowWaterModel (src/Ocean/ShallowWater/ShallowWaterModel.jl): state
// eta, U[2]; auxiliary y, G_U[2], Delta_u[2]; gradient U[2]; gradient flux nu grad U (3 x 2).
// :146-156, :178-191 (gradient argument / flux, ConstantViscosity), :193-233 (first-order and
// advective flux), :246-258 (second-order flux), :260 (wavespeed), :262-285 (source) and the
// Coupled forcing src/Ocean/SplitExplicit/ShallowWaterCoupling.jl:3-7.  The reference runs this
// law on a 2-D grid; here it runs on the 3-D kernels over a one-layer periodic extrusion of
// that grid (fields constant along the extrusion, third flux component identically zero).
// Periodic boxes only (no boundary tags in scope).
//
// iparam[0] advection, [1] turbulence (0 ConstantViscosity, 1 LinearDrag), [2] Coriolis kind,
// [3] coupling; dparam[0..5] = grav H c nu_or_lambda f_o beta.
#pragma once
#include "cmdg_common.h"

namespace cmdg {

struct SWParams {
    int adv, drag, cor, coupled;
    double grav, H, c, nu, fo, beta;
};

struct ShallowWater {
    using Params = SWParams;
    enum { ETA = 0, U1 = 1, U2 = 2 };
    enum { AY = 0, AG = 1, ADU = 3 };
    static constexpr int NS = 3, NAUX = 5, NGRAD = 2, NGF = 6, NGL = 0, NHYP = 0;
    static constexpr bool HAS_UPDATE_AUX = false, FUSE_UPDATE_AUX = false, HAS_SOURCE = true;
    static constexpr bool HAS_COURANT = false, HAS_PENALTY = false;
    static constexpr int NUPD = 0, NDER = 0, NFAUX = 0;
    __host__ __device__ static constexpr int upd_aux(int) { return 0; }
    __host__ __device__ static constexpr int hv_indexmap(int) { return 0; }
    __host__ __device__ static constexpr int face_aux(int) { return 0; }
    __host__ __device__ static bool needs_gradflux(const Params &m) { return !m.drag; }
    __host__ __device__ static bool update_aux_active(const Params &) { return false; }
    static void make_params(Params &p, const int32_t *ip, const double *dp)
    {
        p.adv = ip[0];
        p.drag = ip[1];
        p.cor = ip[2];
        p.coupled = ip[3];
        p.grav = dp[0];
        p.H = dp[1];
        p.c = dp[2];
        p.nu = dp[3];
        p.fo = dp[4];
        p.beta = dp[5];
    }
    __device__ static void flux_first_order(const Params &m, double *F, const double *Q,
                                            const double *, double, int)
    {
        const double Uv[3] = {Q[U1], Q[U2], -0.0};
#pragma unroll
        for (int d = 0; d < 3; ++d) F[d + 3 * ETA] += Uv[d];
        const double ghe = m.grav * m.H * Q[ETA];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) F[d + 3 * (U1 + c)] += ghe * (d == c ? 1.0 : -0.0);
        if (m.adv) {
            const double Hinv = 1 / m.H;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < 3; ++d) F[d + 3 * (U1 + c)] += Hinv * Uv[d] * Q[U1 + c];
        }
    }
    __device__ static void flux_second_order(const Params &m, double *F, const double *,
                                             const double *gf, const double *, const double *,
                                             double)
    {
        if (m.drag) return;
#pragma unroll
        for (int q = 0; q < 6; ++q) F[3 * U1 + q] += gf[q];
    }
    __device__ static void source(const Params &m, double *S, const double *Q, const double *,
                                  const double *aux, const double *, double, int)
    {
        const double f = m.cor == 0 ? -0.0 : (m.cor == 1 ? m.fo : m.fo + m.beta * aux[AY]);
        S[U1] -= -f * Q[U2];
        S[U2] -= f * Q[U1];
        if (m.coupled) {  // forcing_term!(::Coupled): S.U += A.G_U
            S[U1] += aux[AG];
            S[U2] += aux[AG + 1];
        } else {  // kinematic_stress(::SimpleBox, y) = [-0, -0]
            S[U1] += -0.0;
            S[U2] += -0.0;
        }
        if (m.drag) {
            S[U1] -= m.nu * Q[U1];
            S[U2] -= m.nu * Q[U2];
        }
    }
    __device__ static void init_derived(const Params &, double *, const double *) {}
    __device__ static void gradient_argument(const Params &m, double *G, const double *Q,
                                             const double *, double)
    {
        if (m.drag) return;
        G[0] = Q[U1];
        G[1] = Q[U2];
    }
    __device__ static void gradient_flux(const Params &m, double *D, const double *g,
                                         const double *, const double *, double)
    {
        if (m.drag) return;
        const double nu[3] = {m.nu, m.nu, -0.0};
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) D[d + 3 * c] = -nu[d] * g[d + 3 * c];
    }
    __device__ static void post_gradient_laplacian(const Params &, double *, const double *,
                                                   const double *, const double *, double)
    {
    }
    __device__ static void wavespeed(const Params &m, double *ws, const double *, const double *,
                                     const double *, double, int)
    {
        ws[0] = ws[1] = ws[2] = m.c;
    }
    __device__ static void update_penalty(const Params &, double *, const double *, const double *,
                                          const double *)
    {
    }
    __device__ static void boundary_state(const Params &, int, int, double *, double *,
                                          const double *, const double *, const double *, double,
                                          const double *, const double *)
    {
    }
    __device__ static void boundary_flux_second_order(const Params &, int, double *, double *,
                                                      double *, double *, double *, const double *,
                                                      const double *, const double *,
                                                      const double *, const double *, double,
                                                      const double *, const double *,
                                                      const double *)
    {
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
    __device__ static double courant(const Params &, int, const double *, const double *,
                                     const double *, double, double, double, int)
    {
        return 0.0;
    }
};

}  // namespace cmdg
