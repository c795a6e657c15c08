// Device functor for the reference's PressureGradientModel (src/Atmos/Model/ref_state.jl:196-233):
// a mini balance law whose tendency is the DG gradient of the auxiliary field p, "computed as
// div(p I) ... to be numerically consistent with the way this gradient is computed in the
// dynamics".  State: grad p (3); auxiliary: p (1); first-order flux F.grad_p -= p I; no
// source, no second-order terms, boundary_state! does nothing (plus side = minus side).
#pragma once
#include "cmdg_common.h"

namespace cmdg {

struct PGradParams {
    int unused;
};

struct PressureGradient {
    using Params = PGradParams;
    static constexpr int NS = 3, NAUX = 1, NGRAD = 0, NGF = 0, NGL = 0, NHYP = 0;
    static constexpr bool HAS_UPDATE_AUX = false, FUSE_UPDATE_AUX = false, HAS_SOURCE = false;
    static constexpr bool HAS_COURANT = false, HAS_PENALTY = false;
    static constexpr int NUPD = 0, NDER = 0, NFAUX = 1;
    __host__ __device__ static constexpr int upd_aux(int) { return 0; }
    __host__ __device__ static constexpr int hv_indexmap(int) { return 0; }
    __host__ __device__ static constexpr int face_aux(int) { return 0; }
    __host__ __device__ static bool needs_gradflux(const Params &) { return false; }
    static void make_params(Params &p, const int32_t *, const double *) { p.unused = 0; }

    __device__ static void flux_first_order(const Params &, double *F, const double *,
                                            const double *aux, double, int)
    {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int d = 0; d < 3; ++d) F[d + 3 * c] -= aux[0] * (d == c ? 1.0 : 0.0);
    }
    __device__ static void flux_second_order(const Params &, double *, const double *,
                                             const double *, const double *, const double *, double)
    {
    }
    __device__ static void source(const Params &, double *, const double *, const double *,
                                  const double *, const double *, double, int)
    {
    }
    __device__ static void init_derived(const Params &, double *, const double *) {}
    __device__ static void gradient_argument(const Params &, double *, const double *,
                                             const double *, double)
    {
    }
    __device__ static void gradient_flux(const Params &, double *, const double *, const double *,
                                         const double *, double)
    {
    }
    __device__ static void post_gradient_laplacian(const Params &, double *, const double *,
                                                   const double *, const double *, double)
    {
    }
    __device__ static void wavespeed(const Params &, double *ws, const double *, const double *,
                                     const double *, double, int)
    {
        ws[0] = ws[1] = ws[2] = 0.0;
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
    __host__ __device__ static bool update_aux_active(const Params &) { return false; }
    __device__ static double courant(const Params &, int, const double *, const double *,
                                     const double *, double, double, double, int)
    {
        return 0.0;
    }
};

}  // namespace cmdg
