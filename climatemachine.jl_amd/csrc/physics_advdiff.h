// Device functor for the reference's test balance law AdvectionDiffusion{1}
// (test/Numerics/DGMethods/advection_diffusion/advection_diffusion_model.jl:92-617)
// and the problems that supply its coefficients and boundary data:
//   problem 0  Pseudo1D                 pseudo1D_advection_diffusion.jl:28-68
//   problem 1  ConstantHyperDiffusion   periodic_3D_hyperdiffusion.jl:29-63
//   problem 2  ConstantHyperDiffusion{mu, k} with boundary data  hyperdiffusion_bc.jl:25-112
//   problem 3  HeatEqn{n, kappa, A}                              pseudo1D_heat_eqn.jl:28-88
//   problem 7  ReversingDeformationalFlow (velocity refresh)     advection_sphere.jl:56-103
//
// Parameter block (cmdg_desc.iparam / dparam):
//   iparam[0]=num_equations (1)  [1]=advection [2]=diffusion [3]=hyperdiffusion
//   iparam[4]=flux_bc [5]=problem [6]=nbc [7..13]=bc bitmask of boundary tag 1..7
//   bc bit b: InhomogeneousBC{b} for b=0..3, HomogeneousBC{b-4} for b=4..7, bit 8 NoFlowBC
//   (advection_sphere.jl:118-132)
//   dparam: Pseudo1D n[3], alpha, beta, mu, delta | ConstantHyperDiffusion D[9], dim, dir
//
// Grad-type locals are 3 x nvar column-major: g[d + 3*s] (vars_wrappers.jl:52).
#pragma once
#include "cmdg_common.h"

namespace cmdg {

struct AdvDiffParams {
    int flux_bc, problem, nbc;
    int bc[8];
    double d[16];
};

template <bool ADV, bool DIFF, bool HYPER>
struct AdvDiff {
    using Params = AdvDiffParams;
    static constexpr int NS = 1;
    static constexpr int NAUX = 3 + (ADV ? 3 : 0) + (DIFF ? 9 : 0) + (HYPER ? 9 : 0);
    static constexpr int NGRAD = (DIFF || HYPER) ? 1 : 0;
    static constexpr int NGF = DIFF ? 3 : 0;
    static constexpr int NGL = HYPER ? 1 : 0;
    static constexpr int NHYP = HYPER ? 3 : 0;
    static constexpr int OU = 3, OD = 3 + (ADV ? 3 : 0), OH = OD + (DIFF ? 9 : 0);
    // only problems with variable coefficients refresh the auxiliary state (problem 7:
    // ReversingDeformationalFlow, advection_sphere.jl:56-103); see update_aux_active
    static constexpr bool HAS_UPDATE_AUX = ADV;
    static constexpr bool FUSE_UPDATE_AUX = false;
    static constexpr int NUPD = 0;
    __host__ __device__ static constexpr int upd_aux(int) { return 0; }
    static constexpr bool HAS_SOURCE = false;
    static constexpr int NDER = 0;
    __host__ __device__ static bool needs_gradflux(const Params &) { return DIFF; }
    // auxiliary fields the interior-face fluxes read on either side: the velocity (first-order
    // flux and wave speed); boundary faces load the whole minus-side auxiliary state themselves
    static constexpr int NFAUX = ADV ? 3 : 0;
    __host__ __device__ static constexpr int face_aux(int i) { return OU + i; }
    __host__ __device__ static constexpr int hv_indexmap(int) { return 0; }

    static constexpr int BC_INHOM(int o) { return 1 << o; }
    static constexpr int BC_HOM(int o) { return 1 << (o + 4); }
    static constexpr int BC_ANY(int o) { return BC_INHOM(o) | BC_HOM(o); }

    static void make_params(Params &p, const int32_t *ip, const double *dp)
    {
        p.flux_bc = ip[4];
        p.problem = ip[5];
        p.nbc = ip[6];
        for (int i = 0; i < 7; ++i) p.bc[i] = ip[7 + i];
        p.bc[7] = 0;
        for (int i = 0; i < 16; ++i) p.d[i] = dp[i];
    }

    // ---- problems ----------------------------------------------------------------
    __device__ static double problem_rho(const Params &m, const double *x, double t)
    {
        if (m.problem == 0) {  // Pseudo1D initial_condition! (:42-52)
            const double *n = m.d;
            const double al = m.d[3], be = m.d[4], mu = m.d[5], de = m.d[6];
            const double xn = n[0] * x[0] + n[1] * x[1] + n[2] * x[2];
            const double a = xn - mu - al * t;
            return exp(-(a * a) / (4 * be * (de + t))) / sqrt(1 + t / de);
        } else if (m.problem == 1) {  // ConstantHyperDiffusion (:44-63)
            // (every loop has constant bounds: an index that is only known at run time would send
            // the node's coordinates and the parameter block through scratch memory)
            const int dim = (int)m.d[9], dir = (int)m.d[10];
            const double k[3] = {1, 2, 3};
            double c;
            if (dir == 0 || dir == 1) {
                const int dd = dir == 0 ? dim : dim - 1;
                double s2 = 0, skd = 0;
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    if (i < dd) s2 += k[i] * k[i];
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        if (i < dd && j < dd) skd += k[i] * k[j] * m.d[i + 3 * j];
                c = s2 * skd;
            } else {
                const double kd = dim == 1 ? 1.0 : (dim == 2 ? 2.0 : 3.0);
                const double Dd = dim == 1 ? m.d[0] : (dim == 2 ? m.d[4] : m.d[8]);
                c = kd * kd * (kd * kd * Dd);
            }
            double kx = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < dim) kx += k[i] * x[i];
            return sin(kx) * exp(-c * t);
        } else if (m.problem == 2) {  // d[0] = mu, d[1..3] = k
            const double *k = m.d + 1;
            return cos(k[0] * x[0]) * cos(k[1] * x[1]) * cos(k[2] * x[2]) * hbc_e(m, t);
        } else if (m.problem == 3) {  // d[0..2] = n, d[3] = kappa, d[4] = A
            const double *n = m.d, ka = m.d[3], A = m.d[4];
            const double xn = n[0] * x[0] + n[1] * x[1] + n[2] * x[2];
            return xn + A * cos(ka * xn) * exp(-(ka * ka) * t);
        }
        return 0.0;
    }
    __device__ static double hbc_e(const Params &m, double t)
    {
        const double *k = m.d + 1;
        const double k2 = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
        return exp(-(k2 * k2) * m.d[0] * t);
    }
    __device__ static void hbc_sincos(const Params &m, const double *x, double *v)
    {
        const double *k = m.d + 1;
        v[0] = k[0] * sin(k[0] * x[0]) * cos(k[1] * x[1]) * cos(k[2] * x[2]);
        v[1] = k[1] * cos(k[0] * x[0]) * sin(k[1] * x[1]) * cos(k[2] * x[2]);
        v[2] = k[2] * cos(k[0] * x[0]) * cos(k[1] * x[1]) * sin(k[2] * x[2]);
    }
    // inhomogeneous_data!(Val(2)) / Val(3) of hyperdiffusion_bc.jl:80-112
    __device__ static double problem_lap(const Params &m, const double *x, double t)
    {
        if (m.problem != 2) return 0.0;
        const double *k = m.d + 1;
        const double k2 = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
        return -k2 * cos(k[0] * x[0]) * cos(k[1] * x[1]) * cos(k[2] * x[2]) * hbc_e(m, t);
    }
    __device__ static void problem_gradlap(const Params &m, double *g, const double *x, double t)
    {
        g[0] = g[1] = g[2] = 0.0;
        if (m.problem != 2) return;
        const double *k = m.d + 1;
        const double k2 = k[0] * k[0] + k[1] * k[1] + k[2] * k[2];
        double v[3];
        hbc_sincos(m, x, v);
        const double e = hbc_e(m, t);
        for (int i = 0; i < 3; ++i) g[i] = (k2 * v[i]) * e;
    }
    __device__ static void problem_grad(const Params &m, double *g, const double *x, double t)
    {
        if (m.problem == 0) {  // inhomogeneous_data!(Val(1), ::Pseudo1D, ...) (:54-68)
            const double *n = m.d;
            const double al = m.d[3], be = m.d[4], mu = m.d[5], de = m.d[6];
            const double xn = n[0] * x[0] + n[1] * x[1] + n[2] * x[2];
            const double a = xn - mu - al * t;
            for (int i = 0; i < 3; ++i)
                g[i] = -(2 * n[i] * a / (4 * be * (de + t)) *
                         exp(-(a * a) / (4 * be * (de + t))) / sqrt(1 + t / de));
        } else if (m.problem == 3) {  // exact gradient, pseudo1D_heat_eqn.jl:79-88
            const double *n = m.d, ka = m.d[3], A = m.d[4];
            const double xn = n[0] * x[0] + n[1] * x[1] + n[2] * x[2];
            for (int i = 0; i < 3; ++i)
                g[i] = n[i] * (1 - A * ka * sin(ka * xn) * exp(-(ka * ka) * t));
        } else if (m.problem == 2) {  // inhomogeneous_data!(Val(1)) hyperdiffusion_bc.jl:63-79
            double v[3];
            hbc_sincos(m, x, v);
            const double e = hbc_e(m, t);
            for (int i = 0; i < 3; ++i) g[i] = -v[i] * e;
        } else {
            g[0] = g[1] = g[2] = 0.0;
        }
    }

    // ---- balance law -------------------------------------------------------------
    __device__ static void flux_first_order(const Params &, double *F, const double *Q,
                                            const double *aux, double, int)
    {
        if constexpr (ADV)
            for (int d = 0; d < 3; ++d) F[d] += aux[OU + d] * Q[0];
    }
    __device__ static void flux_second_order(const Params &, double *F, const double *,
                                             const double *gf, const double *hyp, const double *,
                                             double)
    {
        if constexpr (DIFF)
            for (int d = 0; d < 3; ++d) F[d] += -gf[d];
        if constexpr (HYPER)
            for (int d = 0; d < 3; ++d) F[d] += hyp[d];
    }
    __device__ static void source(const Params &, double *, const double *, const double *,
                                  const double *, const double *, double, int)
    {
    }
    __device__ static void init_derived(const Params &, double *, const double *) {}
    __device__ static void gradient_argument(const Params &, double *G, const double *Q,
                                             const double *, double)
    {
        if constexpr (NGRAD > 0) G[0] = Q[0];
    }
    __device__ static void matvec3(double *o, const double *A, const double *v)
    {
        for (int i = 0; i < 3; ++i) o[i] = A[i] * v[0] + A[i + 3] * v[1] + A[i + 6] * v[2];
    }
    __device__ static void gradient_flux(const Params &, double *gf, const double *gradG,
                                         const double *, const double *aux, double)
    {
        if constexpr (DIFF) matvec3(gf, aux + OD, gradG);
    }
    __device__ static void post_gradient_laplacian(const Params &, double *hyp,
                                                   const double *gradlap, const double *,
                                                   const double *aux, double)
    {
        if constexpr (HYPER) matvec3(hyp, aux + OH, gradlap);
    }
    __device__ static void wavespeed(const Params &, double *ws, const double *n, const double *,
                                     const double *aux, double, int)
    {
        if constexpr (ADV)
            ws[0] = fabs(n[0] * aux[OU] + n[1] * aux[OU + 1] + n[2] * aux[OU + 2]);
        else
            ws[0] = 0.0;
    }
    // boundary_state!(nf, bcs, m, stateP, auxP, nM, stateM, auxM, t, _...)  (:402-428)
    __device__ static void boundary_state(const Params &m, int kind, int bctag, double *QP,
                                          double *auxP, const double *, const double *QM,
                                          const double *auxM, double t, const double *,
                                          const double *)
    {
        const int bc = m.bc[bctag - 1];
        if (bc & (1 << 8)) {  // NoFlowBC, first-order (Rusanov) flux only
            if constexpr (ADV) {
                if (kind == BS_FIRST) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) auxP[OU + d] = -auxM[OU + d];
                }
            }
            return;
        }
        if (bc & BC_INHOM(0))
            QP[0] = problem_rho(m, auxP, t);
        else if (bc & BC_ANY(1))
            QP[0] = QM[0];
        else if (bc & BC_HOM(0))
            QP[0] = 0.0;
    }
    // boundary_state!(::CentralNumericalFluxSecondOrder, ...) (:430-517) followed by
    // flux_second_order! (NumericalFluxes.jl:921-967), or the flux_bc method (:519-567)
    __device__ static void boundary_flux_second_order(
        const Params &m, int bctag, double *F, double *QP, double *gfP, double *hypP,
        double *auxP, const double *, const double *QM, const double *gfM, const double *hypM,
        const double *auxM, double t, const double *, const double *, const double *)
    {
        const int bc = m.bc[bctag - 1];
        double g[3];
        if constexpr (!DIFF && !HYPER) {
            return;
        } else {
            if (m.flux_bc) {
                if (bc & BC_ANY(0)) {
                    flux_second_order(m, F, QM, gfM, hypM, auxM, t);
                } else if (bc & BC_INHOM(1)) {
                    problem_grad(m, g, auxM, t);
                    const double *D = auxM + OD;
                    for (int i = 0; i < 3; ++i)
                        F[i] = -D[i] * g[0] + -D[i + 3] * g[1] + -D[i + 6] * g[2];
                } else if (bc & BC_HOM(1)) {
                    F[0] = F[1] = F[2] = 0.0;
                }
                return;
            }
            if constexpr (DIFF) {
                if (bc & BC_ANY(0)) {
                    for (int d = 0; d < 3; ++d) gfP[d] = gfM[d];
                } else if (bc & BC_INHOM(1)) {
                    problem_grad(m, g, auxM, t);
                    matvec3(gfP, auxM + OD, g);
                } else if (bc & BC_HOM(1)) {
                    g[0] = g[1] = g[2] = 0.0;
                    matvec3(gfP, auxM + OD, g);
                }
            }
            if constexpr (HYPER) {
                if (bc & BC_INHOM(3)) {
                    problem_gradlap(m, g, auxM, t);
                    matvec3(hypP, auxM + OH, g);
                } else if (bc & BC_HOM(3)) {
                    g[0] = g[1] = g[2] = 0.0;
                    matvec3(hypP, auxM + OH, g);
                }
            }
            flux_second_order(m, F, QP, gfP, hypP, auxP, t);
        }
    }
    // boundary_state!(::CentralNumericalFluxDivergence, ...) (:569-591)
    __device__ static void boundary_state_divergence(const Params &m, int bctag, double *gradP,
                                                     double *, const double *, const double *,
                                                     const double *auxM, double t)
    {
        if constexpr (HYPER) {
            const int bc = m.bc[bctag - 1];
            if (bc & BC_INHOM(1))
                problem_grad(m, gradP, auxM, t);
            else if (bc & BC_HOM(1))
                gradP[0] = gradP[1] = gradP[2] = 0.0;
        }
    }
    // boundary_state!(::CentralNumericalFluxHigherOrder, ...) (:593-617)
    __device__ static void boundary_state_higher_order(const Params &m, int bctag, double *,
                                                       double *, double *lapP, const double *,
                                                       const double *, const double *auxM,
                                                       const double *, double t)
    {
        if constexpr (HYPER) {
            const int bc = m.bc[bctag - 1];
            if (bc & BC_INHOM(2))
                lapP[0] = problem_lap(m, auxM, t);
            else if (bc & BC_HOM(2))
                lapP[0] = 0.0;
        }
    }
    __host__ __device__ static bool update_aux_active(const Params &m) { return ADV && m.problem == 7; }
    // update_velocity_diffusion!(::ReversingDeformationalFlow, ...)  advection_sphere.jl:76-101
    __device__ static void update_aux(const Params &m, const double *, double *aux, double t)
    {
        if constexpr (ADV) {
            if (m.problem != 7) return;
            const double x = aux[0], y = aux[1], z = aux[2];
            const double r = sqrt(x * x + y * y + z * z);
            const double lam = atan2(y, x), phi = asin(z / r);
            const double T = 5.0;
            const double lamp = lam - 2 * M_PI * t / T;
            const double sl = sin(lamp);
            const double ul = 10 * r / T * (sl * sl) * sin(2 * phi) * cos(M_PI * t / T) +
                              2 * M_PI * r / T * cos(phi);
            const double up = 10 * r / T * sin(2 * lamp) * cos(phi) * cos(M_PI * t / T);
            aux[OU + 0] = -ul * sin(lam) - up * cos(lam) * sin(phi);
            aux[OU + 1] = +ul * cos(lam) - up * sin(lam) * sin(phi);
            aux[OU + 2] = +up * cos(phi);
        }
    }
    // the test law defines no local_courant function
    static constexpr bool HAS_COURANT = false;
    static constexpr bool HAS_PENALTY = false;  // update_penalty! is the default no-op
    __device__ static void update_penalty(const Params &, double *, const double *, const double *,
                                          const double *)
    {
    }
    __device__ static double courant(const Params &, int, const double *, const double *,
                                     const double *, double, double, double, int)
    {
        return 0.0;
    }
};

}  // namespace cmdg
