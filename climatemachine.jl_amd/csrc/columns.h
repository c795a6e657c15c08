// Column (stack) operators: kernel_indefinite_stack_integral! (DGModel_kernels.jl:1903-2010)
// and kernel_reverse_indefinite_stack_integral! (:2012-2104).  The integrands are carried as
// a field-combination descriptor (cmdg_stack_integral_desc): integrand_s = scale_s * field_s.
//
// The reference launches one (Nq x Nq)-thread work-group per stack; here a 256-thread block
// walks SPB = 256 / Nq^2 stacks side by side (one thread per (stack, i, j) pencil), which keeps
// the sequential walk up the stack but gives the memory system 10 independent columns per
// block.  The running integral uses the reference's order: kernel value times JcV, then
// sum_n Imat[k, n] * kernel[n] added to the value carried from the element below.
#pragma once
#include "cmdg_common.h"

namespace cmdg {

constexpr int STACK_MAXOUT = 8;

struct StackArgs {
    const double *Q;
    double *aux;
    const double *vgeo;
    const double *Imat;  // device (NQ, NQ) column-major
    int nstate, naux, nvgeo, nvert, jcv;
    int64_t h0, nhorz;  // horizontal elements [h0, h0 + nhorz)
    int is_state[STACK_MAXOUT], src[STACK_MAXOUT], dst[STACK_MAXOUT];
    double scale[STACK_MAXOUT];
};

template <int NQ, int NOUT>
__global__ __launch_bounds__(256) void k_stack_integral(StackArgs a)
{
    constexpr int Nij = NQ * NQ, Np = Nij * NQ, SPB = 256 / Nij;
    __shared__ double sI[NQ * NQ];
    const int tid = threadIdx.x;
    if (tid < NQ * NQ) sI[tid] = a.Imat[tid];
    __syncthreads();
    const int sl = tid / Nij, ij = tid % Nij;
    const int64_t eh = a.h0 + (int64_t)blockIdx.x * SPB + sl;
    if (sl >= SPB || eh >= a.h0 + a.nhorz) return;
    double lint[NOUT][NQ], lker[NOUT][NQ], lnext[NOUT][NQ];
#pragma unroll
    for (int s = 0; s < NOUT; ++s)
#pragma unroll
        for (int k = 0; k < NQ; ++k) lint[s][k] = 0;
    // integrand of one element of the stack (integral_load_auxiliary_state! times the Jacobian)
    auto load_kernel = [&](int ev, double (&out)[NOUT][NQ]) {
        const int64_t e = ev + eh * a.nvert;
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int ijk = ij + Nij * k;
            const double Jc = a.vgeo[ijk + (int64_t)Np * (a.jcv + (int64_t)a.nvgeo * e)];
#pragma unroll
            for (int s = 0; s < NOUT; ++s) {
                const double f = a.is_state[s]
                                     ? a.Q[ijk + (int64_t)Np * (a.src[s] + (int64_t)a.nstate * e)]
                                     : a.aux[ijk + (int64_t)Np * (a.src[s] + (int64_t)a.naux * e)];
                out[s][k] = (a.scale[s] * f) * Jc;
            }
        }
    };
    load_kernel(0, lker);
    for (int ev = 0; ev < a.nvert; ++ev) {
        const int64_t e = ev + eh * a.nvert;
        // the next element's integrand is requested before this element's results are stored:
        // source and destination may be the same column of the same array, which keeps the
        // compiler from moving these loads above the stores by itself, and the walk up the
        // stack would pay a full memory latency per element
        if (ev + 1 < a.nvert) load_kernel(ev + 1, lnext);
#pragma unroll
        for (int s = 0; s < NOUT; ++s)
#pragma unroll
            for (int k = 0; k < NQ; ++k)
#pragma unroll
                for (int n = 0; n < NQ; ++n) lint[s][k] += sI[k + NQ * n] * lker[s][n];
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int ijk = ij + Nij * k;
#pragma unroll
            for (int s = 0; s < NOUT; ++s) {
                a.aux[ijk + (int64_t)Np * (a.dst[s] + (int64_t)a.naux * e)] = lint[s][k];
                lint[s][k] = lint[s][NQ - 1];
            }
        }
        if (ev + 1 < a.nvert) {  // (nothing was prefetched behind the last element)
#pragma unroll
            for (int s = 0; s < NOUT; ++s)
#pragma unroll
                for (int k = 0; k < NQ; ++k) lker[s][k] = lnext[s][k];
        }
    }
}

// reads aux column src[s] (the upward integral), writes top - value to aux column dst[s]
template <int NQ, int NOUT>
__global__ __launch_bounds__(256) void k_reverse_stack_integral(StackArgs a)
{
    constexpr int Nij = NQ * NQ, Np = Nij * NQ, SPB = 256 / Nij;
    const int tid = threadIdx.x;
    const int sl = tid / Nij, ij = tid % Nij;
    const int64_t eh = a.h0 + (int64_t)blockIdx.x * SPB + sl;
    if (sl >= SPB || eh >= a.h0 + a.nhorz) return;
    double lT[NOUT];
    {
        const int ijk = ij + Nij * (NQ - 1);
        const int64_t et = (a.nvert - 1) + eh * a.nvert;
#pragma unroll
        for (int s = 0; s < NOUT; ++s)
            lT[s] = a.aux[ijk + (int64_t)Np * (a.src[s] + (int64_t)a.naux * et)];
    }
    // every element's values are read before anything is stored, and the next element's before
    // this element's stores (source and destination are columns of one array: the compiler may
    // not reorder them, and a load-store pair per value would serialise 5 x NOUT x nvert latencies)
    double v[NOUT][NQ], vnext[NOUT][NQ];
    auto load_values = [&](int ev, double (&out)[NOUT][NQ]) {
        const int64_t e = ev + eh * a.nvert;
#pragma unroll
        for (int k = 0; k < NQ; ++k)
#pragma unroll
            for (int s = 0; s < NOUT; ++s)
                out[s][k] = a.aux[ij + Nij * k + (int64_t)Np * (a.src[s] + (int64_t)a.naux * e)];
    };
    load_values(0, v);
    for (int ev = 0; ev < a.nvert; ++ev) {
        const int64_t e = ev + eh * a.nvert;
        if (ev + 1 < a.nvert) load_values(ev + 1, vnext);
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int ijk = ij + Nij * k;
#pragma unroll
            for (int s = 0; s < NOUT; ++s)
                a.aux[ijk + (int64_t)Np * (a.dst[s] + (int64_t)a.naux * e)] = lT[s] - v[s][k];
        }
        if (ev + 1 < a.nvert) {
#pragma unroll
            for (int s = 0; s < NOUT; ++s)
#pragma unroll
                for (int k = 0; k < NQ; ++k) v[s][k] = vnext[s][k];
        }
    }
}

// The column operators of a law's update_auxiliary_state_gradient! in ONE walk up and (where
// something needs the top value) one walk down the stack -- copy of a gradient-flux column into
// the integrand's column, upward integrals, `top - value` reverse integral, surface value through
// the column (hydrostatic_boussinesq_model.jl:693-726: four launches in the recorded composition,
// 107 us per evaluation on the 48 x 48 x 16 box; this one moves 7 column passes instead of 11).
// Same arithmetic in the same order as k_scaled_column_copy + k_stack_integral +
// k_reverse_stack_integral + k_surface_to_column, value for value.
struct ChainArgs {
    StackArgs a;                    // the upward integrals (is_state / src / scale / dst)
    const double *gf;               // state_gradient_flux (NULL: no fused copy)
    int ngf;
    int gf_col[STACK_MAXOUT];       // >= 0: integrand s = gf_scale[s] * gf[:, gf_col[s]] (the copy into
    double gf_scale[STACK_MAXOUT];  // aux column src[s] is skipped: the integral overwrites it)
    int rev_dst[STACK_MAXOUT];      // >= 0: aux[:, rev_dst[s]] = top of output s - output s
    int surf_dst[STACK_MAXOUT];     // >= 0: aux[:, surf_dst[s]] = top of output s, every node
};

template <int NQ, int NOUT>
__global__ __launch_bounds__(256) void k_column_chain(ChainArgs c)
{
    const StackArgs &a = c.a;
    constexpr int Nij = NQ * NQ, Np = Nij * NQ, SPB = 256 / Nij;
    __shared__ double sI[NQ * NQ];
    const int tid = threadIdx.x;
    if (tid < NQ * NQ) sI[tid] = a.Imat[tid];
    __syncthreads();
    const int sl = tid / Nij, ij = tid % Nij;
    const int64_t eh = a.h0 + (int64_t)blockIdx.x * SPB + sl;
    if (sl >= SPB || eh >= a.h0 + a.nhorz) return;
    double lint[NOUT][NQ], lker[NOUT][NQ], lnext[NOUT][NQ];
#pragma unroll
    for (int s = 0; s < NOUT; ++s)
#pragma unroll
        for (int k = 0; k < NQ; ++k) lint[s][k] = 0;
    auto load_kernel = [&](int ev, double (&out)[NOUT][NQ]) {
        const int64_t e = ev + eh * a.nvert;
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int ijk = ij + Nij * k;
            const double Jc = a.vgeo[ijk + (int64_t)Np * (a.jcv + (int64_t)a.nvgeo * e)];
#pragma unroll
            for (int s = 0; s < NOUT; ++s) {
                const double f =
                    c.gf_col[s] >= 0
                        ? c.gf_scale[s] * c.gf[ijk + (int64_t)Np * (c.gf_col[s] + (int64_t)c.ngf * e)]
                        : (a.is_state[s] ? a.Q[ijk + (int64_t)Np * (a.src[s] + (int64_t)a.nstate * e)]
                                         : a.aux[ijk + (int64_t)Np * (a.src[s] + (int64_t)a.naux * e)]);
                out[s][k] = (a.scale[s] * f) * Jc;
            }
        }
    };
    load_kernel(0, lker);
    for (int ev = 0; ev < a.nvert; ++ev) {
        const int64_t e = ev + eh * a.nvert;
        if (ev + 1 < a.nvert) load_kernel(ev + 1, lnext);
#pragma unroll
        for (int s = 0; s < NOUT; ++s)
#pragma unroll
            for (int k = 0; k < NQ; ++k)
#pragma unroll
                for (int n = 0; n < NQ; ++n) lint[s][k] += sI[k + NQ * n] * lker[s][n];
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int ijk = ij + Nij * k;
#pragma unroll
            for (int s = 0; s < NOUT; ++s) {
                a.aux[ijk + (int64_t)Np * (a.dst[s] + (int64_t)a.naux * e)] = lint[s][k];
                lint[s][k] = lint[s][NQ - 1];
            }
        }
        if (ev + 1 < a.nvert) {
#pragma unroll
            for (int s = 0; s < NOUT; ++s)
#pragma unroll
                for (int k = 0; k < NQ; ++k) lker[s][k] = lnext[s][k];
        }
    }
    // ---- down again: whatever needs the value at the top of the stack (lint[s][*] now)
    bool second = false;
#pragma unroll
    for (int s = 0; s < NOUT; ++s) second = second || c.rev_dst[s] >= 0 || c.surf_dst[s] >= 0;
    if (!second) return;
    double v[NOUT][NQ], vnext[NOUT][NQ];
    auto load_values = [&](int ev, double (&out)[NOUT][NQ]) {  // (this thread's own stores of the walk up)
        const int64_t e = ev + eh * a.nvert;
#pragma unroll
        for (int k = 0; k < NQ; ++k)
#pragma unroll
            for (int s = 0; s < NOUT; ++s)
                out[s][k] = c.rev_dst[s] >= 0
                                ? a.aux[ij + Nij * k + (int64_t)Np * (a.dst[s] + (int64_t)a.naux * e)]
                                : 0.0;
    };
    load_values(0, v);
    for (int ev = 0; ev < a.nvert; ++ev) {
        const int64_t e = ev + eh * a.nvert;
        if (ev + 1 < a.nvert) load_values(ev + 1, vnext);
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int ijk = ij + Nij * k;
#pragma unroll
            for (int s = 0; s < NOUT; ++s) {
                if (c.rev_dst[s] >= 0)
                    a.aux[ijk + (int64_t)Np * (c.rev_dst[s] + (int64_t)a.naux * e)] = lint[s][0] - v[s][k];
                if (c.surf_dst[s] >= 0)
                    a.aux[ijk + (int64_t)Np * (c.surf_dst[s] + (int64_t)a.naux * e)] = lint[s][0];
            }
        }
        if (ev + 1 < a.nvert) {
#pragma unroll
            for (int s = 0; s < NOUT; ++s)
#pragma unroll
                for (int k = 0; k < NQ; ++k) v[s][k] = vnext[s][k];
        }
    }
}

// compute_flow_deviation! (HydrostaticBoussinesqCoupling.jl:43-85) in one launch: the upward
// integral of Q[:, ucol + c] (c = 0, 1; VerticalIntegralModel.jl:60-81, only its top value is
// needed) and dst[:, dcol + c] = Q[:, ucol + c] - top / H.  Same arithmetic as k_stack_integral<NQ, 2>
// followed by k_column_minus_top_over_H.
template <int NQ>
__global__ __launch_bounds__(256) void k_flow_deviation(const double *__restrict__ Q, int nstate, int ucol,
                                                        double *__restrict__ dst, int ndst, int dcol,
                                                        const double *__restrict__ vgeo, int nvgeo, int jcv,
                                                        const double *__restrict__ Imat, double H, int nvert,
                                                        int64_t h0, int64_t nhorz)
{
    constexpr int Nij = NQ * NQ, Np = Nij * NQ, SPB = 256 / Nij;
    __shared__ double sI[NQ * NQ];
    const int tid = threadIdx.x;
    if (tid < NQ * NQ) sI[tid] = Imat[tid];
    __syncthreads();
    const int sl = tid / Nij, ij = tid % Nij;
    const int64_t eh = h0 + (int64_t)blockIdx.x * SPB + sl;
    if (sl >= SPB || eh >= h0 + nhorz) return;
    double lint[2][NQ], lker[2][NQ], lnext[2][NQ];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int k = 0; k < NQ; ++k) lint[s][k] = 0;
    auto load_kernel = [&](int ev, double (&out)[2][NQ]) {
        const int64_t e = ev + eh * nvert;
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int ijk = ij + Nij * k;
            const double Jc = vgeo[ijk + (int64_t)Np * (jcv + (int64_t)nvgeo * e)];
#pragma unroll
            for (int s = 0; s < 2; ++s)
                out[s][k] = (1.0 * Q[ijk + (int64_t)Np * (ucol + s + (int64_t)nstate * e)]) * Jc;
        }
    };
    load_kernel(0, lker);
    for (int ev = 0; ev < nvert; ++ev) {
        if (ev + 1 < nvert) load_kernel(ev + 1, lnext);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int k = 0; k < NQ; ++k)
#pragma unroll
                for (int n = 0; n < NQ; ++n) lint[s][k] += sI[k + NQ * n] * lker[s][n];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int k = 0; k < NQ; ++k) lint[s][k] = lint[s][NQ - 1];
        if (ev + 1 < nvert) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int k = 0; k < NQ; ++k) lker[s][k] = lnext[s][k];
        }
    }
    for (int ev = 0; ev < nvert; ++ev) {
        const int64_t e = ev + eh * nvert;
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int ijk = ij + Nij * k;
#pragma unroll
            for (int s = 0; s < 2; ++s)
                dst[ijk + (int64_t)Np * (dcol + s + (int64_t)ndst * e)] =
                    Q[ijk + (int64_t)Np * (ucol + s + (int64_t)nstate * e)] - lint[s][0] / H;
        }
    }
}

// aux[:, dst, e] = scale * src[:, scol, e] for elements [e0, e1)
static __global__ void k_scaled_column_copy(double *__restrict__ aux, int naux, int dst,
                                            const double *__restrict__ src, int nsrc, int scol,
                                            double scale, int Np, int64_t e0, int64_t e1)
{
    const int64_t n = (e1 - e0) * Np;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = e0 + i / Np;
        const int ijk = (int)(i % Np);
        aux[ijk + (int64_t)Np * (dst + (int64_t)naux * e)] =
            scale * src[ijk + (int64_t)Np * (scol + (int64_t)nsrc * e)];
    }
}

// aux[(i,j,k), dst, every element of the stack] = aux[(i,j,top k), src, top element]
// (the `boxy_wz0 .= flat_wz0` broadcast of hydrostatic_boussinesq_model.jl:717-723)
static __global__ void k_surface_to_column(double *__restrict__ aux, int naux, int src, int dst,
                                           int Nij, int Nqk, int nvert, int64_t h0, int64_t nhorz)
{
    const int Np = Nij * Nqk;
    const int64_t n = nhorz * nvert * Np;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ijk = (int)(i % Np);
        const int64_t ecol = i / Np;  // element counted from h0 * nvert
        const int64_t eh = h0 + ecol / nvert;
        const int64_t e = h0 * nvert + ecol;
        const int64_t et = (nvert - 1) + eh * nvert;
        const int ij = ijk % Nij;
        aux[ijk + (int64_t)Np * (dst + (int64_t)naux * e)] =
            aux[ij + Nij * (Nqk - 1) + (int64_t)Np * (src + (int64_t)naux * et)];
    }
}

// dst[(i,j,k), dcol + c, e] = src[(i,j,k), scol + c, e] - top[(i,j), c, stack] / H for c = 0, 1,
// where top is the value of the column integral `ia` (Np, 2, nelem) at the top node of the top
// element.  dst == src, dcol == scol gives the in-place `dG_u .-= int du / H` of
// Communication.jl:64-65; dst = aux.u_d, src = Q.u the flow deviation of
// HydrostaticBoussinesqCoupling.jl:78-84.
static __global__ void k_column_minus_top_over_H(double *__restrict__ dst, int ndst, int dcol,
                                                 const double *src, int nsrc, int scol,
                                                 const double *__restrict__ ia, double H, int Nij,
                                                 int Nqk, int nvert, int64_t h0, int64_t nhorz)
{
    const int Np = Nij * Nqk;
    const int64_t n = nhorz * nvert * Np;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ijk = (int)(i % Np);
        const int64_t e = h0 * nvert + i / Np;  // stacks [h0, h0 + nhorz)
        const int64_t et = (nvert - 1) + (e / nvert) * nvert;
        const int top = ijk % Nij + Nij * (Nqk - 1);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const double T = ia[top + (int64_t)Np * (c + 2 * et)];
            dst[ijk + (int64_t)Np * (dcol + c + (int64_t)ndst * e)] =
                src[ijk + (int64_t)Np * (scol + c + (int64_t)nsrc * e)] - T / H;
        }
    }
}

// layer[(i,j,k2), lcol + c, eh] = top[(i,j), c, stack eh]: `G_U .= int du` onto the one-layer
// extrusion of the 2-D grid (Communication.jl:59-61)
static __global__ void k_top_to_layer(double *__restrict__ layer, int nlayer, int lcol,
                                      const double *__restrict__ ia, int Nij, int Nqk3, int nvert,
                                      int Nqk2, int64_t nhorz)
{
    const int Np3 = Nij * Nqk3, Np2 = Nij * Nqk2;
    const int64_t n = nhorz * Np2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ijk = (int)(i % Np2);
        const int64_t eh = i / Np2;
        const int64_t et = (nvert - 1) + eh * nvert;
        const int top = ijk % Nij + Nij * (Nqk3 - 1);
#pragma unroll
        for (int c = 0; c < 2; ++c)
            layer[ijk + (int64_t)Np2 * (lcol + c + (int64_t)nlayer * eh)] =
                ia[top + (int64_t)Np3 * (c + 2 * et)];
    }
}

// reconcile_from_fast_to_slow! (Communication.jl:100-170), 2-D part: A2.du = 1/H (U - int u)
static __global__ void k_reconcile_layer(double *__restrict__ A2, int naux2, int ducol,
                                         const double *__restrict__ Q2, int ns2, int Ucol,
                                         const double *__restrict__ ia, double H, int Nij,
                                         int Nqk3, int nvert, int Nqk2, int64_t nhorz)
{
    const int Np3 = Nij * Nqk3, Np2 = Nij * Nqk2;
    const int64_t n = nhorz * Np2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ijk = (int)(i % Np2);
        const int64_t eh = i / Np2;
        const int64_t et = (nvert - 1) + eh * nvert;
        const int ij = ijk % Nij;
        const int top = ij + Nij * (Nqk3 - 1);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const double U = Q2[ij + (int64_t)Np2 * (Ucol + c + (int64_t)ns2 * eh)];
            A2[ijk + (int64_t)Np2 * (ducol + c + (int64_t)naux2 * eh)] =
                1 / H * (U - ia[top + (int64_t)Np3 * (c + 2 * et)]);
        }
    }
}

// ... 3-D part: Q3.u += du through the column, Q3.eta = Q2.eta
static __global__ void k_reconcile_column(double *__restrict__ Q3, int ns3, int ucol, int etacol,
                                          const double *__restrict__ Q2, int ns2, int Ucol,
                                          int eta2col, const double *__restrict__ ia, double H,
                                          int Nij, int Nqk3, int nvert, int Nqk2, int64_t nhorz)
{
    const int Np3 = Nij * Nqk3, Np2 = Nij * Nqk2;
    const int64_t n = nhorz * nvert * Np3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ijk = (int)(i % Np3);
        const int64_t e = i / Np3;
        const int64_t eh = e / nvert;
        const int64_t et = (nvert - 1) + eh * nvert;
        const int ij = ijk % Nij;
        const int top = ij + Nij * (Nqk3 - 1);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const double U = Q2[ij + (int64_t)Np2 * (Ucol + c + (int64_t)ns2 * eh)];
            const double du = 1 / H * (U - ia[top + (int64_t)Np3 * (c + 2 * et)]);
            Q3[ijk + (int64_t)Np3 * (ucol + c + (int64_t)ns3 * e)] += du;
        }
        Q3[ijk + (int64_t)Np3 * (etacol + (int64_t)ns3 * e)] =
            Q2[ij + (int64_t)Np2 * (eta2col + (int64_t)ns2 * eh)];
    }
}

// aux[:, col .. col + ncol - 1, real elements] = value
static __global__ void k_fill_columns(double *__restrict__ A, int nA, int col, int ncol,
                                      double value, int Np, int64_t nelems)
{
    const int64_t n = nelems * ncol * Np;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = i / ((int64_t)ncol * Np);
        const int r = (int)(i % ((int64_t)ncol * Np));
        A[r + (int64_t)Np * (col + (int64_t)nA * e)] = value;
    }
}

}  // namespace cmdg
