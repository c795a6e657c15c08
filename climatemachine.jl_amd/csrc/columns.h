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
    double lint[NOUT][NQ], lker[NOUT][NQ];
#pragma unroll
    for (int s = 0; s < NOUT; ++s)
#pragma unroll
        for (int k = 0; k < NQ; ++k) lint[s][k] = 0;
    for (int ev = 0; ev < a.nvert; ++ev) {
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
                lker[s][k] = (a.scale[s] * f) * Jc;
            }
        }
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
    for (int ev = 0; ev < a.nvert; ++ev) {
        const int64_t e = ev + eh * a.nvert;
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const int ijk = ij + Nij * k;
#pragma unroll
            for (int s = 0; s < NOUT; ++s) {
                const double v = a.aux[ijk + (int64_t)Np * (a.src[s] + (int64_t)a.naux * e)];
                a.aux[ijk + (int64_t)Np * (a.dst[s] + (int64_t)a.naux * e)] = lT[s] - v;
            }
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

}  // namespace cmdg
