// Element filters (src/Numerics/Mesh/Filters.jl).  One work-group per element, like the
// reference's kernels, with the tensor passes staged through LDS:
//   k_apply_filter       kernel_apply_filter!       Filters.jl:651-794
//   k_apply_mp_filter    kernel_apply_mp_filter!    Filters.jl:893-1071
//   k_apply_tmar_filter  kernel_apply_TMAR_filter!  Filters.jl:796-884
// Targets: FilterIndices (Filters.jl:72-100), AtmosFilterPerturbations and
// AtmosSpecificFilterPerturbations (src/Atmos/Model/filters.jl:4-118, dry model).
//
// The reference runs an EveryDirection spectral filter as two launches (horizontal, then
// vertical; Filters.jl:467-503).  k_apply_filter runs both in one launch but keeps the
// arithmetic of the launch boundary (compute_filter_result! followed by
// compute_filter_argument!), so results are identical while Q makes one HBM round trip.
// Sums run in the reference's order: accumulator from zero, n ascending, no contraction.
#pragma once
#include "cmdg_common.h"

namespace cmdg {

enum { TGT_INDICES = 0, TGT_ATMOS_PERT = 1, TGT_ATMOS_SPECIFIC = 2 };
constexpr int FILTER_MAXS = 32;
constexpr int ATMOS_NS = 5;  // rho, rho u[3], rho e

struct FilterArgs {
    double *Q;
    const double *aux;
    const double *vgeo;
    const double *Fh, *Fv;  // device (NQ, NQ) column-major
    int nstate, naux, nvgeo, nfs;
    int idx[FILTER_MAXS];  // 1-based
    int aux_rho, aux_rhoe;
    int do_h, do_v;
    int64_t nreal;
};

template <int NQ>
struct FDims {
    static constexpr int Np = NQ * NQ * NQ;
    static constexpr int NT = ((Np + 63) / 64) * 64;
    static constexpr int pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }
};

// compute_filter_argument! / compute_filter_result! of the atmos targets
template <int TARGET>
__device__ __forceinline__ void atmos_argument(double (&f)[ATMOS_NS], const double (&q)[ATMOS_NS],
                                               double ref_rho, double ref_rhoe)
{
    if constexpr (TARGET == TGT_ATMOS_PERT) {
#pragma unroll
        for (int s = 0; s < ATMOS_NS; ++s) f[s] = q[s];
        f[0] -= ref_rho;
        f[4] -= ref_rhoe;
    } else {
        const double rho_inv = 1 / q[0];
        const double rho_ref_inv = 1 / ref_rho;
#pragma unroll
        for (int s = 0; s < ATMOS_NS; ++s) f[s] = q[s] * rho_inv;
        f[4] -= ref_rhoe * rho_ref_inv;
    }
}
template <int TARGET>
__device__ __forceinline__ void atmos_result(double (&q)[ATMOS_NS], const double (&f)[ATMOS_NS],
                                             double ref_rho, double ref_rhoe)
{
    if constexpr (TARGET == TGT_ATMOS_PERT) {
#pragma unroll
        for (int s = 0; s < ATMOS_NS; ++s) q[s] = f[s];
        q[0] += ref_rho;
        q[4] += ref_rhoe;
    } else {
        const double rho = q[0];
        const double ratio = rho / ref_rho;
#pragma unroll
        for (int s = 0; s < ATMOS_NS; ++s) q[s] = f[s] * rho;
        q[4] += ref_rhoe * ratio;
    }
}

// one tensor pass over all (node, filtered state) items of the element: AXIS 0/1/2 = xi1/2/3
template <int NQ, int AXIS>
__device__ __forceinline__ void filter_pass(const double *__restrict__ sF,
                                            const double *__restrict__ in,
                                            double *__restrict__ out, int nfs, int tid)
{
    constexpr int Np = NQ * NQ * NQ, NT = FDims<NQ>::NT;
    constexpr int stride = AXIS == 0 ? 1 : (AXIS == 1 ? NQ : NQ * NQ);
    for (int w = tid; w < nfs * Np; w += NT) {
        const int ijk = w % Np;
        const int a = (ijk / stride) % NQ;
        const double *col = in + (w - a * stride);
        double acc = 0.0;
#pragma unroll
        for (int n = 0; n < NQ; ++n) acc += sF[a + NQ * n] * col[n * stride];
        out[w] = acc;
    }
}

// the passes of one launch of the reference kernel; returns the buffer holding the result
template <int NQ>
__device__ __forceinline__ double *filter_launch(bool horizontal, const double *sF, double *cur,
                                                 double *nxt, int nfs, int tid)
{
    if (horizontal) {
        filter_pass<NQ, 0>(sF, cur, nxt, nfs, tid);
        __syncthreads();
        filter_pass<NQ, 1>(sF, nxt, cur, nfs, tid);
        __syncthreads();
        return cur;
    }
    filter_pass<NQ, 2>(sF, cur, nxt, nfs, tid);
    __syncthreads();
    return nxt;
}

template <int NQ, int TARGET>
__global__ __launch_bounds__(FDims<NQ>::NT) void k_apply_filter(FilterArgs a)
{
    constexpr int Np = FDims<NQ>::Np, NT = FDims<NQ>::NT;
    extern __shared__ double lds[];
    __shared__ double sFh[NQ * NQ], sFv[NQ * NQ];
    const int tid = threadIdx.x;
    const int64_t e = xcd_remap(blockIdx.x, gridDim.x);
    const int nfs = TARGET == TGT_INDICES ? a.nfs : ATMOS_NS;
    double *cur = lds, *nxt = lds + nfs * Np;
    if (tid < NQ * NQ) {
        sFh[tid] = a.Fh[tid];
        sFv[tid] = a.Fv[tid];
    }
    double *Qe = a.Q + (int64_t)Np * a.nstate * e;
    [[maybe_unused]] double q[ATMOS_NS], ref_rho = 0, ref_rhoe = 0;
    if constexpr (TARGET == TGT_INDICES) {
        for (int w = tid; w < nfs * Np; w += NT) {
            const int fs = w / Np, ijk = w - fs * Np;
            cur[w] = Qe[ijk + Np * (a.idx[fs] - 1)];
        }
    } else {
        if (tid < Np) {
            const double *ae = a.aux + (int64_t)Np * a.naux * e;
#pragma unroll
            for (int s = 0; s < ATMOS_NS; ++s) q[s] = Qe[tid + Np * s];
            ref_rho = ae[tid + Np * a.aux_rho];
            ref_rhoe = ae[tid + Np * a.aux_rhoe];
            double f[ATMOS_NS];
            atmos_argument<TARGET>(f, q, ref_rho, ref_rhoe);
#pragma unroll
            for (int s = 0; s < ATMOS_NS; ++s) cur[tid + Np * s] = f[s];
        }
    }
    __syncthreads();
    if (a.do_h) {
        double *res = filter_launch<NQ>(true, sFh, cur, nxt, nfs, tid);
        if (a.do_v) {
            if constexpr (TARGET != TGT_INDICES) {
                // end of the horizontal launch + start of the vertical one
                if (tid < Np) {
                    double f[ATMOS_NS];
#pragma unroll
                    for (int s = 0; s < ATMOS_NS; ++s) f[s] = res[tid + Np * s];
                    atmos_result<TARGET>(q, f, ref_rho, ref_rhoe);
                    atmos_argument<TARGET>(f, q, ref_rho, ref_rhoe);
#pragma unroll
                    for (int s = 0; s < ATMOS_NS; ++s) res[tid + Np * s] = f[s];
                }
                __syncthreads();
            }
            double *other = res == cur ? nxt : cur;
            res = filter_launch<NQ>(false, sFv, res, other, nfs, tid);
        }
        cur = res;
    } else if (a.do_v) {
        cur = filter_launch<NQ>(false, sFv, cur, nxt, nfs, tid);
    }
    if constexpr (TARGET == TGT_INDICES) {
        for (int w = tid; w < nfs * Np; w += NT) {
            const int fs = w / Np, ijk = w - fs * Np;
            Qe[ijk + Np * (a.idx[fs] - 1)] = cur[w];
        }
    } else {
        if (tid < Np) {
            double f[ATMOS_NS];
#pragma unroll
            for (int s = 0; s < ATMOS_NS; ++s) f[s] = cur[tid + Np * s];
            atmos_result<TARGET>(q, f, ref_rho, ref_rhoe);
#pragma unroll
            for (int s = 0; s < ATMOS_NS; ++s) Qe[tid + Np * s] = q[s];
        }
    }
}

// Two vertical spectral FilterIndices filters on disjoint states in ONE launch (the ocean models'
// update_auxiliary_state!: cutoff filter on (u, v), exponential filter on theta,
// hydrostatic_boussinesq_model.jl:654-680): a state's result depends on its own column and its own
// filter matrix only, so the values are those of the two launches.  The first nfa filtered states
// of a.idx take Fv, the rest Fv2.
// One thread = one vertical line of one filtered state: its NQ values into registers, the NQ x NQ
// product in the reference's summation order, NQ stores -- no staging of the element, no barrier
// but the one behind the two matrices.  (The one-element-per-work-group form before it moved 6 KB per
// work-group behind two barriers: 85 us per launch on the 48 x 48 x 16 box where the bytes take 40.)
template <int NQ>
__global__ __launch_bounds__(256) void k_apply_vfilter_pair(FilterArgs a, const double *__restrict__ Fv2, int nfa)
{
    constexpr int Nij = NQ * NQ, Np = FDims<NQ>::Np;
    __shared__ double sFa[NQ * NQ], sFb[NQ * NQ];
    const int tid = threadIdx.x;
    if (tid < NQ * NQ) {
        sFa[tid] = a.Fv[tid];
        sFb[tid] = Fv2[tid];
    }
    __syncthreads();
    const int nfs = a.nfs;
    const int64_t line = (int64_t)blockIdx.x * blockDim.x + tid;  // (ij, filtered state, element)
    if (line >= (int64_t)Nij * nfs * a.nreal) return;
    const int64_t e = line / (Nij * nfs);
    const int r = (int)(line - e * (Nij * nfs)), fs = r / Nij, ij = r - fs * Nij;
    double *col = a.Q + ij + (int64_t)Np * ((a.idx[fs] - 1) + (int64_t)a.nstate * e);
    const double *sF = fs < nfa ? sFa : sFb;
    double v[NQ], o[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) v[n] = col[n * Nij];
#pragma unroll
    for (int kq = 0; kq < NQ; ++kq) {  // filter_pass<NQ, 2> with the state's own matrix
        double acc = 0.0;
#pragma unroll
        for (int n = 0; n < NQ; ++n) acc += sF[kq + NQ * n] * v[n];
        o[kq] = acc;
    }
#pragma unroll
    for (int kq = 0; kq < NQ; ++kq) col[kq * Nij] = o[kq];
}

// the reference's shared-memory tree (Filters.jl:848-861, 1047-1060): for n = 11..1, if
// nreduce >= 2^n, entry ijk (1-based) adds entry ijk + 2^(n-1) when that partner exists
template <int COUNT>
__device__ __forceinline__ void tree_reduce(double *v, int tid)
{
    constexpr int nreduce = FDims<1>::pow2(COUNT);
#pragma unroll
    for (int n = 11; n >= 1; --n) {
        if (nreduce >= (1 << n)) {
            const int h = 1 << (n - 1);
            if (tid + 1 <= h && tid + 1 + h <= COUNT) v[tid] += v[tid + h];
            __syncthreads();
        }
    }
}

// one launch = one direction (a.do_h xor a.do_v), as in the reference
template <int NQ, int TARGET>
__global__ __launch_bounds__(FDims<NQ>::NT) void k_apply_mp_filter(FilterArgs a)
{
    constexpr int Np = FDims<NQ>::Np, NT = FDims<NQ>::NT;
    extern __shared__ double lds[];
    __shared__ double sF[NQ * NQ], sM[NT], sB[NT], sA[NT];
    const int tid = threadIdx.x;
    const int64_t e = xcd_remap(blockIdx.x, gridDim.x);
    const int nfs = TARGET == TGT_INDICES ? a.nfs : ATMOS_NS;
    double *cur = lds, *nxt = lds + nfs * Np;
    if (tid < NQ * NQ) sF[tid] = a.do_h ? a.Fh[tid] : a.Fv[tid];
    double *Qe = a.Q + (int64_t)Np * a.nstate * e;
    [[maybe_unused]] double q0[ATMOS_NS], q[ATMOS_NS], ref_rho = 0, ref_rhoe = 0;
    double M = 0;
    if (tid < Np) M = a.vgeo[tid + Np * (VM + (int64_t)a.nvgeo * e)];
    if constexpr (TARGET == TGT_INDICES) {
        for (int w = tid; w < nfs * Np; w += NT) {
            const int fs = w / Np, ijk = w - fs * Np;
            cur[w] = Qe[ijk + Np * (a.idx[fs] - 1)];
        }
    } else {
        if (tid < Np) {
            const double *ae = a.aux + (int64_t)Np * a.naux * e;
#pragma unroll
            for (int s = 0; s < ATMOS_NS; ++s) q0[s] = q[s] = Qe[tid + Np * s];
            ref_rho = ae[tid + Np * a.aux_rho];
            ref_rhoe = ae[tid + Np * a.aux_rhoe];
            double f[ATMOS_NS];
            atmos_argument<TARGET>(f, q, ref_rho, ref_rhoe);
#pragma unroll
            for (int s = 0; s < ATMOS_NS; ++s) cur[tid + Np * s] = f[s];
        }
    }
    __syncthreads();
    cur = filter_launch<NQ>(a.do_h != 0, sF, cur, nxt, nfs, tid);
    if constexpr (TARGET != TGT_INDICES) {
        if (tid < Np) {
            double f[ATMOS_NS];
#pragma unroll
            for (int s = 0; s < ATMOS_NS; ++s) f[s] = cur[tid + Np * s];
            atmos_result<TARGET>(q, f, ref_rho, ref_rhoe);
        }
    }
    if (tid < Np) sM[tid] = M;
    __syncthreads();
    tree_reduce<Np>(sM, tid);
    const double Minv = 1 / sM[0];
    // every state is rewritten as p_Q + M^-1 (sum(M Q_before) - sum(M Q_after))
    for (int s = 0; s < a.nstate; ++s) {
        int fs = -1;
        if constexpr (TARGET == TGT_INDICES) {
            for (int k = 0; k < nfs; ++k)
                if (a.idx[k] - 1 == s) fs = k;  // last writer wins, as compute_filter_result!
        } else {
            fs = s < ATMOS_NS ? s : -1;
        }
        double before = 0, after = 0;
        if (tid < Np) {
            if constexpr (TARGET == TGT_INDICES) {
                before = Qe[tid + Np * s];
                after = fs >= 0 ? cur[tid + Np * fs] : before;
            } else {
                before = fs >= 0 ? q0[fs] : Qe[tid + Np * s];
                after = fs >= 0 ? q[fs] : before;
            }
            sB[tid] = M * before;
            sA[tid] = M * after;
        }
        __syncthreads();
        tree_reduce<Np>(sB, tid);
        tree_reduce<Np>(sA, tid);
        if (tid < Np) Qe[tid + Np * s] = after + Minv * (sB[0] - sA[0]);
        __syncthreads();
    }
}

// block = 64 threads, one element; threads (i,j) < NQ*NQ own a pencil along k
template <int NQ>
__global__ __launch_bounds__(64) void k_apply_tmar_filter(FilterArgs a)
{
    constexpr int Np = NQ * NQ * NQ, Nij = NQ * NQ;
    static_assert(Nij <= 64, "TMAR kernel: one wavefront per element");
    __shared__ double sMJQ[64], sMJQc[64];
    const int tid = threadIdx.x;
    const int64_t e = xcd_remap(blockIdx.x, gridDim.x);
    double *Qe = a.Q + (int64_t)Np * a.nstate * e;
    double MJ[NQ], lQ[NQ];
    if (tid < Nij) {
#pragma unroll
        for (int k = 0; k < NQ; ++k) MJ[k] = a.vgeo[tid + Nij * k + Np * (VM + (int64_t)a.nvgeo * e)];
    }
    for (int sf = 0; sf < a.nfs; ++sf) {
        double *qs = Qe + Np * (a.idx[sf] - 1);
        if (tid < Nij) {
            double MJQ = 0, MJQc = 0;
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const double Qs = qs[tid + Nij * k];
                lQ[k] = Qs;
                const double Qc = Qs >= 0 ? Qs : 0.0;
                MJQ += MJ[k] * Qs;
                MJQc += MJ[k] * Qc;
            }
            sMJQ[tid] = MJQ;
            sMJQc[tid] = MJQc;
        }
        __syncthreads();
        tree_reduce<Nij>(sMJQ, tid);
        tree_reduce<Nij>(sMJQc, tid);
        const double avg = sMJQ[0], cavg = sMJQc[0];
        const double r = avg > 0 ? avg / cavg : 0.0;
        if (tid < Nij) {
#pragma unroll
            for (int k = 0; k < NQ; ++k) qs[tid + Nij * k] = lQ[k] >= 0 ? r * lQ[k] : 0.0;
        }
        __syncthreads();
    }
}

// update!  LowStorageRungeKuttaMethod.jl:146-158 (used when a tendency filter sits between
// the right-hand side and the update, so the update cannot be fused into k_tendency)
static __global__ void k_lsrk_update(double *__restrict__ dQ, double *__restrict__ Q, double rka,
                                     double rkb_dt, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        Q[i] += rkb_dt * dQ[i];
        dQ[i] *= rka;
    }
}

}  // namespace cmdg
