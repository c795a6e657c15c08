// Shared device-side definitions for libcmdg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmdg {

// vgeo column ids, 0-based (reference ids Grids.jl:76-92 minus one)
enum { XI1X1 = 0, XI2X1, XI3X1, XI1X2, XI2X2, XI3X2, XI1X3, XI2X3, XI3X3, VM, VMI };
// sgeo row ids (Grids.jl:129-130)
enum { SN1 = 0, SN2, SN3, SSM, SVMI };
enum { DIR_EVERY = 0, DIR_HORIZONTAL = 1, DIR_VERTICAL = 2 };
enum {
    NF_RUSANOV = 0, NF_CENTRAL = 1, NF_ROE = 2, NF_HLLC = 3, NF_LMARS = 4,
    // RoeNumericalFluxMoist(LM, HH, LV, LVPP) of the moist AtmosModel (AtmosModel.jl:1276-1513):
    // plain, low-Mach, Harten-Hyman, LeVeque, positivity-preserving LeVeque
    NF_ROE_MOIST = 5, NF_ROE_MOIST_LM = 6, NF_ROE_MOIST_HH = 7, NF_ROE_MOIST_LV = 8,
    NF_ROE_MOIST_LVPP = 9
};
enum { BS_FIRST = 0, BS_GRADIENT = 1 };

constexpr int NXCD = 8;

struct GridDev {
    const double *vgeo, *sgeo;
    const int64_t *vmapM, *vmapP, *elemtobndy;
    const double *D;   // device, (Nq, Nq) column-major, horizontal
    const double *Dv;  // device, (Nqv, Nqv) column-major, vertical (== D for one order)
    int nvgeo;
    // Face tables digested once at cmdg_create from the reference's (sgeo, vmap-, vmap+,
    // elemtobndy), real elements only, indexed by the face task t = 0 .. NFT-1 of an element:
    //   faceP[e][t]     int32, 0-based global node id of the plus side (eP * Np + vidP; the
    //                   minus node itself on boundary faces, DGModel_kernels.jl:686-692)
    //   faceG[e][c][t]  n1, n2, n3, sM (c = 0..3), one coalesced load per component
    // vmap- is a pure function of (f, n) (checked at create) and vMI is MI of that node
    // (Grids.jl:1097-1101, checked at create): neither is stored.  36 B per face node
    // instead of the 56 B of the reference tables.
    const int32_t *faceP;
    const double *faceG;
};

// Ghost exchange without pack / unpack launches (handles with neighbours; every pointer is NULL
// otherwise).  The wire format is the reference's (nstate, nvmap) state-fastest buffer of
// kernel_fillsendbuf! / kernel_transferrecvbuf! (MPIStateArrays.jl:837-871).
//   sender:   the exterior launch of a pass copies the element's nodes of vmapsend out of LDS into
//             the send buffer of every array it produces (the nodes of vmapsend all belong to
//             exterior elements, Topologies.jl:250-251);
//   receiver: the plus side of a face whose neighbour is a ghost element is read from the receive
//             buffer of the array instead of from the ghost element.
struct SendEnt {
    int32_t node;  // node of the element, 0-based
    int32_t pos;   // position in vmapsend, 0-based
};
struct HaloDev {
    const int32_t *sendoff;    // (nreal + 1) CSR over the real elements; NULL: this launch sends nothing
    const SendEnt *sendent;    // ascending position within an element
    double *send[2];           // send buffers of this pass's outputs (NULL: not exchanged)
    const int32_t *ghostslot;  // (nghost * Np) position in vmaprecv of a ghost node; NULL: the ghost
                               // elements hold the data (exchanges are unpacked)
    const double *recvQ, *recvGF, *recvHG, *recvHD;
    int64_t nreal;
};

// NQ = horizontal points per direction, NQV = vertical ones (polynomialorder = (N_h, N_v);
// reference: `info.Nq`, `info.Nqk`, `Nfp_h`, `Nfp_v` of basic_grid_info, SpaceDiscretization.jl:18-60)
template <int NQ, int NQV = NQ>
struct KDims {
    static constexpr int Nij = NQ * NQ;
    static constexpr int Np = NQ * NQ * NQV;
    static constexpr int Nfph = NQ * NQV;  // faces 1..4
    static constexpr int Nfpv = NQ * NQ;   // faces 5, 6
    // stride of the face tables: sgeo / vmap are (Nfp_max, nface, nelem) (Metrics.jl:485-488)
    static constexpr int Nfp = Nfph > Nfpv ? Nfph : Nfpv;
    static constexpr int NFT = 4 * Nfph + 2 * Nfpv;  // face-node tasks per element
    static constexpr int NT = (((Np > NFT ? Np : NFT) + 63) / 64) * 64;
    __device__ __forceinline__ static void face_task(int t, int &f, int &n)
    {
        if constexpr (NQ == NQV) {
            f = t / Nfp;
            n = t % Nfp;
        } else if (t < 4 * Nfph) {
            f = t / Nfph;
            n = t % Nfph;
        } else {
            f = 4 + (t - 4 * Nfph) / Nfpv;
            n = (t - 4 * Nfph) % Nfpv;
        }
    }
};

// Blocks b and b+8 share an XCD (and its L2); hand each XCD a contiguous run of the
// element list so face neighbours (adjacent in the stacked/Hilbert order) hit one L2.
__device__ __forceinline__ int64_t xcd_remap(int64_t b, int64_t G)
{
    const int64_t q = G / NXCD, r = G % NXCD;
    const int64_t x = b % NXCD, y = b / NXCD;
    return x * q + (x < r ? x : r) + y;
}

template <int N>
__device__ __forceinline__ void fill_negzero(double (&a)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = -0.0;
}
// zero-length-safe local array (C++ forbids T[0])
template <int N>
struct Vec {
    double v[N > 0 ? N : 1];
    __device__ __forceinline__ double &operator[](int i) { return v[i]; }
    __device__ __forceinline__ const double &operator[](int i) const { return v[i]; }
    __device__ __forceinline__ operator double *() { return v; }
    __device__ __forceinline__ operator const double *() const { return v; }
    __device__ __forceinline__ void negzero()
    {
#pragma unroll
        for (int i = 0; i < (N > 0 ? N : 1); ++i) v[i] = -0.0;
    }
};

template <int NVAR, int Np>
__device__ __forceinline__ void load_state(Vec<NVAR> &dst, const double *__restrict__ arr, int ijk,
                                           int64_t e)
{
#pragma unroll
    for (int s = 0; s < NVAR; ++s) dst[s] = arr[ijk + (int64_t)Np * (s + (int64_t)NVAR * e)];
}

// position in vmaprecv of the plus-side node (eP, vidP), -1 when it is read from its element
template <int Np>
__device__ __forceinline__ int ghost_slot(const HaloDev &h, int64_t eP, int vidP)
{
    return (h.ghostslot != nullptr && eP >= h.nreal) ? h.ghostslot[(eP - h.nreal) * Np + vidP] : -1;
}
// the first NLOAD of the NVAR columns of the plus-side node: from its element, or from the
// receive buffer of the array (gslot >= 0), whose positions hold NVR columns
template <int NVAR, int Np, int NLOAD, int NVR = NVAR>
__device__ __forceinline__ void load_plus(Vec<NLOAD> &dst, const double *__restrict__ arr,
                                          const double *__restrict__ recv, int gslot, int vidP,
                                          int64_t eP)
{
    if (gslot >= 0) {
#pragma unroll
        for (int s = 0; s < NLOAD; ++s) dst[s] = recv[s + (int64_t)NVR * gslot];
    } else {
#pragma unroll
        for (int s = 0; s < NLOAD; ++s) dst[s] = arr[vidP + (int64_t)Np * (s + (int64_t)NVAR * eP)];
    }
}
// Qhypervisc_grad inside the library (round 4): node-major within an element, (NHG, Np, nelem)
// instead of the reference's (Np, NHG, nelem).  A plus-side gather of a face node then reads NHG * 8
// contiguous bytes instead of NHG values at a stride of Np * 8: on a xi1 / xi2 face of an N = 4
// element the column-major form touches every 128-byte line of all NHG columns of the neighbour
// (5 x the bytes used), the node-major form 1.3-2.3 x.  The volume reads are one contiguous
// record per thread, the stores go out of LDS in memory order.  A caller that hands
// Qhypervisc_grad to cmdg_create gets the reference layout copied out after every evaluation
// (k_export_hg); CMDG_HG_NODE_MAJOR=0 builds the reference layout in place (the A/B).
#ifndef CMDG_HG_NODE_MAJOR
#define CMDG_HG_NODE_MAJOR 1
#endif
template <bool NODE_MAJOR, int NCOL, int Np>
__device__ __forceinline__ int64_t col_at(int n, int s, int64_t e)
{
    return NODE_MAJOR ? s + (int64_t)NCOL * (n + (int64_t)Np * e) : n + (int64_t)Np * (s + (int64_t)NCOL * e);
}
template <int NHG, int Np>
__device__ __forceinline__ int64_t hg_at(int n, int s, int64_t e)
{
    return col_at<CMDG_HG_NODE_MAJOR != 0, NHG, Np>(n, s, e);
}
template <int NHG, int Np, int NLOAD>
__device__ __forceinline__ void load_plus_hg(Vec<NLOAD> &dst, const double *__restrict__ arr,
                                             const double *__restrict__ recv, int gslot, int vidP, int64_t eP)
{
    if (gslot >= 0) {
#pragma unroll
        for (int s = 0; s < NLOAD; ++s) dst[s] = recv[s + (int64_t)NHG * gslot];
    } else {
#pragma unroll
        for (int s = 0; s < NLOAD; ++s) dst[s] = arr[hg_at<NHG, Np>(vidP, s, eP)];
    }
}
// the first NW of the NHG columns of element e of a node-major array out of LDS (src[s * Np + n])
// in memory order
template <int NHG, int Np, int NW>
__device__ __forceinline__ void store_node_major(double *__restrict__ arr, int64_t e, int tid, int nthreads,
                                                 const double *src)
{
    double *dst = arr + (int64_t)NHG * Np * e;
    for (int m = tid; m < NHG * Np; m += nthreads) {
        const int n = m / NHG, s = m - n * NHG;
        if (NW == NHG || s < NW) dst[m] = src[s * Np + n];
    }
}
// exterior launches: the element's nodes of vmapsend into send buffer `which` (NVAR columns per
// position, the first NW written); value(s, node) is the value the pass stores for the node
template <int NVAR, int NW, class F>
__device__ __forceinline__ void send_nodes(const HaloDev &h, int which, int64_t e, int tid,
                                           int nthreads, F value)
{
    if (h.sendoff == nullptr || h.send[which] == nullptr) return;
    const int b1 = h.sendoff[e + 1];
    for (int k = h.sendoff[e] + tid; k < b1; k += nthreads) {
        const SendEnt ent = h.sendent[k];
        double *dst = h.send[which] + (int64_t)NVAR * ent.pos;
#pragma unroll
        for (int s = 0; s < NW; ++s) dst[s] = value(s, ent.node);
    }
}

}  // namespace cmdg
