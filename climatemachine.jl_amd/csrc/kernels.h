// Hand-written gfx950 kernels of the DG hot path.  One workgroup per element (the tendency pass
// of elements above N = 4 takes two, see TendencyShape):
//   phase 1  thread = volume node: pointwise physics, contravariant fluxes / gradient
//            arguments staged in LDS;
//   phase 2  thread = volume node: (N+1)-point contractions with D out of LDS, result
//            into an LDS accumulator;
//   phase 3  thread = face node (6 faces x Nfp tasks at once): numerical / boundary
//            fluxes, lifted into the LDS accumulator one opposite-face pair at a time
//            (race free, reference order);
//   phase 4  thread = volume node: one coalesced store per output field (for the
//            tendency kernel with the LSRK update fused in).
// Each kernel replaces a *group* of reference kernels (horizontal + vertical volume
// kernel and the per-direction interface launches) of
// src/Numerics/DGMethods/DGModel_kernels.jl; the accumulation order of the reference
// is kept term by term (see DESIGN.md "summation order").
#pragma once
#include <type_traits>

#include "cmdg_common.h"

// minimum waves per SIMD requested from the register allocator (tuning knobs)
// Memory round trips of the tendency pass issued as early as they can be (bits: 1 = the old tendency
// of all states in one batch instead of one dependent load per state inside the contraction; 2 =
// the metric rows with the state, before the flux arithmetic, where that does not cost residency).
// The pass is bound by its chain of dependent loads, not by bytes (profiles/r04_ab_tendency_hoist.txt;
// pinning every plus-side gather of a face node before the first-order flux was the third candidate
// and lost).
#ifndef CMDG_TEND_HOIST
#define CMDG_TEND_HOIST 3
#endif
#ifndef CMDG_TEND_MINW
#define CMDG_TEND_MINW 1
#endif
// elements per work-group of k_tendency for N >= 5 (see TendencyShape)
#ifndef CMDG_TEND_EPB_LARGE
#define CMDG_TEND_EPB_LARGE 2
#endif
// ... and for N <= 4 (one: 192 threads, minus side of the faces staged in LDS; two measured
// slower on Held-Suarez and on the rising bubble, profiles/r03_ab_two_elements_n4.txt)
#ifndef CMDG_TEND_EPB_SMALL
#define CMDG_TEND_EPB_SMALL 1
#endif
// k_gradients: the Held-Suarez instantiation needs 130 VGPRs unconstrained (3 waves/SIMD);
// asking for 4 gives 128 without scratch and 9 % less time per launch (profiles/r01_ab_*.txt)
#ifndef CMDG_GRAD_MINW
#define CMDG_GRAD_MINW 4
#endif
#ifndef CMDG_GRAD_BOUND_LARGE
#define CMDG_GRAD_BOUND_LARGE 1024
#endif
#ifndef CMDG_LAP_MINW
#define CMDG_LAP_MINW 1
#endif
// paired work-groups of the tendency pass (TendencyShape<..., PAIR>, CMDG_OPT_TENDENCY_PAIRS): the
// round-4 structural experiment, measured and rejected (profiles/r04_ab_tendency_pairs.txt), kept
// buildable: make EXTRA=-DCMDG_TEND_PAIRS=1
#ifndef CMDG_TEND_PAIRS
#define CMDG_TEND_PAIRS 0
#endif
// four-wave work-groups for the tendency pass of large elements (k_tendency_big,
// CMDG_OPT_TENDENCY_FOUR_WAVES): measured and rejected in round 4 (BOMEX k_tendency 596 -> 869 us,
// profiles/r04_ab_tendency_four_waves.txt), kept buildable: make EXTRA=-DCMDG_TEND_FOUR_WAVES=1
#ifndef CMDG_TEND_FOUR_WAVES
#define CMDG_TEND_FOUR_WAVES 0
#endif
// tendency pass of large elements (N >= 5) in two launches, volume then interface + update
// (see TendencyShape::SPLIT; off: measured slower than two elements per work-group), and the
// register budgets asked for the two halves (waves per SIMD)
#ifndef CMDG_TEND_SPLIT_LARGE
#define CMDG_TEND_SPLIT_LARGE 0
#endif
#ifndef CMDG_TENDV_MINW
#define CMDG_TENDV_MINW 4
#endif
#ifndef CMDG_TENDF_MINW
#define CMDG_TENDF_MINW 4
#endif

namespace cmdg {

// ---------------------------------------------------------------------------------
template <class P>
struct PassArgs {
    typename P::Params prm;
    GridDev g;
    const int64_t *elems;  // 1-based element list (interior or exterior)
    int64_t nelems;
    // state arrays
    const double *Q;
    const double *aux;
    double *aux_rw;         // same array, for the fused auxiliary refresh
    const double *derived;  // handle-owned time-invariant per-node fields of the law (P::NDER)
    double *gf;        // state_gradient_flux (written by gradients, read by tendency)
    double *hypgrad;   // Qhypervisc_grad
    double *hypdiv;    // Qhypervisc_div
    double *tendency;  // tendency (== dQ when the LSRK update is fused)
    double *Qout;      // LSRK: updated state
    double t, alpha, beta;
    const double *tptr;  // time of the evaluation in device memory (a captured step is replayed
                         // with the time a one-thread kernel of the graph advances); NULL: t
    double rkb_dt, rka_next;
    int direction;     // direction of this pass (dg.direction or dg.diffusion_direction)
    int model_dir;     // dg.direction handed to the pointwise fluxes
    int nf_first;
    HaloDev h;         // ghost exchange without pack / unpack launches (cmdg_common.h)
};

struct FacePt {
    double n[3], sM, vMI;
    int64_t eP;
    int vidM, vidP, bctag;
};

// Volume node of face node n of face f: faces 1..6 = xi1-, xi1+, xi2-, xi2+, xi3-, xi3+, the
// remaining two indices in order, lower axis fastest (Grids.jl:586-594).  This is vmap- minus
// the element offset; cmdg_create checks the caller's vmap- against it.
template <int NQ, int NQV = NQ>
__host__ __device__ __forceinline__ constexpr int face_vid(int f, int n)
{
    const int a = n % NQ, b = n / NQ;
    switch (f) {
    case 0: return NQ * (a + NQ * b);
    case 1: return (NQ - 1) + NQ * (a + NQ * b);
    case 2: return a + NQ * NQ * b;
    case 3: return a + NQ * ((NQ - 1) + NQ * b);
    case 4: return n;
    default: return n + NQ * NQ * (NQV - 1);
    }
}

// index half of face_setup: issued at kernel start so that the plus-side gathers do not wait
// for a dependent table load when the interface phase begins
template <int NQ, int NQV = NQ>
__device__ __forceinline__ void face_index(const GridDev &g, int64_t e, int t, int f, int32_t &idP,
                                           int &bctag)
{
    idP = g.faceP[(int64_t)KDims<NQ, NQV>::NFT * e + t];
    bctag = (int)g.elemtobndy[f + 6 * e];
}
template <int NQ, int NQV = NQ>
__device__ __forceinline__ void face_geometry(const GridDev &g, int64_t e, int t, int f, int n,
                                              int32_t idP, int bctag, FacePt &fp)
{
    constexpr int Np = KDims<NQ, NQV>::Np, NFT = KDims<NQ, NQV>::NFT;
    const double *sg = g.faceG + (int64_t)4 * NFT * e + t;
    fp.n[0] = sg[0];
    fp.n[1] = sg[NFT];
    fp.n[2] = sg[2 * NFT];
    fp.sM = sg[3 * NFT];
    fp.vidM = face_vid<NQ, NQV>(f, n);
    fp.vMI = g.vgeo[fp.vidM + (int64_t)Np * (VMI + (int64_t)g.nvgeo * e)];
    fp.bctag = bctag;
    fp.eP = idP / Np;
    fp.vidP = idP - (int)fp.eP * Np;
}
template <int NQ, int NQV = NQ>
__device__ __forceinline__ void face_setup(const GridDev &g, int64_t e, int t, int f, int n,
                                           FacePt &fp)
{
    int32_t idP;
    int bctag;
    face_index<NQ, NQV>(g, e, t, f, idP, bctag);
    face_geometry<NQ, NQV>(g, e, t, f, n, idP, bctag, fp);
}

// One-time digest of the reference face tables (see GridDev); bad[0] collects what does not
// hold: bit 0 vmap- is not the canonical face numbering, bit 1 sgeo's vMI is not vgeo's MI at
// the face node, bit 2 a plus-side id does not fit 32 bits.
static __global__ void k_face_digest(const double *__restrict__ vgeo, int nvgeo,
                                     const double *__restrict__ sgeo,
                                     const int64_t *__restrict__ vmapM,
                                     const int64_t *__restrict__ vmapP,
                                     const int64_t *__restrict__ elemtobndy, int NQ, int NQV,
                                     int64_t nreal, int32_t *__restrict__ faceP,
                                     double *__restrict__ faceG, int *__restrict__ bad)
{
    const int Np = NQ * NQ * NQV, Nfph = NQ * NQV, Nfpv = NQ * NQ;
    const int Nfp = Nfph > Nfpv ? Nfph : Nfpv, NFT = 4 * Nfph + 2 * Nfpv;
    const int64_t I = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= nreal * NFT) return;
    const int64_t e = I / NFT;
    const int t = (int)(I % NFT);
    int f, n;
    if (t < 4 * Nfph) {
        f = t / Nfph;
        n = t % Nfph;
    } else {
        f = 4 + (t - 4 * Nfph) / Nfpv;
        n = (t - 4 * Nfph) % Nfpv;
    }
    const int64_t o = n + (int64_t)Nfp * (f + 6 * e);
    const int a = n % NQ, b = n / NQ;
    int vid;
    switch (f) {
    case 0: vid = NQ * (a + NQ * b); break;
    case 1: vid = (NQ - 1) + NQ * (a + NQ * b); break;
    case 2: vid = a + NQ * NQ * b; break;
    case 3: vid = a + NQ * ((NQ - 1) + NQ * b); break;
    case 4: vid = n; break;
    default: vid = n + NQ * NQ * (NQV - 1); break;
    }
    int flags = 0;
    const int64_t idM = vmapM[o] - 1;
    if (idM != e * Np + vid) flags |= 1;
    int64_t idP = vmapP[o] - 1;
    if (elemtobndy[f + 6 * e] != 0) idP = e * Np + vid;  // DGModel_kernels.jl:686-692
    if (idP < 0 || idP > 2147483647LL) flags |= 4;
    faceP[I] = (int32_t)idP;
    const double *sg = sgeo + 5 * o;
#pragma unroll
    for (int c = 0; c < 4; ++c) faceG[((int64_t)4 * e + c) * NFT + t] = sg[c];
    const double mi = vgeo[vid + (int64_t)Np * (VMI + (int64_t)nvgeo * e)];
    if (!(sg[SVMI] == mi)) flags |= 2;
    if (flags) atomicOr(bad, flags);
}

// A law may offer flux_first_order and wavespeed of one state in one call (both need the same
// thermodynamic state; the moist law's costs a saturation adjustment): detected by the member
// HAS_FLUX_WAVESPEED, absent from the other laws, whose code is unchanged.
template <class P, class = void>
struct has_flux_wavespeed : std::false_type {
};
template <class P>
struct has_flux_wavespeed<P, std::void_t<decltype(P::HAS_FLUX_WAVESPEED)>> : std::true_type {
};

// Laws whose state_gradient_flux the library keeps node-major, (NGF, Np, nelem), like
// Qhypervisc_grad (cmdg_common.h): `static constexpr bool GF_NODE_MAJOR = true` in the functor.
// The dry atmosphere takes it (10 columns gathered on the plus side of every face node; the moist
// law was measured and keeps the reference layout, physics_moist.h); laws whose hooks work on the
// array in the reference layout do not.
#ifndef CMDG_GF_NODE_MAJOR
#define CMDG_GF_NODE_MAJOR 1
#endif
template <class P, class = void>
struct gf_node_major : std::false_type {
};
template <class P>
struct gf_node_major<P, std::void_t<decltype(P::GF_NODE_MAJOR)>>
    : std::integral_constant<bool, P::GF_NODE_MAJOR && CMDG_GF_NODE_MAJOR != 0 && (P::NGF > 0)> {
};
template <class P, int Np>
__device__ __forceinline__ int64_t gf_at(int n, int s, int64_t e)
{
    return col_at<gf_node_major<P>::value, P::NGF, Np>(n, s, e);
}
template <class P, int Np>
__device__ __forceinline__ void load_gf(Vec<P::NGF> &dst, const double *__restrict__ arr, int n, int64_t e)
{
#pragma unroll
    for (int s = 0; s < P::NGF; ++s) dst[s] = arr[gf_at<P, Np>(n, s, e)];
}
template <class P, int Np>
__device__ __forceinline__ void load_plus_gf(Vec<P::NGF> &dst, const double *__restrict__ arr,
                                             const double *__restrict__ recv, int gslot, int vidP, int64_t eP)
{
    if (gslot >= 0) {
#pragma unroll
        for (int s = 0; s < P::NGF; ++s) dst[s] = recv[s + (int64_t)P::NGF * gslot];
    } else {
        load_gf<P, Np>(dst, arr, vidP, eP);
    }
}

// A law may also keep a few derived values per node (P::NCACHE doubles, P::node_cache) that its
// first-order flux and wave speed are computed from: k_tendency then evaluates them once per
// volume node and hands the staged copy to the minus side of the faces.
template <class P, class = void>
struct node_cache_size : std::integral_constant<int, 0> {
};
template <class P>
struct node_cache_size<P, std::void_t<decltype(P::NCACHE)>> : std::integral_constant<int, P::NCACHE> {
};

// byte accounting: columns of Q / of the auxiliary state the volume code of a pass reads
// (P::state_read, P::aux_read; every column for a law that does not say)
template <class P, class = void>
struct law_reads {
    static constexpr int state(int) { return P::NS; }
    static constexpr int aux(int) { return P::NAUX; }
};
template <class P>
struct law_reads<P, std::void_t<decltype(P::aux_read(0))>> {
    static constexpr int state(int pass) { return P::state_read(pass); }
    static constexpr int aux(int pass) { return P::aux_read(pass); }
};

// laws that carry numerical_flux_first_order! methods of their own (P::LAW_NF)
template <class P, class = void>
struct has_law_nf : std::false_type {
};
template <class P>
struct has_law_nf<P, std::void_t<decltype(P::LAW_NF)>> : std::integral_constant<bool, P::LAW_NF> {
};

// numerical_flux_first_order!  NumericalFluxes.jl:223-285 (Rusanov) / :300-340 (central)
template <class P>
__device__ __forceinline__ void nf_first_order(const typename P::Params &prm, int nf,
                                               Vec<P::NS> &fluxn, const double *n,
                                               const double *QM, const double *auxM,
                                               const double *QP, const double *auxP, double t,
                                               int facedir, const double *cacheM = nullptr)
{
    constexpr int NS = P::NS;
    if constexpr (has_law_nf<P>::value) {
        if (nf >= NF_ROE) {
            P::numerical_flux_law(prm, nf, fluxn, n, QM, auxM, QP, auxP, t, facedir);
            return;
        }
    }
    Vec<3 * NS> FM, FP;
    Vec<NS> wM, wP;
    FM.negzero();
    FP.negzero();
    if constexpr (has_flux_wavespeed<P>::value) {
        if (nf == NF_RUSANOV) {
            if constexpr (node_cache_size<P>::value > 0) {
                if (cacheM)
                    P::flux_wavespeed_cached(prm, FM, wM, n, QM, auxM, cacheM);
                else
                    P::flux_wavespeed(prm, FM, wM, n, QM, auxM, t, facedir);
            } else {
                P::flux_wavespeed(prm, FM, wM, n, QM, auxM, t, facedir);
            }
            P::flux_wavespeed(prm, FP, wP, n, QP, auxP, t, facedir);
        } else {
            P::flux_first_order(prm, FM, QM, auxM, t, facedir);
            P::flux_first_order(prm, FP, QP, auxP, t, facedir);
        }
    } else {
        P::flux_first_order(prm, FM, QM, auxM, t, facedir);
        P::flux_first_order(prm, FP, QP, auxP, t, facedir);
    }
    const double nh0 = n[0] / 2, nh1 = n[1] / 2, nh2 = n[2] / 2;
#pragma unroll
    for (int s = 0; s < NS; ++s)
        fluxn[s] += (FM[3 * s] + FP[3 * s]) * nh0 + (FM[3 * s + 1] + FP[3 * s + 1]) * nh1 +
                    (FM[3 * s + 2] + FP[3 * s + 2]) * nh2;
    if (nf == NF_RUSANOV) {
        if constexpr (!has_flux_wavespeed<P>::value) {
            P::wavespeed(prm, wM, n, QM, auxM, t, facedir);
            P::wavespeed(prm, wP, n, QP, auxP, t, facedir);
        }
        Vec<NS> pen;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const double mw = wM[s] > wP[s] ? wM[s] : wP[s];
            pen[s] = mw * (QM[s] - QP[s]);
        }
        if constexpr (P::HAS_PENALTY) P::update_penalty(prm, pen, n, QM, QP);  // (:266-279)
#pragma unroll
        for (int s = 0; s < NS; ++s) fluxn[s] += pen[s] / 2;
    }
}

// ---------------------------------------------------------------------------------
// Index of a surface node among the element's surface nodes (-1: interior node).  The
// minus-side face data of the interface phases is staged in LDS for surface nodes only.
template <int NQ, int NQV = NQ>
__device__ __forceinline__ int surf_index(int ijk)
{
    constexpr int NI = NQ - 2, NIV = NQV > 2 ? NQV - 2 : 0;
    const int i = ijk % NQ, j = (ijk / NQ) % NQ, k = ijk / (NQ * NQ);
    const bool ii = i >= 1 && i <= NI, jj = j >= 1 && j <= NI, kk = k >= 1 && k <= NIV;
    if (ii && jj && kk) return -1;
    auto clampi = [](int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); };
    int before = NI * NI * clampi(k - 1, NIV);
    if (kk) {
        before += NI * clampi(j - 1, NI);
        if (jj) before += clampi(i - 1, NI);
    }
    return ijk - before;
}
template <int NQ, int NQV = NQ>
struct SurfDims {
    static constexpr int NSURF =
        NQ * NQ * NQV - (NQ - 2) * (NQ - 2) * (NQV > 2 ? NQV - 2 : 0);
};

// compute_gradient_flux! of a node whose gradient ARGUMENT is at hand (the volume node's own, the
// minus side of a face): a law whose gradient flux needs a quantity that is also an entry of its
// argument (the dry atmosphere's theta_v under SmagorinskyLilly, a pow) takes it from there.
template <class P, class = void>
struct has_gradient_flux_g : std::false_type {
};
template <class P>
struct has_gradient_flux_g<P, std::void_t<decltype(&P::gradient_flux_g)>> : std::true_type {
};
template <class P>
__device__ __forceinline__ void law_gradient_flux(const typename P::Params &prm, double *gf,
                                                  const double *g, const double *Q, const double *aux,
                                                  double t, const double *G)
{
    if constexpr (has_gradient_flux_g<P>::value)
        P::gradient_flux_g(prm, gf, g, Q, aux, t, G);
    else
        P::gradient_flux(prm, gf, g, Q, aux, t);
}

// A law may ask for its own register budget in the gradient pass (waves per SIMD the allocator
// is to aim for): light laws gain residency, heavy ones keep the default.
template <class P, class = void>
struct grad_min_waves : std::integral_constant<int, CMDG_GRAD_MINW> {
};
template <class P>
struct grad_min_waves<P, std::void_t<decltype(P::GRAD_MIN_WAVES)>> : std::integral_constant<int, P::GRAD_MIN_WAVES> {
};

// ---------------------------------------------------------------------------------
// Launch shape of the tendency pass.  N <= 4: one element per work-group, 192 threads.  Larger
// elements: 343 nodes are six waves, which the four SIMDs of a CU take as 2-1-2-1 -- two of them
// carry twice the work and a second six-wave work-group does not become resident above 128
// VGPRs (profiles/r02_lds_occupancy.jsonl).  Two elements per work-group (686 threads, eleven
// waves, 3-3-3-2) even that out; the minus side of the faces is then read from memory instead
// of being staged, so that both elements' flux buffers fit the LDS of a CU.
//
// SPLIT (large elements): the pass in two launches with separate register budgets, as the
// reference has them (volume_tendency! / dgsem_interface_tendency!, DGModel_kernels.jl:64-548 /
// :588-901) -- k_tendency<..., TEND_VOLUME> forms the volume part and stores it, k_tendency<...,
// TEND_FACES> reads it back, adds the lifted face fluxes in the same order and applies the fused
// update.  Measured on BOMEX (N = 6, 8 192 elements, profiles/r03_ab_bomex_split.txt): the two
// launches take 727 us against 603 us for the fused two-element work-group -- the interface half
// still needs 167-181 VGPRs and 73 KB of LDS (two six-wave work-groups per CU at best, no more
// waves than the fused kernel's eleven), and the round trip of the tendency comes on top.  Kept
// behind CMDG_TEND_SPLIT_LARGE as the recorded loser.
//
// PAIR (round 4, the structural experiment): the two elements of a work-group are horizontal
// neighbours across a xi1 face -- element 0's face 2 (xi1+) is element 1's face 1 (xi1-), node for
// node -- and that face never goes to memory: its plus side is the partner's staged minus side
// (state out of sQ, face fields out of sM or, for large elements, out of the face-plane buffer sP).
// The xi1 faces are the expensive gathers (stride Nq doubles: every line of the neighbour's column
// is touched for one value in Nq), so a pair removes 1/2 of them.  The host builds the pair lists
// at create (EngineBase::build_pairs): entries (e0, e1) 1-based, e1 < 0 for two unrelated elements
// sharing a work-group, 0 for none.  Small elements keep the minus side staged (two 192-thread
// halves, wave aligned); large ones read it from memory as the unpaired two-element shape does.
enum { TEND_FUSED = 0, TEND_VOLUME = 1, TEND_FACES = 2 };
template <class P, int NQ, int NQV, bool PAIR = false>
struct TendencyShape {
    using KD = KDims<NQ, NQV>;
    static constexpr bool PAIRABLE = CMDG_TEND_PAIRS != 0 && node_cache_size<P>::value == 0 && NQ == NQV;
    static constexpr bool SPLIT = KD::Np > 125 && node_cache_size<P>::value == 0 && CMDG_TEND_SPLIT_LARGE != 0 && !PAIR;
    static constexpr int EPB = PAIR ? 2
                                    : (node_cache_size<P>::value != 0 || SPLIT
                                           ? 1
                                           : (KD::Np > 125 ? CMDG_TEND_EPB_LARGE : CMDG_TEND_EPB_SMALL));
    static constexpr bool STAGE_M = EPB == 1 || (PAIR && KD::Np <= 125);
    static constexpr int NTE = EPB == 1 || (PAIR && KD::Np <= 125) ? KD::NT : (KD::Np > KD::NFT ? KD::Np : KD::NFT);
    static constexpr int NT = EPB == 1 ? KD::NT : ((EPB * NTE + 63) / 64) * 64;
    static constexpr int NTV = ((KD::Np + 63) / 64) * 64;  // threads of the volume half
    static int64_t blocks(int64_t nelems) { return (nelems + EPB - 1) / EPB; }
};

// ---------------------------------------------------------------------------------
// Tendency pass: volume_tendency! (:64-548) + dgsem_interface_tendency! (:588-901),
// optionally fused with the LSRK update! (LowStorageRungeKuttaMethod.jl:146-158).
// RECV: the plus side of ghost neighbours comes from the receive buffers (exterior launches of a
// handle whose exchanges are not unpacked).  A variant of its own: the second addressing mode
// costs the Held-Suarez instantiation 24 VGPRs (128 -> 150, one wave per SIMD less), which the
// interior launches and single-rank handles do not pay.
template <class P, int NQ, int NQV, int MODE, bool PAIR = false>
constexpr int tendency_threads()
{
    return MODE == TEND_VOLUME ? TendencyShape<P, NQ, NQV, PAIR>::NTV : TendencyShape<P, NQ, NQV, PAIR>::NT;
}
// doubles of LDS a paired fused launch takes (what tendency_body lays out; checked there)
template <class P, int NQ, int NQV, bool USE_GF, bool PAIR>
struct TendencyLds {
    using KD = KDims<NQ, NQV>;
    using SH = TendencyShape<P, NQ, NQV, PAIR>;
    static constexpr int NPLANE = P::NFAUX + (USE_GF ? P::NGF : 0) + P::NHYP;
    static constexpr int NMF = (SH::STAGE_M ? NPLANE : 0) + node_cache_size<P>::value;
    static constexpr int doubles =
        NQ * NQ + (NQV == NQ ? 0 : NQV * NQV) + SH::EPB * 3 * P::NS * KD::Np +
        SH::EPB * (NMF > 0 ? NMF : 1) * (NMF > 0 ? SurfDims<NQ, NQV>::NSURF : 1) + SH::EPB * P::NS * KD::Np +
        (PAIR && !SH::STAGE_M && NPLANE > 0 ? 2 * NPLANE * KD::Nfph : 1);
};

template <class P, int NQ, int NQV, bool LSRK, bool USE_GF, bool RECV = false, int MODE = TEND_FUSED,
          bool PAIR = false, bool DYNLDS = false>
__device__ __forceinline__ void tendency_body(const PassArgs<P> &a)
{
    using KD = KDims<NQ, NQV>;
    const double a_t = a.tptr ? *a.tptr : a.t;  // (uniform: one scalar load)
    using SH = TendencyShape<P, NQ, NQV, PAIR>;
    constexpr int EPB = SH::EPB;
    constexpr bool VOL = MODE != TEND_FACES, FACES = MODE != TEND_VOLUME;
    static_assert(MODE == TEND_FUSED || (EPB == 1 && node_cache_size<P>::value == 0),
                  "the two-launch form takes one element per work-group and no node cache");
    static_assert(!PAIR || (MODE == TEND_FUSED && SH::PAIRABLE), "paired work-groups: fused form, one order, no node cache");
    constexpr bool STAGE_M = SH::STAGE_M;
    constexpr int Np = KD::Np, NS = P::NS, NAUX = P::NAUX, NGF = P::NGF,
                  NHYP = P::NHYP, NHG = 3 * P::NGL, NFA = P::NFAUX,
                  NSURF = SurfDims<NQ, NQV>::NSURF, NGFS = USE_GF ? NGF : 0,
                  NCA = node_cache_size<P>::value,
                  NMF = (STAGE_M && FACES ? NFA + NGFS + NHYP : 0) + NCA, OCA = NMF - NCA;
    // LDS: sD derivative matrices; sF_ contravariant flux [d][s][ijk], later the accumulator
    // (interface half alone: the accumulator); sM_ minus side, surface nodes [field][sidx]; sQ_ the
    // prognostic state of every node (minus side of the faces, and the "Q" of the fused update at the
    // end: re-reading it from memory 20 us after the first read misses L2); sP_ (PAIR, minus side
    // not staged) the face fields of the nodes of the shared face [elem][field][n].
    // DYNLDS: the same arrays carved out of the launch's dynamic LDS -- the compiler then does not
    // know the footprint and cannot relax a register budget on its account (k_tendency_pair_small).
    constexpr int NPF = PAIR && !STAGE_M ? NFA + NGFS + NHYP : 0, NPL = KD::Nfph;
    constexpr int LD = VOL ? NQ * NQ + (NQV == NQ ? 0 : NQV * NQV) : 1, LF = EPB * (VOL ? 3 : 1) * NS * Np,
                  LM = EPB * (NMF > 0 ? NMF : 1) * (NMF > 0 ? NSURF : 1), LQ = FACES ? EPB * NS * Np : 1,
                  LP = NPF > 0 ? 2 * NPF * NPL : 1;
    static_assert(!DYNLDS || TendencyLds<P, NQ, NQV, USE_GF, PAIR>::doubles == LD + LF + LM + LQ + LP,
                  "TendencyLds out of step with the kernel");
    double *sD, *sF_, *sM_, *sQ_, *sP_;
    if constexpr (DYNLDS) {
        extern __shared__ double cmdg_dyn_lds[];
        sD = cmdg_dyn_lds;
        sF_ = sD + LD;
        sM_ = sF_ + LF;
        sQ_ = sM_ + LM;
        sP_ = sQ_ + LQ;
    } else {
        __shared__ double aD[LD], aF[LF], aM[LM], aQ[LQ], aP[LP];
        sD = aD, sF_ = aF, sM_ = aM, sQ_ = aQ, sP_ = aP;
    }
    const double *const sDv = sD + (NQV == NQ ? 0 : NQ * NQ);  // vertical derivative matrix
    // this thread's element of the work-group and its index there
    const int sub = EPB == 1 ? 0 : (int)threadIdx.x / SH::NTE;
    const int tid = EPB == 1 ? (int)threadIdx.x : (int)threadIdx.x - sub * SH::NTE;
    const int64_t li = (int64_t)EPB * xcd_remap(blockIdx.x, gridDim.x) + sub;
    // PAIR: entry (e0, e1) of the pair list; e1 > 0: the two share e0's xi1+ face
    int64_t raw = 0;
    bool paired = false;
    if constexpr (PAIR) {
        const int64_t r1 = a.elems[li - sub + 1];
        paired = r1 > 0;
        raw = sub == 0 ? a.elems[li - sub] : (sub == 1 ? (r1 < 0 ? -r1 : r1) : 0);
    }
    const bool live = PAIR ? raw != 0 : (EPB == 1 || (sub < EPB && li < a.nelems));
    const int64_t e = live ? (PAIR ? raw : a.elems[li]) - 1 : 0;
    const int lsub = live ? sub : 0;
    double *const sF = sF_ + lsub * ((VOL ? 3 : 1) * NS * Np);
    double *const sM = sM_ + lsub * ((NMF > 0 ? NMF : 1) * (NMF > 0 ? NSURF : 1));
    double *const sQ = sQ_ + (FACES ? lsub * (NS * Np) : 0);
    double *const sT = sF;              // tendency accumulator [s][ijk] (aliases sF after phase 2)
    if constexpr (VOL) {
        if (threadIdx.x < NQ * NQ) sD[threadIdx.x] = a.g.D[threadIdx.x];
        if constexpr (NQV != NQ) {
            if (threadIdx.x < NQV * NQV) sD[NQ * NQ + threadIdx.x] = a.g.Dv[threadIdx.x];
        }
    }
    const bool hz = a.direction != DIR_VERTICAL, vt = a.direction != DIR_HORIZONTAL;
    // USE_GF: does flux_second_order depend on the gradient-flux state at all?  With zero
    // viscosity (Held-Suarez) tau = -2*0*S and D_t = 0: the 9 fields only ever multiply
    // zeros, so the host picks the instantiation that does not read them (identical results).
    constexpr bool use_gf = NGF > 0 && USE_GF;
    int32_t f_idP = 0;
    int f_bctag = 0;
    bool face_on = false;
    if constexpr (FACES) {
        int f_f = 0, f_n = 0;
        KD::face_task(tid, f_f, f_n);
        face_on = live && tid < KD::NFT && (f_f < 4 ? hz : vt);
        // (the shared face of a pair has no table entry to load: interior, plus side in LDS)
        if (face_on && !(PAIR && paired && f_f == 1 - lsub)) face_index<NQ, NQV>(a.g, e, tid, f_f, f_idP, f_bctag);
    }
    Vec<NS> S;
    double MI = 0;
    if constexpr (MODE == TEND_FACES) {
        // interface half: the volume part of the tendency (the other launch stored it), the
        // state, and the minus side of the surface nodes
        if (live && tid < Np) {
            const int sidx = surf_index<NQ, NQV>(tid);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int64_t o = tid + (int64_t)Np * (s + (int64_t)NS * e);
                sT[s * Np + tid] = a.tendency[o];
                sQ[s * Np + tid] = a.Q[o];
            }
            if (sidx >= 0) {
#pragma unroll
                for (int s = 0; s < NFA; ++s)
                    sM[s * NSURF + sidx] = a.aux[tid + (int64_t)Np * (P::face_aux(s) + (int64_t)NAUX * e)];
                if (use_gf) {
#pragma unroll
                    for (int s = 0; s < NGF; ++s)
                        sM[(NFA + s) * NSURF + sidx] = a.gf[gf_at<P, Np>(tid, s, e)];
                }
#pragma unroll
                for (int s = 0; s < NHYP; ++s)
                    sM[(NFA + NGFS + s) * NSURF + sidx] = a.hypgrad[hg_at<NHG, Np>(tid, s, e)];
            }
        }
    }
    if (VOL && live && tid < Np) {
        const double *vg = a.g.vgeo + (int64_t)Np * a.g.nvgeo * e + tid;
        const double M = vg[VM * Np];
        MI = vg[VMI * Np];
        Vec<NS> lQ;
        Vec<NAUX> laux;
        Vec<NGF> lgf;
        Vec<NHYP> lhyp;
        double x11 = 0, x12 = 0, x13 = 0, x21 = 0, x22 = 0, x23 = 0, x31 = 0, x32 = 0, x33 = 0;
        // the metric rows with the first batch of loads -- unless that costs a work-group of
        // residency: the hyperdiffusive inviscid instantiation at N <= 4 (Held-Suarez) sits at 126
        // VGPRs, five work-groups per CU; with nine more doubles live across the flux arithmetic it
        // takes 148 and runs on four (661 us against 636; rising bubble 132 -> 124 us, BOMEX 589 -> 578)
        constexpr bool METRICS_FIRST = (CMDG_TEND_HOIST & 2) != 0 && !(NHYP > 0 && !use_gf && Np <= 125 && NS >= 5);
        if constexpr (METRICS_FIRST) {
            if (hz) {
                x11 = vg[XI1X1 * Np], x12 = vg[XI1X2 * Np], x13 = vg[XI1X3 * Np];
                x21 = vg[XI2X1 * Np], x22 = vg[XI2X2 * Np], x23 = vg[XI2X3 * Np];
            }
            if (vt) x31 = vg[XI3X1 * Np], x32 = vg[XI3X2 * Np], x33 = vg[XI3X3 * Np];
        }
        load_state<NS, Np>(lQ, a.Q, tid, e);
        load_state<NAUX, Np>(laux, a.aux, tid, e);
#pragma unroll
        for (int s = 0; s < NGF; ++s) lgf[s] = 0.0;
        if (use_gf) load_gf<P, Np>(lgf, a.gf, tid, e);
#pragma unroll
        for (int s = 0; s < NHYP; ++s)
            lhyp[s] = a.hypgrad[hg_at<NHG, Np>(tid, s, e)];
        if constexpr (METRICS_FIRST) __builtin_amdgcn_sched_barrier(0);  // (all of them in flight here)
        const int sidx = surf_index<NQ, NQV>(tid);
        if constexpr (FACES) {
#pragma unroll
            for (int s = 0; s < NS; ++s) sQ[s * Np + tid] = lQ[s];
        }
        if constexpr (NPF > 0) {  // the shared face's fields, for the partner's plus side
            if (paired && tid % NQ == (lsub == 0 ? NQ - 1 : 0)) {
                double *sP = sP_ + lsub * (NPF * NPL) + tid / NQ;
#pragma unroll
                for (int s = 0; s < NFA; ++s) sP[s * NPL] = laux[P::face_aux(s)];
                if (use_gf) {
#pragma unroll
                    for (int s = 0; s < NGF; ++s) sP[(NFA + s) * NPL] = lgf[s];
                }
#pragma unroll
                for (int s = 0; s < NHYP; ++s) sP[(NFA + NGFS + s) * NPL] = lhyp[s];
            }
        }
        if (FACES && STAGE_M && sidx >= 0) {  // stage the minus side of the interface phase
#pragma unroll
            for (int s = 0; s < NFA; ++s) sM[s * NSURF + sidx] = laux[P::face_aux(s)];
            if (use_gf) {
#pragma unroll
                for (int s = 0; s < NGF; ++s) sM[(NFA + s) * NSURF + sidx] = lgf[s];
            }
#pragma unroll
            for (int s = 0; s < NHYP; ++s) sM[(NFA + NGFS + s) * NSURF + sidx] = lhyp[s];
        }
        Vec<3 * NS> F, F2;
        F.negzero();
        if constexpr (NCA > 0) {
            Vec<NCA> lc;
            P::node_cache(a.prm, lc, lQ, laux);
            P::flux_first_order_cached(a.prm, F, lQ, laux, lc);
            if (sidx >= 0) {
#pragma unroll
                for (int s = 0; s < NCA; ++s)
                    sM[(OCA + s) * NSURF + sidx] = lc[s];
            }
        } else {
            P::flux_first_order(a.prm, F, lQ, laux, a_t, a.model_dir);
        }
        F2.negzero();
        P::flux_second_order(a.prm, F2, lQ, lgf, lhyp, laux, a_t);
#pragma unroll
        for (int q = 0; q < 3 * NS; ++q) F[q] += F2[q];
        if (hz) {
            if constexpr (!METRICS_FIRST) {
                x11 = vg[XI1X1 * Np], x12 = vg[XI1X2 * Np], x13 = vg[XI1X3 * Np];
                x21 = vg[XI2X1 * Np], x22 = vg[XI2X2 * Np], x23 = vg[XI2X3 * Np];
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const double F1 = F[3 * s], F2_ = F[3 * s + 1], F3 = F[3 * s + 2];
                sF[(0 * NS + s) * Np + tid] = M * (x11 * F1 + x12 * F2_ + x13 * F3);
                sF[(1 * NS + s) * Np + tid] = M * (x21 * F1 + x22 * F2_ + x23 * F3);
            }
        }
        if (vt) {
            // (both directions' rows in ONE late batch were tried too: nine doubles live across the two
            // blocks cost the same 24 VGPRs as the early batch)
            if constexpr (!METRICS_FIRST) x31 = vg[XI3X1 * Np], x32 = vg[XI3X2 * Np], x33 = vg[XI3X3 * Np];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const double F1 = F[3 * s], F2_ = F[3 * s + 1], F3 = F[3 * s + 2];
                sF[(2 * NS + s) * Np + tid] = M * (x31 * F1 + x32 * F2_ + x33 * F3);
            }
        }
        S.negzero();
#ifndef CMDG_DBG_NOSRC
        if constexpr (P::HAS_SOURCE) {
            Vec<P::NDER> lder;
            load_state<P::NDER, Np>(lder, a.derived, tid, e);
            P::source(a.prm, S, lQ, lgf, laux, lder, a_t, a.model_dir);
        }
#endif
    }
    __syncthreads();
    Vec<NS> Tv;
    if (VOL && live && tid < Np) {
        const int i = tid % NQ, j = (tid / NQ) % NQ, k = tid / (NQ * NQ);
        Vec<NS> Tprev;
        if constexpr ((CMDG_TEND_HOIST & 1) != 0) {
#pragma unroll
            for (int s = 0; s < NS; ++s) Tprev[s] = 0.0;
            if (a.beta != 0) {
#pragma unroll
                for (int s = 0; s < NS; ++s) Tprev[s] = a.tendency[tid + (int64_t)Np * (s + (int64_t)NS * e)];
            }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            double T = 0.0;
            const double Told = (CMDG_TEND_HOIST & 1) != 0
                                    ? Tprev[s]
                                    : (a.beta != 0 ? a.tendency[tid + (int64_t)Np * (s + (int64_t)NS * e)] : 0.0);
            if (hz) {  // generic kernel called with HorizontalDirection() (:64-309)
                double lt = 0.0;
                if (a.direction == DIR_HORIZONTAL && P::HAS_SOURCE) lt += S[s];
#ifdef CMDG_DBG_NOCONTRACT
                lt += MI * sD[i] * sF[(0 * NS + s) * Np + tid] + MI * sD[j] * sF[(1 * NS + s) * Np + tid];
                if (false)
#endif
#pragma unroll
                for (int n = 0; n < NQ; ++n) {
                    lt += MI * sD[n + NQ * i] * sF[(0 * NS + s) * Np + n + NQ * (j + NQ * k)];
                    lt += MI * sD[n + NQ * j] * sF[(1 * NS + s) * Np + i + NQ * (n + NQ * k)];
                }
                T = a.beta != 0 ? a.alpha * lt + a.beta * Told : a.alpha * lt;
            }
            if (vt) {  // ::VerticalDirection kernel (:312-548); beta = true after the
                       // horizontal call in EveryDirection (SpaceDiscretization.jl:1192)
                double lt = 0.0;
#ifdef CMDG_DBG_NOCONTRACT
                lt += MI * sDv[k] * sF[(2 * NS + s) * Np + tid] + S[s];
                if (false)
#endif
#pragma unroll
                for (int kk = 0; kk < NQV; ++kk) {
                    lt += MI * sDv[kk + NQV * k] * sF[(2 * NS + s) * Np + i + NQ * (j + NQ * kk)];
                    if (kk == k && P::HAS_SOURCE) lt += S[s];
                }
                if (hz)
                    T = a.alpha * lt + T;
                else
                    T = a.beta != 0 ? a.alpha * lt + a.beta * Told : a.alpha * lt;
            }
            Tv[s] = T;
        }
    }
    if constexpr (MODE == TEND_VOLUME) {  // volume half: store, the interface launch goes on from here
        if (live && tid < Np) {
#pragma unroll
            for (int s = 0; s < NS; ++s) a.tendency[tid + (int64_t)Np * (s + (int64_t)NS * e)] = Tv[s];
        }
        return;
    }
    if constexpr (MODE == TEND_FUSED) {
        __syncthreads();  // every read of sF is done: it becomes the accumulator sT
        if (live && tid < Np) {
#pragma unroll
            for (int s = 0; s < NS; ++s) sT[s * Np + tid] = Tv[s];
        }
    }
    // ---- faces: dgsem_interface_tendency! ------------------------------------------
    Vec<NS> lift;
    int vidM = 0, fpair = -1;
#ifdef CMDG_DBG_NOFACE
    if (false) {
#else
    if (tid < KD::NFT) {
#endif
        int f, n;  // recomputed: cheaper than two registers live across the volume phases
        KD::face_task(tid, f, n);
        if (face_on) {
            const int facedir = f < 4 ? DIR_HORIZONTAL : DIR_VERTICAL;
            FacePt fp;
            face_geometry<NQ, NQV>(a.g, e, tid, f, n, f_idP, f_bctag, fp);
            Vec<NS> QM, QPn, QPd, flux;
            Vec<NAUX> auxM, auxPn, auxPd;
            Vec<NGF> gfM, gfP;
            Vec<NHYP> hypM, hypP;
            const int sidx = surf_index<NQ, NQV>(fp.vidM);
#pragma unroll
            for (int s = 0; s < NAUX; ++s) auxM[s] = 0;
#pragma unroll
            for (int s = 0; s < NS; ++s) QM[s] = sQ[s * Np + fp.vidM];
            Vec<NCA> cM;  // the law's per-node cache of the minus side (see node_cache_size)
#pragma unroll
            for (int s = 0; s < NCA; ++s) cM[s] = sM[(OCA + s) * NSURF + sidx];
#pragma unroll
            for (int s = 0; s < NFA; ++s)
                auxM[P::face_aux(s)] =
                    STAGE_M ? sM[s * NSURF + sidx]
                            : a.aux[fp.vidM + (int64_t)Np * (P::face_aux(s) + (int64_t)NAUX * e)];
#pragma unroll
            for (int s = 0; s < NGF; ++s) gfM[s] = gfP[s] = 0.0;
            // PAIR: the face this element shares with its partner in the work-group
            const bool shared_face = PAIR && paired && f == 1 - lsub;
            const int gslot = RECV && !shared_face ? ghost_slot<Np>(a.h, fp.eP, fp.vidP) : -1;
            if (use_gf) {
                if constexpr (STAGE_M) {
#pragma unroll
                    for (int s = 0; s < NGF; ++s) gfM[s] = sM[(NFA + s) * NSURF + sidx];
                } else {
                    load_gf<P, Np>(gfM, a.gf, fp.vidM, e);
                }
            }
#pragma unroll
            for (int s = 0; s < NHYP; ++s)
                hypM[s] = STAGE_M ? sM[(NFA + NGFS + s) * NSURF + sidx]
                                  : a.hypgrad[hg_at<NHG, Np>(fp.vidM, s, e)];
#pragma unroll
            for (int s = 0; s < NAUX; ++s) auxPn[s] = 0;
            if (shared_face) {  // plus side = the partner's staged minus side at the same face node
                const int psub = 1 - lsub, vidP = face_vid<NQ, NQV>(f ^ 1, n);
                const double *sQp = sQ_ + psub * (NS * Np);
#pragma unroll
                for (int s = 0; s < NS; ++s) QPn[s] = sQp[s * Np + vidP];
                if constexpr (STAGE_M) {
                    const double *sMp = sM_ + psub * ((NMF > 0 ? NMF : 1) * (NMF > 0 ? NSURF : 1));
                    const int sidxP = surf_index<NQ, NQV>(vidP);
#pragma unroll
                    for (int s = 0; s < NFA; ++s) auxPn[P::face_aux(s)] = sMp[s * NSURF + sidxP];
                    if (use_gf) {
#pragma unroll
                        for (int s = 0; s < NGF; ++s) gfP[s] = sMp[(NFA + s) * NSURF + sidxP];
                    }
#pragma unroll
                    for (int s = 0; s < NHYP; ++s) hypP[s] = sMp[(NFA + NGFS + s) * NSURF + sidxP];
                } else if constexpr (NPF > 0) {
                    const double *sPp = sP_ + psub * (NPF * NPL) + n;
#pragma unroll
                    for (int s = 0; s < NFA; ++s) auxPn[P::face_aux(s)] = sPp[s * NPL];
                    if (use_gf) {
#pragma unroll
                        for (int s = 0; s < NGF; ++s) gfP[s] = sPp[(NFA + s) * NPL];
                    }
#pragma unroll
                    for (int s = 0; s < NHYP; ++s) hypP[s] = sPp[(NFA + NGFS + s) * NPL];
                }
            } else {
                if (use_gf) load_plus_gf<P, Np>(gfP, a.gf, a.h.recvGF, gslot, fp.vidP, fp.eP);
                load_plus<NS, Np, NS>(QPn, a.Q, a.h.recvQ, gslot, fp.vidP, fp.eP);
#pragma unroll
                for (int s = 0; s < NFA; ++s)
                    auxPn[P::face_aux(s)] =
                        a.aux[fp.vidP + (int64_t)Np * (P::face_aux(s) + (int64_t)NAUX * fp.eP)];
                load_plus_hg<NHG, Np, NHYP>(hypP, a.hypgrad, a.h.recvHG, gslot, fp.vidP, fp.eP);
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) QPd[s] = QPn[s];
#pragma unroll
            for (int s = 0; s < NAUX; ++s) auxPd[s] = auxPn[s];
            flux.negzero();
            if (fp.bctag == 0) {
                nf_first_order<P>(a.prm, a.nf_first, flux, fp.n, QM, auxM, QPn, auxPn, a_t,
                                  facedir, NCA > 0 ? (const double *)cM : nullptr);
                // CentralNumericalFluxSecondOrder  NumericalFluxes.jl:670-715
                Vec<3 * NS> FM, FP;
                FM.negzero();
                P::flux_second_order(a.prm, FM, QM, gfM, hypM, auxM, a_t);
                FP.negzero();
                P::flux_second_order(a.prm, FP, QPd, gfP, hypP, auxPd, a_t);
                const double nh0 = fp.n[0] / 2, nh1 = fp.n[1] / 2, nh2 = fp.n[2] / 2;
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    flux[s] += (FM[3 * s] + FP[3 * s]) * nh0 +
                               (FM[3 * s + 1] + FP[3 * s + 1]) * nh1 +
                               (FM[3 * s + 2] + FP[3 * s + 2]) * nh2;
            } else {
                // boundary: the plus side is a copy of the minus side (e+ = e-, :686-692)
                // and the full minus-side auxiliary state may be needed by the law
                load_state<NAUX, Np>(auxM, a.aux, fp.vidM, e);
#pragma unroll
                for (int s = 0; s < NAUX; ++s) auxPn[s] = auxPd[s] = auxM[s];
                Vec<NS> Q1;
                Vec<NAUX> aux1;
                Vec<NGF> gf1;
                for (int s = 0; s < NS; ++s) Q1[s] = 0;
                for (int s = 0; s < NAUX; ++s) aux1[s] = 0;
                for (int s = 0; s < NGF; ++s) gf1[s] = 0;
                if (f == 4) {  // bottom face: first interior node (:786-816)
                    load_state<NS, Np>(Q1, a.Q, n + NQ * NQ, e);
                    load_state<NAUX, Np>(aux1, a.aux, n + NQ * NQ, e);
                    if (use_gf) load_gf<P, Np>(gf1, a.gf, n + NQ * NQ, e);
                }
                // numerical_boundary_flux_first_order!  NumericalFluxes.jl:163-205
                P::boundary_state(a.prm, BS_FIRST, fp.bctag, QPn, auxPn, fp.n, QM, auxM, a_t, Q1,
                                  aux1);
                nf_first_order<P>(a.prm, a.nf_first, flux, fp.n, QM, auxM, QPn, auxPn, a_t,
                                  facedir, NCA > 0 ? (const double *)cM : nullptr);
                // normal_boundary_flux_second_order!  NumericalFluxes.jl:872-918
                Vec<3 * NS> FP;
                FP.negzero();
                P::boundary_flux_second_order(a.prm, fp.bctag, FP, QPd, gfP, hypP, auxPd, fp.n, QM,
                                              gfM, hypM, auxM, a_t, Q1, gf1, aux1);
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    flux[s] +=
                        FP[3 * s] * fp.n[0] + FP[3 * s + 1] * fp.n[1] + FP[3 * s + 2] * fp.n[2];
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) lift[s] = a.alpha * fp.vMI * fp.sM * flux[s];
            vidM = fp.vidM;
            fpair = f / 2;
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 3; ++p) {  // opposite faces touch disjoint nodes (:898-899)
        if (fpair == p) {
#pragma unroll
            for (int s = 0; s < NS; ++s) sT[s * Np + vidM] -= lift[s];
        }
        __syncthreads();
    }
    if (live && tid < Np) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t o = tid + (int64_t)Np * (s + (int64_t)NS * e);
            const double T = sT[s * Np + tid];
            if constexpr (LSRK) {  // update!: Q += rkb*dt*dQ; dQ *= rka
                a.Qout[o] = sQ[s * Np + tid] + a.rkb_dt * T;
                a.tendency[o] = T * a.rka_next;
            } else {
                a.tendency[o] = T;
            }
        }
    }
    if constexpr (LSRK) {  // the updated state of the nodes of vmapsend, straight to the send buffer
        if (live)
            send_nodes<NS, NS>(a.h, 0, e, tid, EPB == 1 ? (int)SH::NT : (int)SH::NTE, [&](int s, int n) {
                return sQ[s * Np + n] + a.rkb_dt * sT[s * Np + n];
            });
    }
}

template <class P, int NQ, int NQV, bool LSRK, bool USE_GF, bool RECV = false, int MODE = TEND_FUSED,
          bool PAIR = false>
__global__ void __launch_bounds__((tendency_threads<P, NQ, NQV, MODE, PAIR>()),
                                   (MODE == TEND_FUSED ? CMDG_TEND_MINW
                                                       : (MODE == TEND_VOLUME ? CMDG_TENDV_MINW : CMDG_TENDF_MINW)))
    k_tendency(const PassArgs<P> a)
{
    tendency_body<P, NQ, NQV, LSRK, USE_GF, RECV, MODE, PAIR>(a);
}
// Paired work-groups of small elements: 384 threads are six waves, which a CU places 2-1-2-1 over its
// SIMDs, so a second work-group is resident only at <= 128 VGPRs (profiles/r02_lds_occupancy.jsonl;
// unconstrained the Held-Suarez instantiation takes 154 and runs at half the occupancy, twice the
// time: profiles/r04_ab_tendency_pairs.txt).  Neither a waves-per-SIMD request nor amdgpu_num_vgpr
// gets there with static LDS -- the compiler relaxes both to what 62 KB of LDS per work-group
// allow, three waves per SIMD ("failed to meet occupancy target") -- so this kernel takes its LDS
// dynamically: the footprint is then unknown at compile time and four waves per SIMD are honoured.
template <class P, int NQ, int NQV, bool LSRK, bool USE_GF, bool RECV>
__global__ void __launch_bounds__((tendency_threads<P, NQ, NQV, TEND_FUSED, true>()), 4)
    k_tendency_pair_small(const PassArgs<P> a)
{
    tendency_body<P, NQ, NQV, LSRK, USE_GF, RECV, TEND_FUSED, true, true>(a);
}

// ---------------------------------------------------------------------------------
// Tendency pass of LARGE elements (Np > 125) on four-wave work-groups (round 4).  The fused
// two-element shape above keeps ONE eleven-wave work-group on a CU (132 KB of LDS, 167 VGPRs):
// its phases -- loads, 2 800 fp64 VALU instructions per wave, barrier, contraction, gathers, flux,
// lift -- overlap with nobody's, and the kernel is bound by their latencies, not by bytes
// (profiles/r04_bomex_n6_8192_pmc_sq.json, r04_ab_tendency_pairs.txt).  A second SIX-wave
// work-group is resident only at <= 128 VGPRs (they are placed 2-1-2-1 over the SIMDs), which the
// moist law's interface phase cannot meet.  FOUR-wave work-groups are placed 1-1-1-1: three share
// a CU at <= 168 VGPRs if each takes <= 53 KB of LDS.  So: 256 threads per element, every thread
// owns node tid and node tid + 256 (the second round is 87 nodes = two waves at N = 6), face tasks
// likewise in two rounds; LDS holds the three contravariant fluxes only (49 KB) -- the state is
// not staged (the minus side of the faces and the fused update read it again, L2-warm) and the
// source is kept in registers.  Arithmetic and summation order are those of tendency_body.
// Measured (BOMEX, N = 6): 869 us against 596 us for the shape above -- three independent work-groups
// per CU do not make up for 204 B of scratch per thread, the state read three times instead of once
// and second rounds that fill a quarter of the lanes.  Not compiled in by default (CMDG_TEND_FOUR_WAVES).
template <class P, int NQ, int NQV, bool LSRK, bool USE_GF, bool RECV>
__global__ void __launch_bounds__(256, 3) k_tendency_big(const PassArgs<P> a)
{
    using KD = KDims<NQ, NQV>;
    static_assert(node_cache_size<P>::value == 0, "no node cache in the four-wave shape");
    const double a_t = a.tptr ? *a.tptr : a.t;
    constexpr int Np = KD::Np, NS = P::NS, NAUX = P::NAUX, NGF = P::NGF, NHYP = P::NHYP, NHG = 3 * P::NGL,
                  NFA = P::NFAUX, NT = 256, RV = (Np + NT - 1) / NT, RF = (KD::NFT + NT - 1) / NT;
    static_assert(RV <= 2 && RF <= 2, "two rounds at most");
    constexpr bool use_gf = NGF > 0 && USE_GF;
    __shared__ double sD[NQ * NQ + (NQV == NQ ? 0 : NQV * NQV)];
    const double *const sDv = sD + (NQV == NQ ? 0 : NQ * NQ);
    __shared__ double sF[3 * NS * Np];  // [d][s][ijk]; afterwards the accumulator sT[s][ijk]
    double *const sT = sF;
    const int tid = threadIdx.x;
    const int64_t e = a.elems[xcd_remap(blockIdx.x, gridDim.x)] - 1;
    if (tid < NQ * NQ) sD[tid] = a.g.D[tid];
    if constexpr (NQV != NQ) {
        if (tid < NQV * NQV) sD[NQ * NQ + tid] = a.g.Dv[tid];
    }
    const bool hz = a.direction != DIR_VERTICAL, vt = a.direction != DIR_HORIZONTAL;
    Vec<NS> S[RV];
    double MI[RV];
    // ---- phase 1: pointwise physics of the nodes of both rounds
#pragma unroll
    for (int r = 0; r < RV; ++r) {
        const int node = tid + NT * r;
        MI[r] = 0;
        S[r].negzero();
        if (node < Np) {
            const double *vg = a.g.vgeo + (int64_t)Np * a.g.nvgeo * e + node;
            const double M = vg[VM * Np];
            MI[r] = vg[VMI * Np];
            Vec<NS> lQ;
            Vec<NAUX> laux;
            Vec<NGF> lgf;
            Vec<NHYP> lhyp;
            load_state<NS, Np>(lQ, a.Q, node, e);
            load_state<NAUX, Np>(laux, a.aux, node, e);
#pragma unroll
            for (int s = 0; s < NGF; ++s) lgf[s] = 0.0;
            if (use_gf) load_gf<P, Np>(lgf, a.gf, node, e);
#pragma unroll
            for (int s = 0; s < NHYP; ++s) lhyp[s] = a.hypgrad[hg_at<NHG, Np>(node, s, e)];
            Vec<3 * NS> F, F2;
            F.negzero();
            P::flux_first_order(a.prm, F, lQ, laux, a_t, a.model_dir);
            F2.negzero();
            P::flux_second_order(a.prm, F2, lQ, lgf, lhyp, laux, a_t);
#pragma unroll
            for (int q = 0; q < 3 * NS; ++q) F[q] += F2[q];
            if (hz) {
                const double x11 = vg[XI1X1 * Np], x12 = vg[XI1X2 * Np], x13 = vg[XI1X3 * Np];
                const double x21 = vg[XI2X1 * Np], x22 = vg[XI2X2 * Np], x23 = vg[XI2X3 * Np];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const double F1 = F[3 * s], F2_ = F[3 * s + 1], F3 = F[3 * s + 2];
                    sF[(0 * NS + s) * Np + node] = M * (x11 * F1 + x12 * F2_ + x13 * F3);
                    sF[(1 * NS + s) * Np + node] = M * (x21 * F1 + x22 * F2_ + x23 * F3);
                }
            }
            if (vt) {
                const double x31 = vg[XI3X1 * Np], x32 = vg[XI3X2 * Np], x33 = vg[XI3X3 * Np];
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const double F1 = F[3 * s], F2_ = F[3 * s + 1], F3 = F[3 * s + 2];
                    sF[(2 * NS + s) * Np + node] = M * (x31 * F1 + x32 * F2_ + x33 * F3);
                }
            }
            if constexpr (P::HAS_SOURCE) {
                Vec<P::NDER> lder;
                load_state<P::NDER, Np>(lder, a.derived, node, e);
                P::source(a.prm, S[r], lQ, lgf, laux, lder, a_t, a.model_dir);
            }
        }
    }
    __syncthreads();
    // ---- phase 2: contractions (reference order: horizontal part, then vertical with the source)
    Vec<NS> Tv[RV];
#pragma unroll
    for (int r = 0; r < RV; ++r) {
        const int node = tid + NT * r;
        if (node < Np) {
            const int i = node % NQ, j = (node / NQ) % NQ, k = node / (NQ * NQ);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                double T = 0.0;
                const double Told = a.beta != 0 ? a.tendency[node + (int64_t)Np * (s + (int64_t)NS * e)] : 0.0;
                if (hz) {
                    double lt = 0.0;
                    if (a.direction == DIR_HORIZONTAL && P::HAS_SOURCE) lt += S[r][s];
#pragma unroll
                    for (int n = 0; n < NQ; ++n) {
                        lt += MI[r] * sD[n + NQ * i] * sF[(0 * NS + s) * Np + n + NQ * (j + NQ * k)];
                        lt += MI[r] * sD[n + NQ * j] * sF[(1 * NS + s) * Np + i + NQ * (n + NQ * k)];
                    }
                    T = a.beta != 0 ? a.alpha * lt + a.beta * Told : a.alpha * lt;
                }
                if (vt) {
                    double lt = 0.0;
#pragma unroll
                    for (int kk = 0; kk < NQV; ++kk) {
                        lt += MI[r] * sDv[kk + NQV * k] * sF[(2 * NS + s) * Np + i + NQ * (j + NQ * kk)];
                        if (kk == k && P::HAS_SOURCE) lt += S[r][s];
                    }
                    if (hz)
                        T = a.alpha * lt + T;
                    else
                        T = a.beta != 0 ? a.alpha * lt + a.beta * Told : a.alpha * lt;
                }
                Tv[r][s] = T;
            }
        }
    }
    __syncthreads();  // every read of sF is done: it becomes the accumulator sT
#pragma unroll
    for (int r = 0; r < RV; ++r) {
        const int node = tid + NT * r;
        if (node < Np) {
#pragma unroll
            for (int s = 0; s < NS; ++s) sT[s * Np + node] = Tv[r][s];
        }
    }
    __syncthreads();
    // ---- faces, one round of tasks at a time; a round's lifts are applied before the next round
    // (pair by pair: opposite faces touch disjoint nodes, and every node still receives its pairs in
    // the order 1-2, 3-4, 5-6 -- the second round holds tasks of the last face only)
    static_assert(RF == 1 || (RF - 1) * NT >= 4 * KD::Nfph, "later rounds must hold tasks of faces 5, 6 only");
#pragma unroll
    for (int r = 0; r < RF; ++r) {
        const int t = tid + NT * r;
        Vec<NS> lift;
        int vidM = 0, fpair = -1;
        if (t < KD::NFT) {
            int f, n;
            KD::face_task(t, f, n);
            if (f < 4 ? hz : vt) {
                const int facedir = f < 4 ? DIR_HORIZONTAL : DIR_VERTICAL;
                FacePt fp;
                face_setup<NQ, NQV>(a.g, e, t, f, n, fp);
                Vec<NS> QM, QPn, QPd, flux;
                Vec<NAUX> auxM, auxPn, auxPd;
                Vec<NGF> gfM, gfP;
                Vec<NHYP> hypM, hypP;
#pragma unroll
                for (int s = 0; s < NAUX; ++s) auxM[s] = 0;
                load_state<NS, Np>(QM, a.Q, fp.vidM, e);
#pragma unroll
                for (int s = 0; s < NFA; ++s)
                    auxM[P::face_aux(s)] = a.aux[fp.vidM + (int64_t)Np * (P::face_aux(s) + (int64_t)NAUX * e)];
#pragma unroll
                for (int s = 0; s < NGF; ++s) gfM[s] = gfP[s] = 0.0;
                const int gslot = RECV ? ghost_slot<Np>(a.h, fp.eP, fp.vidP) : -1;
                if (use_gf) {
                    load_gf<P, Np>(gfM, a.gf, fp.vidM, e);
                    load_plus_gf<P, Np>(gfP, a.gf, a.h.recvGF, gslot, fp.vidP, fp.eP);
                }
#pragma unroll
                for (int s = 0; s < NHYP; ++s)
                    hypM[s] = a.hypgrad[hg_at<NHG, Np>(fp.vidM, s, e)];
                load_plus<NS, Np, NS>(QPn, a.Q, a.h.recvQ, gslot, fp.vidP, fp.eP);
#pragma unroll
                for (int s = 0; s < NAUX; ++s) auxPn[s] = 0;
#pragma unroll
                for (int s = 0; s < NFA; ++s)
                    auxPn[P::face_aux(s)] =
                        a.aux[fp.vidP + (int64_t)Np * (P::face_aux(s) + (int64_t)NAUX * fp.eP)];
                load_plus_hg<NHG, Np, NHYP>(hypP, a.hypgrad, a.h.recvHG, gslot, fp.vidP, fp.eP);
#pragma unroll
                for (int s = 0; s < NS; ++s) QPd[s] = QPn[s];
#pragma unroll
                for (int s = 0; s < NAUX; ++s) auxPd[s] = auxPn[s];
                flux.negzero();
                if (fp.bctag == 0) {
                    nf_first_order<P>(a.prm, a.nf_first, flux, fp.n, QM, auxM, QPn, auxPn, a_t, facedir, nullptr);
                    Vec<3 * NS> FM, FP;
                    FM.negzero();
                    P::flux_second_order(a.prm, FM, QM, gfM, hypM, auxM, a_t);
                    FP.negzero();
                    P::flux_second_order(a.prm, FP, QPd, gfP, hypP, auxPd, a_t);
                    const double nh0 = fp.n[0] / 2, nh1 = fp.n[1] / 2, nh2 = fp.n[2] / 2;
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        flux[s] += (FM[3 * s] + FP[3 * s]) * nh0 + (FM[3 * s + 1] + FP[3 * s + 1]) * nh1 +
                                   (FM[3 * s + 2] + FP[3 * s + 2]) * nh2;
                } else {
                    load_state<NAUX, Np>(auxM, a.aux, fp.vidM, e);
#pragma unroll
                    for (int s = 0; s < NAUX; ++s) auxPn[s] = auxPd[s] = auxM[s];
                    Vec<NS> Q1;
                    Vec<NAUX> aux1;
                    Vec<NGF> gf1;
                    for (int s = 0; s < NS; ++s) Q1[s] = 0;
                    for (int s = 0; s < NAUX; ++s) aux1[s] = 0;
                    for (int s = 0; s < NGF; ++s) gf1[s] = 0;
                    if (f == 4) {
                        load_state<NS, Np>(Q1, a.Q, n + NQ * NQ, e);
                        load_state<NAUX, Np>(aux1, a.aux, n + NQ * NQ, e);
                        if (use_gf) load_gf<P, Np>(gf1, a.gf, n + NQ * NQ, e);
                    }
                    P::boundary_state(a.prm, BS_FIRST, fp.bctag, QPn, auxPn, fp.n, QM, auxM, a_t, Q1, aux1);
                    nf_first_order<P>(a.prm, a.nf_first, flux, fp.n, QM, auxM, QPn, auxPn, a_t, facedir, nullptr);
                    Vec<3 * NS> FP;
                    FP.negzero();
                    P::boundary_flux_second_order(a.prm, fp.bctag, FP, QPd, gfP, hypP, auxPd, fp.n, QM, gfM,
                                                  hypM, auxM, a_t, Q1, gf1, aux1);
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        flux[s] += FP[3 * s] * fp.n[0] + FP[3 * s + 1] * fp.n[1] + FP[3 * s + 2] * fp.n[2];
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) lift[s] = a.alpha * fp.vMI * fp.sM * flux[s];
                vidM = fp.vidM;
                fpair = f / 2;
            }
        }
#pragma unroll
        for (int pp = 0; pp < 3; ++pp) {
            if (fpair == pp) {
#pragma unroll
                for (int s = 0; s < NS; ++s) sT[s * Np + vidM] -= lift[s];
            }
            __syncthreads();
        }
    }
    // ---- store, with the LSRK update fused in
#pragma unroll
    for (int r = 0; r < RV; ++r) {
        const int node = tid + NT * r;
        if (node < Np) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int64_t o = node + (int64_t)Np * (s + (int64_t)NS * e);
                const double T = sT[s * Np + node];
                if constexpr (LSRK) {
                    a.Qout[o] = a.Q[o] + a.rkb_dt * T;
                    a.tendency[o] = T * a.rka_next;
                } else {
                    a.tendency[o] = T;
                }
            }
        }
    }
    if constexpr (LSRK)
        send_nodes<NS, NS>(a.h, 0, e, tid, NT, [&](int s, int n) {
            return a.Q[n + (int64_t)Np * (s + (int64_t)NS * e)] + a.rkb_dt * sT[s * Np + n];
        });
}

// ---------------------------------------------------------------------------------
// Gradient pass: volume_gradients! (:934-1328) + dgsem_interface_gradients! (:1365-1651)
#ifdef CMDG_MFMA_GRAD
// A/B variant (not the default, see DESIGN.md "MFMA"): the (N+1)-point derivative contractions of
// the gradient pass on the matrix pipe, v_mfma_f64_16x16x4_f64.  One 16x16x16 product (four
// instructions) applies blockdiag(D, D, D) to 3 x 16 lines of five values: 48 lines, 2 400 useful
// of 8 192 issued flops (K = 5 does not tile the instruction).  Lines of direction dir (0: xi1,
// 1: xi2, 2: xi3) of the active fields are taken from sG and the derivatives land in
// sDer[dir][field][node]; thread = node then applies the metric terms as the VALU path does.
typedef double cmdg_v4d __attribute__((ext_vector_type(4)));
template <int NFLD>
__device__ __forceinline__ void mfma_line_derivatives(const double *sD, const double *sG,
                                                      double *sDer, unsigned fldpack, int nact,
                                                      int ndir, unsigned dirpack, int tid, int nthreads)
{
    constexpr int NQ = 5, Np = 125;
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    const int nlines = 25 * nact, ngroups = (nlines + 47) / 48;
    const int c = lane & 15, kr = lane >> 4;
    double Aop[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int ip = c, r = 4 * kk + kr;  // A[row ip][col r]
        Aop[kk] = (ip < 15 && r < 15 && ip / 5 == r / 5) ? sD[(ip % 5) + NQ * (r % 5)] : 0.0;
    }
    for (int w = wave; w < ndir * ngroups; w += nwaves) {
        const int dir = (dirpack >> (2 * (w / ngroups))) & 3, g = w % ngroups;
        cmdg_v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int r = 4 * kk + kr, L = 48 * g + 16 * (r / 5) + c, n = r % 5;
            double b = 0.0;
            if (r < 15 && L < nlines) {
                const int aa = L % 25, s = (fldpack >> (4 * (L / 25))) & 15;
                const int node = dir == 0 ? 5 * aa + n
                                          : (dir == 1 ? (aa % 5) + 5 * n + 25 * (aa / 5) : aa + 25 * n);
                b = sG[s * Np + node];
            }
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Aop[kk], b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int ip = kr + 4 * reg, L = 48 * g + 16 * (ip / 5) + c, i = ip % 5;
            if (ip < 15 && L < nlines) {
                const int aa = L % 25, s = (fldpack >> (4 * (L / 25))) & 15;
                const int node = dir == 0 ? 5 * aa + i
                                          : (dir == 1 ? (aa % 5) + 5 * i + 25 * (aa / 5) : aa + 25 * i);
                sDer[(dir * NFLD + s) * Np + node] = acc[reg];
            }
        }
    }
}
#endif

// USE_GF = false: the law's second-order flux does not read the gradient-flux state (zero
// viscosity): only the gradients the hyperdiffusion passes consume are formed and stored, and
// state_gradient_flux is left untouched (cmdg_set_option(CMDG_OPT_KEEP_GRADFLUX) restores it).
template <class P, bool USE_GF>
__host__ __device__ constexpr unsigned gradient_argument_mask()
{
    if (USE_GF) return ~0u;
    unsigned m = 0;
    for (int s = 0; s < P::NGL; ++s) m |= 1u << P::hv_indexmap(s);
    return m;
}

template <class P, int NQ, int NQV = NQ, bool USE_GF = true>
// (six-wave work-groups of the large elements share a CU only at <= 128 VGPRs, see TendencyShape;
// a launch bound of 1024 threads is the hard form of that request)
__global__ void __launch_bounds__((KDims<NQ, NQV>::Np > 125 ? CMDG_GRAD_BOUND_LARGE : KDims<NQ, NQV>::NT),
                                   (NQ == 5 && NQV == 5 ? grad_min_waves<P>::value : 1)) k_gradients(const PassArgs<P> a)
{
    using KD = KDims<NQ, NQV>;
    const double a_t = a.tptr ? *a.tptr : a.t;  // (uniform: one scalar load)
    constexpr int Np = KD::Np, NS = P::NS, NAUX = P::NAUX, NGRAD = P::NGRAD,
                  NGF = USE_GF ? P::NGF : 0, NGL = P::NGL, NHG = 3 * NGL, NACC = NGF + NHG;
    constexpr unsigned GMASK = gradient_argument_mask<P, USE_GF>();
    __shared__ double sD[NQ * NQ + (NQV == NQ ? 0 : NQV * NQV)];
    const double *const sDv = sD + (NQV == NQ ? 0 : NQ * NQ);  // vertical derivative matrix
    __shared__ double sG[(NGRAD > 0 ? NGRAD : 1) * Np];
    __shared__ double sA[(NACC > 0 ? NACC : 1) * Np];  // [gf..., hypgrad...][ijk]
#ifdef CMDG_MFMA_GRAD
    constexpr bool MFMA = NQ == 5 && NQV == 5 && NGRAD > 0;
    __shared__ double sDer[MFMA ? 3 * NGRAD * Np : 1];
#endif
    const int tid = threadIdx.x;
    const int64_t e = a.elems[xcd_remap(blockIdx.x, gridDim.x)] - 1;
    if (tid < NQ * NQ) sD[tid] = a.g.D[tid];
    if constexpr (NQV != NQ) {
        if (tid < NQV * NQV) sD[NQ * NQ + tid] = a.g.Dv[tid];
    }
    const bool hz = a.direction != DIR_VERTICAL, vt = a.direction != DIR_HORIZONTAL;
    Vec<NS> lQ;
    Vec<NAUX> laux;
    if (tid < Np) {
        load_state<NS, Np>(lQ, a.Q, tid, e);
        load_state<NAUX, Np>(laux, a.aux, tid, e);
        if constexpr (P::HAS_UPDATE_AUX && P::FUSE_UPDATE_AUX) {
            // kernel_nodal_update_auxiliary_state! of the real elements, fused: the refreshed
            // entries are read by no kernel of this evaluation (see P::FUSE_UPDATE_AUX)
            P::update_aux(a.prm, lQ, laux, a_t);
#pragma unroll
            for (int s = 0; s < P::NUPD; ++s)
                a.aux_rw[tid + (int64_t)Np * (P::upd_aux(s) + (int64_t)NAUX * e)] =
                    laux[P::upd_aux(s)];
        }
        Vec<NGRAD> G;
        G.negzero();
        P::gradient_argument(a.prm, G, lQ, laux, a_t);
#pragma unroll
        for (int s = 0; s < NGRAD; ++s)
            if (GMASK >> s & 1) sG[s * Np + tid] = G[s];
    }
    __syncthreads();
#ifdef CMDG_MFMA_GRAD
    if constexpr (MFMA) {
        static_assert(NGRAD <= 8, "field ordinals are packed four bits each");
        unsigned fldpack = 0, dirpack = 0;  // active fields / directions, packed
        int nact = 0, ndir = 0;
#pragma unroll
        for (int s = 0; s < NGRAD; ++s)
            if (GMASK >> s & 1) fldpack |= (unsigned)s << (4 * nact++);
        if (hz) dirpack |= 0u << (2 * ndir++), dirpack |= 1u << (2 * ndir++);
        if (vt) dirpack |= 2u << (2 * ndir++);
        mfma_line_derivatives<NGRAD>(sD, sG, sDer, fldpack, nact, ndir, dirpack, tid, KD::NT);
        __syncthreads();
    }
#endif
    if (tid < Np) {
        const int i = tid % NQ, j = (tid / NQ) % NQ, k = tid / (NQ * NQ);
        const double *vg = a.g.vgeo + (int64_t)Np * a.g.nvgeo * e + tid;
        Vec<3 * NGRAD> gh, gv;
        gh.negzero();
        gv.negzero();
        if (hz) {
            const double x11 = vg[XI1X1 * Np], x12 = vg[XI1X2 * Np], x13 = vg[XI1X3 * Np];
            const double x21 = vg[XI2X1 * Np], x22 = vg[XI2X2 * Np], x23 = vg[XI2X3 * Np];
#pragma unroll
            for (int s = 0; s < NGRAD; ++s) {
                if (!(GMASK >> s & 1)) continue;
                double G1 = 0.0, G2 = 0.0;
#if defined(CMDG_MFMA_GRAD)
                if constexpr (MFMA) {
                    G1 = sDer[(0 * NGRAD + s) * Np + tid];
                    G2 = sDer[(1 * NGRAD + s) * Np + tid];
                } else
#elif defined(CMDG_DBG_NOCONTRACT)
                // ablation: the contraction's share of the kernel (results are wrong)
                G1 = sD[i] * sG[s * Np + tid], G2 = sD[j] * sG[s * Np + tid];
                if (false)
#endif
#pragma unroll
                for (int n = 0; n < NQ; ++n) {
                    G1 += sD[i + NQ * n] * sG[s * Np + n + NQ * (j + NQ * k)];
                    G2 += sD[j + NQ * n] * sG[s * Np + i + NQ * (n + NQ * k)];
                }
                gh[3 * s + 0] += x11 * G1;
                gh[3 * s + 1] += x12 * G1;
                gh[3 * s + 2] += x13 * G1;
                gh[3 * s + 0] += x21 * G2;
                gh[3 * s + 1] += x22 * G2;
                gh[3 * s + 2] += x23 * G2;
            }
        }
        if (vt) {
            const double x31 = vg[XI3X1 * Np], x32 = vg[XI3X2 * Np], x33 = vg[XI3X3 * Np];
#pragma unroll
            for (int s = 0; s < NGRAD; ++s) {
                if (!(GMASK >> s & 1)) continue;
                double G3 = -0.0;
#if defined(CMDG_MFMA_GRAD)
                if constexpr (MFMA) {
                    G3 = sDer[(2 * NGRAD + s) * Np + tid];
                } else
#elif defined(CMDG_DBG_NOCONTRACT)
                G3 = sDv[k] * sG[s * Np + tid];
                if (false)
#endif
#pragma unroll
                for (int n = 0; n < NQV; ++n)
                    G3 += sDv[k + NQV * n] * sG[s * Np + i + NQ * (j + NQ * n)];
                gv[3 * s + 0] += x31 * G3;
                gv[3 * s + 1] += x32 * G3;
                gv[3 * s + 2] += x33 * G3;
            }
        }
        // hyperdiffusion gradients: "=" by the first kernel, "+=" by the vertical one
#pragma unroll
        for (int s = 0; s < NGL; ++s)
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int q = d + 3 * P::hv_indexmap(s);
                sA[(NGF + 3 * s + d) * Np + tid] = hz ? (vt ? gh[q] + gv[q] : gh[q]) : gv[q];
            }
        if constexpr (NGF > 0) {
            Vec<NGF> o1, o2;
            o1.negzero();
            o2.negzero();
            Vec<NGRAD> Gn;  // this node's gradient argument (every entry is staged when NGF > 0)
#pragma unroll
            for (int s = 0; s < NGRAD; ++s) Gn[s] = (GMASK >> s & 1) ? sG[s * Np + tid] : 0.0;
            if (hz) law_gradient_flux<P>(a.prm, o1, gh, lQ, laux, a_t, Gn);
            if (vt) law_gradient_flux<P>(a.prm, o2, gv, lQ, laux, a_t, Gn);
#pragma unroll
            for (int s = 0; s < NGF; ++s)
                sA[s * Np + tid] = hz ? (vt ? o1[s] + o2[s] : o1[s]) : o2[s];
        }
    }
    __syncthreads();
    Vec<NACC> corr;
    int vidM = 0, fpair = -1;
    if (tid < KD::NFT) {
        int f, n;
        KD::face_task(tid, f, n);
        const bool on = f < 4 ? hz : vt;
        if (on) {
            FacePt fp;
            face_setup<NQ, NQV>(a.g, e, tid, f, n, fp);
            Vec<NS> QM, QP;
            Vec<NAUX> auxM, auxP;
            Vec<NGRAD> GM, GP;
            load_state<NS, Np>(QM, a.Q, fp.vidM, e);       // only kept if the law reads them
            load_state<NAUX, Np>(auxM, a.aux, fp.vidM, e);  // (gradient_flux / boundary_state)
#pragma unroll
            for (int s = 0; s < NGRAD; ++s)
                GM[s] = (GMASK >> s & 1) ? sG[s * Np + fp.vidM] : 0.0;  // == G(Q-, aux-)
            GP.negzero();
            Vec<NGF> lgf;
            lgf.negzero();
            Vec<3 * NGRAD> tg, nGM;
            if (fp.bctag == 0) {  // CentralNumericalFluxGradient  NumericalFluxes.jl:67-83
                // the plus side is read inside the branch that uses it: of the neighbour's
                // auxiliary columns only those the law's gradient argument touches are gathered
                load_plus<NS, Np, NS>(QP, a.Q, a.h.recvQ, ghost_slot<Np>(a.h, fp.eP, fp.vidP), fp.vidP,
                                      fp.eP);
                load_state<NAUX, Np>(auxP, a.aux, fp.vidP, fp.eP);
                P::gradient_argument(a.prm, GP, QP, auxP, a_t);
#pragma unroll
                for (int s = 0; s < NGRAD; ++s)
#pragma unroll
                    for (int d = 0; d < 3; ++d) tg[d + 3 * s] = fp.n[d] * (GP[s] + GM[s]) / 2;
            } else {  // numerical_boundary_flux_gradient!  NumericalFluxes.jl:85-123
                // e+ = e-, vid+ = vid- on boundary faces (:686-692): the plus side starts as
                // a copy of the minus side, already loaded
#pragma unroll
                for (int s = 0; s < NS; ++s) QP[s] = QM[s];
#pragma unroll
                for (int s = 0; s < NAUX; ++s) auxP[s] = auxM[s];
                Vec<NS> Q1;
                Vec<NAUX> aux1;
                for (int s = 0; s < NS; ++s) Q1[s] = 0;
                for (int s = 0; s < NAUX; ++s) aux1[s] = 0;
                if (f == 4) {
                    load_state<NS, Np>(Q1, a.Q, n + NQ * NQ, e);
                    load_state<NAUX, Np>(aux1, a.aux, n + NQ * NQ, e);
                }
                P::boundary_state(a.prm, BS_GRADIENT, fp.bctag, QP, auxP, fp.n, QM, auxM, a_t, Q1,
                                  aux1);
                P::gradient_argument(a.prm, GP, QP, auxP, a_t);
#pragma unroll
                for (int s = 0; s < NGRAD; ++s)
#pragma unroll
                    for (int d = 0; d < 3; ++d) tg[d + 3 * s] = fp.n[d] * GP[s];
            }
            if constexpr (NGF > 0) law_gradient_flux<P>(a.prm, lgf, tg, QM, auxM, a_t, GM);
#pragma unroll
            for (int s = 0; s < NGRAD; ++s)
#pragma unroll
                for (int d = 0; d < 3; ++d) nGM[d + 3 * s] = fp.n[d] * GM[s];
#pragma unroll
            for (int s = 0; s < NGL; ++s)
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int q = d + 3 * P::hv_indexmap(s);
                    corr[NGF + 3 * s + d] = fp.vMI * fp.sM * (tg[q] - nGM[q]);
                }
            if constexpr (NGF > 0) {
                Vec<NGF> visc;
                for (int s = 0; s < NGF; ++s) visc[s] = 0;
                law_gradient_flux<P>(a.prm, visc, nGM, QM, auxM, a_t, GM);
#pragma unroll
                for (int s = 0; s < NGF; ++s) corr[s] = fp.vMI * fp.sM * (lgf[s] - visc[s]);
            }
            vidM = fp.vidM;
            fpair = f / 2;
        }
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        if (fpair == p) {
#pragma unroll
            for (int s = 0; s < NACC; ++s) sA[s * Np + vidM] += corr[s];
        }
        __syncthreads();
    }
    if (tid < Np) {
        if constexpr (!gf_node_major<P>::value) {
#pragma unroll
            for (int s = 0; s < NGF; ++s)
                a.gf[tid + (int64_t)Np * (s + (int64_t)P::NGF * e)] = sA[s * Np + tid];
        }
        if constexpr (!CMDG_HG_NODE_MAJOR) {
#pragma unroll
            for (int s = 0; s < NHG; ++s)
                a.hypgrad[hg_at<NHG, Np>(tid, s, e)] = sA[(NGF + s) * Np + tid];
        }
    }
    if constexpr (gf_node_major<P>::value && NGF > 0)
        store_node_major<P::NGF, Np, NGF>(a.gf, e, tid, (int)blockDim.x, sA);
    if constexpr (CMDG_HG_NODE_MAJOR && NHG > 0)
        store_node_major<NHG, Np, NHG>(a.hypgrad, e, tid, (int)blockDim.x, sA + NGF * Np);
    if constexpr (NGF > 0)
        send_nodes<P::NGF, NGF>(a.h, 0, e, tid, (int)blockDim.x, [&](int s, int n) { return sA[s * Np + n]; });
    if constexpr (NHG > 0)
        send_nodes<NHG, NHG>(a.h, 1, e, tid, (int)blockDim.x,
                             [&](int s, int n) { return sA[(NGF + s) * Np + n]; });
}

// ---------------------------------------------------------------------------------
// Laplacian pass: volume_divergence_of_gradients! (:2132-2329) +
// interface_divergence_of_gradients! (:2360-2494)
template <class P, int NQ, int NQV = NQ>
__global__ void __launch_bounds__((KDims<NQ, NQV>::NT)) k_divgrad(const PassArgs<P> a)
{
    using KD = KDims<NQ, NQV>;
    const double a_t = a.tptr ? *a.tptr : a.t;  // (uniform: one scalar load)
    constexpr int Np = KD::Np, NAUX = P::NAUX, NGL = P::NGL, NHG = 3 * NGL,
                  NHYP = P::NHYP, NG = NGL > 0 ? NGL : 1;
    __shared__ double sD[NQ * NQ + (NQV == NQ ? 0 : NQV * NQV)];
    const double *const sDv = sD + (NQV == NQ ? 0 : NQ * NQ);  // vertical derivative matrix
    __shared__ double sC[3 * NG * Np];  // M * (xi_d . grad) [d][s][ijk]
    __shared__ double sA[NG * Np];
    constexpr int NSURF = SurfDims<NQ, NQV>::NSURF;
    __shared__ double sM[(NHG > 0 ? NHG : 1) * NSURF];  // minus-side gradients, surface nodes
    const int tid = threadIdx.x;
    const int64_t e = a.elems[xcd_remap(blockIdx.x, gridDim.x)] - 1;
    if (tid < NQ * NQ) sD[tid] = a.g.D[tid];
    if constexpr (NQV != NQ) {
        if (tid < NQV * NQV) sD[NQ * NQ + tid] = a.g.Dv[tid];
    }
    const bool hz = a.direction != DIR_VERTICAL, vt = a.direction != DIR_HORIZONTAL;
    double MI = 0;
    if (tid < Np) {
        const double *vg = a.g.vgeo + (int64_t)Np * a.g.nvgeo * e + tid;
        const double M = vg[VM * Np];
        MI = vg[VMI * Np];
        // only the metric rows of the directions this pass differentiates in are read (a
        // horizontal diffusion_direction, Held-Suarez, never touches xi3)
        double x11 = 0, x12 = 0, x13 = 0, x21 = 0, x22 = 0, x23 = 0, x31 = 0, x32 = 0, x33 = 0;
        if (hz) {
            x11 = vg[XI1X1 * Np], x12 = vg[XI1X2 * Np], x13 = vg[XI1X3 * Np];
            x21 = vg[XI2X1 * Np], x22 = vg[XI2X2 * Np], x23 = vg[XI2X3 * Np];
        }
        if (vt) x31 = vg[XI3X1 * Np], x32 = vg[XI3X2 * Np], x33 = vg[XI3X3 * Np];
        Vec<NHG> G;  // (all of them in flight before the first LDS store: -17 % on this pass)
#pragma unroll
        for (int q = 0; q < NHG; ++q) G[q] = a.hypgrad[hg_at<NHG, Np>(tid, q, e)];
#pragma unroll
        for (int s = 0; s < NGL; ++s) {
            const double G1 = G[3 * s + 0], G2 = G[3 * s + 1], G3 = G[3 * s + 2];
            if (hz) {
                sC[(0 * NG + s) * Np + tid] = M * (x11 * G1 + x12 * G2 + x13 * G3);
                sC[(1 * NG + s) * Np + tid] = M * (x21 * G1 + x22 * G2 + x23 * G3);
            }
            if (vt) sC[(2 * NG + s) * Np + tid] = M * (x31 * G1 + x32 * G2 + x33 * G3);
            const int sidx = surf_index<NQ, NQV>(tid);
            if (sidx >= 0) {
                sM[(3 * s + 0) * NSURF + sidx] = G1;
                sM[(3 * s + 1) * NSURF + sidx] = G2;
                sM[(3 * s + 2) * NSURF + sidx] = G3;
            }
        }
    }
    __syncthreads();
    if (tid < Np) {
        const int i = tid % NQ, j = (tid / NQ) % NQ, k = tid / (NQ * NQ);
#pragma unroll
        for (int s = 0; s < NGL; ++s) {
            double dh = 0.0, dv = 0.0;
            if (hz) {
#pragma unroll
                for (int n = 0; n < NQ; ++n) {
                    dh -= MI * sD[n + NQ * i] * sC[(0 * NG + s) * Np + n + NQ * (j + NQ * k)];
                    dh -= MI * sD[n + NQ * j] * sC[(1 * NG + s) * Np + i + NQ * (n + NQ * k)];
                }
            }
            if (vt) {
#pragma unroll
                for (int kk = 0; kk < NQV; ++kk)
                    dv -= MI * sDv[kk + NQV * k] * sC[(2 * NG + s) * Np + i + NQ * (j + NQ * kk)];
            }
            sA[s * Np + tid] = hz ? (vt ? dh + dv : dh) : dv;
        }
    }
    __syncthreads();
    Vec<NGL> corr;
    int vidM = 0, fpair = -1;
    if (tid < KD::NFT) {
        int f, n;
        KD::face_task(tid, f, n);
        const bool on = f < 4 ? hz : vt;
        if (on) {
            FacePt fp;
            face_setup<NQ, NQV>(a.g, e, tid, f, n, fp);
            Vec<NHG> gM, gP;
            const int sidx = surf_index<NQ, NQV>(fp.vidM);
#pragma unroll
            for (int q = 0; q < NHG; ++q) gM[q] = sM[q * NSURF + sidx];
            load_plus_hg<NHG, Np, NHG>(gP, a.hypgrad, a.h.recvHG, ghost_slot<Np>(a.h, fp.eP, fp.vidP),
                                    fp.vidP, fp.eP);
            if (fp.bctag != 0) {  // numerical_boundary_flux_divergence!  :732-763
                Vec<NAUX> auxM, auxP;
                load_state<NAUX, Np>(auxM, a.aux, fp.vidM, e);
                load_state<NAUX, Np>(auxP, a.aux, fp.vidP, fp.eP);
                P::boundary_state_divergence(a.prm, fp.bctag, gP, auxP, fp.n, gM, auxM, a_t);
            }
            // CentralNumericalFluxDivergence  NumericalFluxes.jl:720-730
            const double nh0 = fp.n[0] / 2, nh1 = fp.n[1] / 2, nh2 = fp.n[2] / 2;
#pragma unroll
            for (int s = 0; s < NGL; ++s) {
                const double ldiv = (gP[3 * s] + gM[3 * s]) * nh0 +
                                    (gP[3 * s + 1] + gM[3 * s + 1]) * nh1 +
                                    (gP[3 * s + 2] + gM[3 * s + 2]) * nh2;
                corr[s] = fp.vMI * fp.sM * ldiv;
            }
            vidM = fp.vidM;
            fpair = f / 2;
        }
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        if (fpair == p) {
#pragma unroll
            for (int s = 0; s < NGL; ++s) sA[s * Np + vidM] += corr[s];
        }
        __syncthreads();
    }
    if (tid < Np) {
#pragma unroll
        for (int s = 0; s < NGL; ++s)
            a.hypdiv[tid + (int64_t)Np * (s + (int64_t)NHYP * e)] = sA[s * Np + tid];
    }
    // (the NGL columns this pass writes are all the next one reads of Qhypervisc_div)
    if constexpr (NGL > 0)
        send_nodes<NGL, NGL>(a.h, 0, e, tid, (int)blockDim.x, [&](int s, int n) { return sA[s * Np + n]; });
}

// ---------------------------------------------------------------------------------
// Gradient-of-Laplacian pass: volume_gradients_of_laplacians! (:2525-2824) +
// interface_gradients_of_laplacians! (:2859-3026)
template <class P, int NQ, int NQV = NQ>
__global__ void __launch_bounds__((KDims<NQ, NQV>::NT), CMDG_LAP_MINW) k_gradlap(const PassArgs<P> a)
{
    using KD = KDims<NQ, NQV>;
    const double a_t = a.tptr ? *a.tptr : a.t;  // (uniform: one scalar load)
    constexpr int Np = KD::Np, NS = P::NS, NAUX = P::NAUX, NGL = P::NGL,
                  NHG = 3 * NGL, NHYP = P::NHYP, NG = NGL > 0 ? NGL : 1,
                  NH = NHYP > 0 ? NHYP : 1;
    __shared__ double sD[NQ * NQ + (NQV == NQ ? 0 : NQV * NQV)];
    const double *const sDv = sD + (NQV == NQ ? 0 : NQ * NQ);  // vertical derivative matrix
    __shared__ double sL[NG * Np];
    __shared__ double sA[NH * Np];
    const int tid = threadIdx.x;
    const int64_t e = a.elems[xcd_remap(blockIdx.x, gridDim.x)] - 1;
    if (tid < NQ * NQ) sD[tid] = a.g.D[tid];
    if constexpr (NQV != NQ) {
        if (tid < NQV * NQV) sD[NQ * NQ + tid] = a.g.Dv[tid];
    }
    const bool hz = a.direction != DIR_VERTICAL, vt = a.direction != DIR_HORIZONTAL;
    if (tid < Np) {
        Vec<NGL> l;  // (all loads in flight before the first LDS store, as in k_divgrad)
#pragma unroll
        for (int s = 0; s < NGL; ++s) l[s] = a.hypdiv[tid + (int64_t)Np * (s + (int64_t)NHYP * e)];
#pragma unroll
        for (int s = 0; s < NGL; ++s) sL[s * Np + tid] = l[s];
    }
    __syncthreads();
    if (tid < Np) {
        const int i = tid % NQ, j = (tid / NQ) % NQ, k = tid / (NQ * NQ);
        const double *vg = a.g.vgeo + (int64_t)Np * a.g.nvgeo * e + tid;
        Vec<NS> lQ;
        Vec<NAUX> laux;
        load_state<NS, Np>(lQ, a.Q, tid, e);
        load_state<NAUX, Np>(laux, a.aux, tid, e);
        Vec<NHG> lh, lv;
        lh.negzero();
        lv.negzero();
        if (hz) {
            const double x11 = vg[XI1X1 * Np], x12 = vg[XI1X2 * Np], x13 = vg[XI1X3 * Np];
            const double x21 = vg[XI2X1 * Np], x22 = vg[XI2X2 * Np], x23 = vg[XI2X3 * Np];
#pragma unroll
            for (int s = 0; s < NGL; ++s) {
                double l1 = 0.0, l2 = 0.0;
#pragma unroll
                for (int n = 0; n < NQ; ++n) {
                    l1 += sD[i + NQ * n] * sL[s * Np + n + NQ * (j + NQ * k)];
                    l2 += sD[j + NQ * n] * sL[s * Np + i + NQ * (n + NQ * k)];
                }
                lh[3 * s + 0] = x11 * l1;
                lh[3 * s + 1] = x12 * l1;
                lh[3 * s + 2] = x13 * l1;
                lh[3 * s + 0] += x21 * l2;
                lh[3 * s + 1] += x22 * l2;
                lh[3 * s + 2] += x23 * l2;
            }
        }
        if (vt) {
            const double x31 = vg[XI3X1 * Np], x32 = vg[XI3X2 * Np], x33 = vg[XI3X3 * Np];
#pragma unroll
            for (int s = 0; s < NGL; ++s) {
                double l3 = -0.0;
#pragma unroll
                for (int n = 0; n < NQV; ++n)
                    l3 += sDv[k + NQV * n] * sL[s * Np + i + NQ * (j + NQ * n)];
                lv[3 * s + 0] += x31 * l3;
                lv[3 * s + 1] += x32 * l3;
                lv[3 * s + 2] += x33 * l3;
            }
        }
        Vec<NHYP> h1, h2;
        h1.negzero();
        h2.negzero();
        if (hz) P::post_gradient_laplacian(a.prm, h1, lh, lQ, laux, a_t);
        if (vt) P::post_gradient_laplacian(a.prm, h2, lv, lQ, laux, a_t);
#pragma unroll
        for (int s = 0; s < NHYP; ++s) sA[s * Np + tid] = hz ? (vt ? h1[s] + h2[s] : h1[s]) : h2[s];
    }
    __syncthreads();
    Vec<NHYP> corr;
    int vidM = 0, fpair = -1;
    if (tid < KD::NFT) {
        int f, n;
        KD::face_task(tid, f, n);
        const bool on = f < 4 ? hz : vt;
        if (on) {
            FacePt fp;
            face_setup<NQ, NQV>(a.g, e, tid, f, n, fp);
            Vec<NS> QM, QP;
            Vec<NAUX> auxM, auxP;
            Vec<NGL> lapM, lapP;
            load_state<NS, Np>(QM, a.Q, fp.vidM, e);
            load_state<NAUX, Np>(auxM, a.aux, fp.vidM, e);
            const int gslot = ghost_slot<Np>(a.h, fp.eP, fp.vidP);
            load_plus<NS, Np, NS>(QP, a.Q, a.h.recvQ, gslot, fp.vidP, fp.eP);
            load_state<NAUX, Np>(auxP, a.aux, fp.vidP, fp.eP);
#pragma unroll
            for (int s = 0; s < NGL; ++s) lapM[s] = sL[s * Np + fp.vidM];
            load_plus<NHYP, Np, NGL, NGL>(lapP, a.hypdiv, a.h.recvHD, gslot, fp.vidP, fp.eP);
            if (fp.bctag != 0)  // numerical_boundary_flux_higher_order!  :792-832
                P::boundary_state_higher_order(a.prm, fp.bctag, QP, auxP, lapP, fp.n, QM, auxM,
                                               lapM, a_t);
            // CentralNumericalFluxHigherOrder  NumericalFluxes.jl:768-790
            Vec<NHG> G;
#pragma unroll
            for (int s = 0; s < NGL; ++s)
#pragma unroll
                for (int d = 0; d < 3; ++d) G[d + 3 * s] = fp.n[d] * (lapP[s] - lapM[s]) / 2;
            Vec<NHYP> lh;
            for (int s = 0; s < NHYP; ++s) lh[s] = 0;
            P::post_gradient_laplacian(a.prm, lh, G, QM, auxM, a_t);
#pragma unroll
            for (int s = 0; s < NHYP; ++s) corr[s] = fp.vMI * fp.sM * lh[s];
            vidM = fp.vidM;
            fpair = f / 2;
        }
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        if (fpair == p) {
#pragma unroll
            for (int s = 0; s < NHYP; ++s) sA[s * Np + vidM] += corr[s];
        }
        __syncthreads();
    }
    if constexpr (CMDG_HG_NODE_MAJOR) {
        if constexpr (NHYP > 0) store_node_major<NHG, Np, NHYP>(a.hypgrad, e, tid, (int)blockDim.x, sA);
    } else if (tid < Np) {
#pragma unroll
        for (int s = 0; s < NHYP; ++s)
            a.hypgrad[hg_at<NHG, Np>(tid, s, e)] = sA[s * Np + tid];
    }
    if constexpr (NHYP > 0)
        send_nodes<NHG, NHYP>(a.h, 0, e, tid, (int)blockDim.x, [&](int s, int n) { return sA[s * Np + n]; });
}

// ---------------------------------------------------------------------------------
// kernel_nodal_update_auxiliary_state!  (:1769-1825); elements [e0, e1)
template <class P, int NQ, int NQV = NQ>
__global__ void k_update_aux(typename P::Params prm, const double *Q, double *aux,
                             const uint8_t *activedofs, double t, int64_t e0, int64_t e1)
{
    constexpr int Np = KDims<NQ, NQV>::Np, NS = P::NS, NAUX = P::NAUX;
    const int64_t I = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t e = e0 + I / Np;
    const int n = (int)(I % Np);
    if (e >= e1) return;
    if (activedofs && !activedofs[n + e * Np]) return;
    Vec<NS> lQ;
    Vec<NAUX> laux;
    load_state<NS, Np>(lQ, Q, n, e);
    load_state<NAUX, Np>(laux, aux, n, e);
    P::update_aux(prm, lQ, laux, t);
    if constexpr (P::NUPD > 0) {  // the law names the entries its refresh rewrites
#pragma unroll
        for (int s = 0; s < P::NUPD; ++s)
            aux[n + (int64_t)Np * (P::upd_aux(s) + (int64_t)NAUX * e)] = laux[P::upd_aux(s)];
    } else {
#pragma unroll
        for (int s = 0; s < NAUX; ++s) aux[n + (int64_t)Np * (s + (int64_t)NAUX * e)] = laux[s];
    }
}

// one-time: time-invariant per-node fields the law would otherwise recompute every call
template <class P, int NQ, int NQV = NQ>
__global__ void k_init_derived(typename P::Params prm, const double *aux, double *derived,
                               int64_t nelem)
{
    constexpr int Np = KDims<NQ, NQV>::Np, NAUX = P::NAUX, NDER = P::NDER;
    const int64_t I = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t e = I / Np;
    const int n = (int)(I % Np);
    if (e >= nelem) return;
    Vec<NAUX> laux;
    load_state<NAUX, Np>(laux, aux, n, e);
    Vec<NDER> d;
    P::init_derived(prm, d, laux);
#pragma unroll
    for (int s = 0; s < NDER; ++s) derived[n + (int64_t)Np * (s + (int64_t)NDER * e)] = d[s];
}

// ---------------------------------------------------------------------------------
// kernel_fillsendbuf! / kernel_transferrecvbuf!  MPIStateArrays.jl:837-871
// (nvar = columns per position of the packed buffer = the leading columns of the ncol-column array;
// the reference packs whole arrays, nvar == ncol)
static __global__ void k_fillsendbuf(double *__restrict__ sendbuf, const double *__restrict__ buf,
                              const int64_t *__restrict__ vmapsend, int64_t nvmap, int Np,
                              int nvar, int ncol, int node_major = 0)
{
    const int64_t I = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= nvmap * nvar) return;
    const int64_t i = I / nvar;
    const int s = (int)(I % nvar);
    const int64_t id = vmapsend[i] - 1;
    const int64_t e = id / Np, n = id % Np;
    sendbuf[s + (int64_t)nvar * i] = node_major ? buf[s + (int64_t)ncol * (n + (int64_t)Np * e)]
                                                : buf[n + (int64_t)Np * (s + (int64_t)ncol * e)];
}
static __global__ void k_transferrecvbuf(double *__restrict__ buf, const double *__restrict__ recvbuf,
                                  const int64_t *__restrict__ vmaprecv, int64_t nvmap, int Np,
                                  int nvar, int ncol, int node_major = 0)
{
    const int64_t I = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= nvmap * nvar) return;
    const int64_t i = I / nvar;
    const int s = (int)(I % nvar);
    const int64_t id = vmaprecv[i] - 1;
    const int64_t e = id / Np, n = id % Np;
    buf[node_major ? s + (int64_t)ncol * (n + (int64_t)Np * e) : n + (int64_t)Np * (s + (int64_t)ncol * e)] =
        recvbuf[s + (int64_t)nvar * i];
}

// ---------------------------------------------------------------------------------
// A node-major array of the library (ncol, Np, nelem) into the reference layout (Np, ncol, nelem) of
// a caller that asked for it (create_states.jl:17-26), and back.
static __global__ void k_export_node_major(double *__restrict__ dst, const double *__restrict__ src, int Np,
                                           int ncol, int64_t nelem)
{
    const int64_t I = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= (int64_t)Np * ncol * nelem) return;
    const int64_t e = I / ((int64_t)Np * ncol);
    const int r = (int)(I - e * Np * ncol), s = r / Np, n = r - s * Np;
    dst[I] = src[s + (int64_t)ncol * (n + (int64_t)Np * e)];
}
static __global__ void k_import_node_major(double *__restrict__ dst, const double *__restrict__ src, int Np,
                                           int ncol, int64_t nelem)
{
    const int64_t I = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (I >= (int64_t)Np * ncol * nelem) return;
    const int64_t e = I / ((int64_t)Np * ncol);
    const int r = (int)(I - e * Np * ncol), n = r / ncol, s = r - n * ncol;
    dst[I] = src[n + (int64_t)Np * (s + (int64_t)ncol * e)];
}

// ---------------------------------------------------------------------------------
// local part of norm / euclidean_distance (MPIStateArrays.jl:583-644): per-block
// partial sums in a fixed order (deterministic), finished on the host.
static __global__ void k_wsum2(const double *__restrict__ A, const double *__restrict__ B,
                        const double *__restrict__ vgeo, int nvgeo, int Np, int nvar,
                        int64_t nreal, int weighted, double *__restrict__ partial)
{
    __shared__ double sh[256];
    const int64_t total = (int64_t)Np * nvar * nreal;
    double acc = 0.0;
    for (int64_t I = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; I < total;
         I += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = I / ((int64_t)Np * nvar);
        const int n = (int)(I % Np);
        double d = A[I];
        if (B) d -= B[I];
        const double w = weighted ? vgeo[n + (int64_t)Np * (VM + (int64_t)nvgeo * e)] : 1.0;
        acc += w * d * d;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// ---- courant(local_courant, dg, m, Q, dt, t, direction)  SpaceDiscretization.jl:307-365 ----
// One block per real element: node coordinates staged in LDS, the minimum neighbour distance
// (kernel_min_neighbor_distance!, Grids.jl:1228-1333) and the law's local Courant number
// (kernel_local_courant!, DGModel_kernels.jl:3028-3096) evaluated per node, then a block-wide
// extremum.  MODE 0: out[e] = min distance of the element, MODE 1: out[e] = max Courant.
template <class P, int NQ, int NQV, int MODE>
__global__ __launch_bounds__((KDims<NQ, NQV>::Np <= 128 ? 128 : 256)) void k_courant(
    typename P::Params prm, const double *__restrict__ vgeo, int nvgeo,
    const double *__restrict__ Q, const double *__restrict__ aux, const double *__restrict__ gf,
    int kind, double dt, double t, int direction, double *__restrict__ out)
{
    constexpr int Np = KDims<NQ, NQV>::Np, NT = Np <= 128 ? 128 : 256;
    __shared__ double sx[3][Np], sred[NT];
    const int tid = threadIdx.x;
    const int64_t e = blockIdx.x;
    constexpr int X1 = 12;  // _x1 (Grids.jl:76-92), 0-based column
    for (int n = tid; n < Np; n += NT)
#pragma unroll
        for (int d = 0; d < 3; ++d) sx[d][n] = vgeo[n + (int64_t)Np * (X1 + d + (int64_t)nvgeo * e)];
    __syncthreads();
    double val = MODE == 0 ? INFINITY : -INFINITY;
    for (int n = tid; n < Np; n += NT) {
        const int i = n % NQ, j = (n / NQ) % NQ, k = n / (NQ * NQ);
        double md = INFINITY;
        auto dist = [&](int m) {
            const double d0 = sx[0][n] - sx[0][m], d1 = sx[1][n] - sx[1][m], d2 = sx[2][n] - sx[2][m];
            return sqrt(d0 * d0 + d1 * d1 + d2 * d2);
        };
        if (direction != DIR_VERTICAL) {
            if (i > 0) md = fmin(md, dist(n - 1));
            if (i < NQ - 1) md = fmin(md, dist(n + 1));
            if (j > 0) md = fmin(md, dist(n - NQ));
            if (j < NQ - 1) md = fmin(md, dist(n + NQ));
        }
        if (direction != DIR_HORIZONTAL) {
            if (k > 0) md = fmin(md, dist(n - NQ * NQ));
            if (k < NQV - 1) md = fmin(md, dist(n + NQ * NQ));
        }
        if constexpr (MODE == 0) {
            val = fmin(val, md);
        } else {
            Vec<P::NS> lQ;
            Vec<P::NAUX> lA;
            Vec<P::NGF> lG;
            load_state<P::NS, Np>(lQ, Q, n, e);
            load_state<P::NAUX, Np>(lA, aux, n, e);
            if constexpr (P::NGF > 0) load_gf<P, Np>(lG, gf, n, e);
            val = fmax(val, P::courant(prm, kind, lQ, lA, lG, md, dt, t, direction));
        }
    }
    sred[tid] = val;
    __syncthreads();
    for (int h = NT / 2; h > 0; h >>= 1) {
        if (tid < h) sred[tid] = MODE == 0 ? fmin(sred[tid], sred[tid + h]) : fmax(sred[tid], sred[tid + h]);
        __syncthreads();
    }
    if (tid == 0) out[e] = sred[0];
}

// extremum of n values into out[0] (one block)
static __global__ void k_extremum(const double *__restrict__ v, int64_t n, int is_min,
                                  double *__restrict__ out)
{
    __shared__ double s[1024];
    double a = is_min ? INFINITY : -INFINITY;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) a = is_min ? fmin(a, v[i]) : fmax(a, v[i]);
    s[threadIdx.x] = a;
    __syncthreads();
    for (int h = blockDim.x / 2; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h)
            s[threadIdx.x] = is_min ? fmin(s[threadIdx.x], s[threadIdx.x + h])
                                    : fmax(s[threadIdx.x], s[threadIdx.x + h]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = s[0];
}

}  // namespace cmdg
