/* ORACLE -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C + OpenMP) of the reference's DG kernels, in the
 * reference's own launch and summation order.  Only tests/, the smoke check in
 * __graft_entry__.py and bench.py's cpu_baseline leg may build or call this; the
 * product (climatemachine.jl_amd/, libcmdg.so) never does.
 *
 * Pinned against the reference's own golden values (tests/golden/, e.g.
 * test/Numerics/DGMethods/advection_diffusion/pseudo1D_advection_diffusion.jl:254).
 *
 * Array layouts are the reference's column-major ones:
 *   state (Np, nstate, nelem)      vgeo (Np, nvgeo, nelem)   sgeo (5, Nfp, nface, nelem)
 *   vmapM/vmapP (Nfp, nface, nelem) int64, 1-based          elemtobndy (nface, nelem) int64
 *   Grad-type locals are 3 x nvar column-major: g[d + 3*s]   (vars_wrappers.jl:52)
 */
#ifndef DG_ORACLE_H
#define DG_ORACLE_H
#include <stdint.h>

#define ORC_MAXS 32 /* max vars of any one state type held in a local array */

enum { ORC_EVERY = 0, ORC_HORIZONTAL = 1, ORC_VERTICAL = 2 };
enum { ORC_NF_RUSANOV = 0, ORC_NF_CENTRAL = 1, ORC_NF_ROE = 2, ORC_NF_HLLC = 3, ORC_NF_LMARS = 4 };
/* which numerical flux asks for the boundary state (dispatch of boundary_state!) */
enum { ORC_BS_FIRST = 0, ORC_BS_GRADIENT = 1 };

typedef struct orc_physics {
    int ns, naux, ngrad, ngf, ngl, nhyp;
    int hv_indexmap[ORC_MAXS]; /* hyperdiff_indexmap: ngl entries, 0-based into Gradient vars */
    int nf_first;              /* ORC_NF_* */
    const void *p;             /* parameter block of the law */
    /* pointwise balance-law callbacks (BalanceLaws/interface.jl:37-464) */
    void (*flux_first_order)(const void *p, double *F, const double *Q, const double *aux,
                             double t, int dir);
    void (*flux_second_order)(const void *p, double *F, const double *Q, const double *gf,
                              const double *hyp, const double *aux, double t);
    void (*source)(const void *p, double *S, const double *Q, const double *gf,
                   const double *aux, double t, int dir);
    void (*gradient_argument)(const void *p, double *G, const double *Q, const double *aux,
                              double t);
    void (*gradient_flux)(const void *p, double *gf, const double *gradG, const double *Q,
                          const double *aux, double t);
    void (*post_gradient_laplacian)(const void *p, double *hyp, const double *gradlap,
                                    const double *Q, const double *aux, double t);
    /* ws has ns entries (a scalar wavespeed is broadcast) */
    void (*wavespeed)(const void *p, double *ws, const double *n, const double *Q,
                      const double *aux, double t, int facedir);
    /* boundary_state! for first-order / gradient numerical fluxes: fills QP, auxP */
    void (*boundary_state)(const void *p, int kind, int bctag, double *QP, double *auxP,
                           const double *n, const double *QM, const double *auxM, double t,
                           const double *Q1, const double *aux1);
    /* boundary_flux_second_order!: fills F (3 x ns) */
    void (*boundary_flux_second_order)(const void *p, int bctag, double *F, double *QP,
                                       double *gfP, double *hypP, double *auxP, const double *n,
                                       const double *QM, const double *gfM, const double *hypM,
                                       const double *auxM, double t, const double *Q1,
                                       const double *gf1, const double *aux1);
    void (*boundary_state_divergence)(const void *p, int bctag, double *gradP, double *auxP,
                                      const double *n, const double *gradM, const double *auxM,
                                      double t);
    void (*boundary_state_higher_order)(const void *p, int bctag, double *QP, double *auxP,
                                        double *lapP, const double *n, const double *QM,
                                        const double *auxM, const double *lapM, double t);
    /* nodal update_auxiliary_state! hook (NULL: the law's hook returns false) */
    void (*update_aux)(const void *p, const double *Q, double *aux, double t);
    /* local_courant functions of the law (src/Atmos/Model/courant.jl): kind 0 advective,
     * 1 nondiffusive, 2 diffusive; NULL when the law defines none */
    double (*courant)(const void *p, int kind, const double *Q, const double *aux,
                      const double *gf, double dx, double dt, double t, int direction);
    /* update_penalty!(::RusanovNumericalFlux, law, n, lambda, penalty, Q-, A-, Q+, A+, t)
     * (NumericalFluxes.jl:222, 266-279); NULL = the default no-op */
    void (*update_penalty)(const void *p, double *penalty, const double *n, const double *QM,
                           const double *QP);
    /* law-specific numerical_flux_first_order! methods (ORC_NF_ROE, ORC_NF_HLLC,
     * ORC_NF_LMARS: src/Atmos/Model/AtmosModel.jl:1006-1130, :1154-1276, :1515-1600); adds the normal flux to fluxn */
    void (*numerical_flux_law)(const void *p, int nf, double *fluxn, const double *n,
                               const double *QM, const double *auxM, const double *QP,
                               const double *auxP, double t, int facedir);
} orc_physics;

typedef struct orc_grid {
    int dim;          /* 3 only */
    int Nq[3];        /* Nq[0] == Nq[1] */
    int Np, Nfp, nface, nvgeo;
    int64_t nelem;    /* real + ghost */
    int64_t nreal;
    const double *vgeo, *sgeo;
    const int64_t *vmapM, *vmapP, *elemtobndy;
    const double *D[3]; /* (Nq, Nq) column-major: D[i + Nq*n] = D_{i n} */
} orc_grid;

#ifdef __cplusplus
extern "C" {
#endif
/* SpaceDiscretization.jl:1135-1199 -> DGModel_kernels.jl:64-548.  `direction` is the
 * kernel variant (ORC_HORIZONTAL = the generic kernel called with
 * HorizontalDirection(), ORC_VERTICAL); model_dir = dg.direction. */
void orc_volume_tendency(const orc_physics *ph, const orc_grid *g, int model_dir, int direction,
                         double *tendency, const double *Q, const double *gf, const double *hyp,
                         const double *aux, double t, double alpha, double beta, int add_source);
/* DGModel_kernels.jl:588-901; elems: 1-based list */
void orc_interface_tendency(const orc_physics *ph, const orc_grid *g, int direction,
                            double *tendency, const double *Q, const double *gf,
                            const double *hyp, const double *aux, double t, const int64_t *elems,
                            int64_t nelems, double alpha);
/* DGModel_kernels.jl:934-1328 */
void orc_volume_gradients(const orc_physics *ph, const orc_grid *g, int direction,
                          const double *Q, double *gf, double *hypgrad, const double *aux,
                          double t, int increment);
/* DGModel_kernels.jl:1365-1651 */
void orc_interface_gradients(const orc_physics *ph, const orc_grid *g, int direction,
                             const double *Q, double *gf, double *hypgrad, const double *aux,
                             double t, const int64_t *elems, int64_t nelems);
/* DGModel_kernels.jl:2132-2329 */
void orc_volume_divergence_of_gradients(const orc_physics *ph, const orc_grid *g, int direction,
                                        const double *hypgrad, double *hypdiv, int increment);
/* DGModel_kernels.jl:2360-2494 */
void orc_interface_divergence_of_gradients(const orc_physics *ph, const orc_grid *g,
                                           int direction, const double *hypgrad, double *hypdiv,
                                           const double *aux, double t, const int64_t *elems,
                                           int64_t nelems);
/* DGModel_kernels.jl:2525-2824 */
void orc_volume_gradients_of_laplacians(const orc_physics *ph, const orc_grid *g, int direction,
                                        double *hypgrad, const double *hypdiv, const double *Q,
                                        const double *aux, double t, int increment);
/* DGModel_kernels.jl:2859-3026 */
void orc_interface_gradients_of_laplacians(const orc_physics *ph, const orc_grid *g,
                                           int direction, double *hypgrad, const double *hypdiv,
                                           const double *Q, const double *aux, double t,
                                           const int64_t *elems, int64_t nelems);
/* DGModel_kernels.jl:1769-1825; elems = [e0, e1) 0-based range */
void orc_update_auxiliary_state(const orc_physics *ph, const orc_grid *g, const double *Q,
                                double *aux, double t, int64_t e0, int64_t e1,
                                const uint8_t *activedofs);
/* LowStorageRungeKuttaMethod.jl:146-158 */
void orc_lsrk_update(double *dQ, double *Q, double rka, double rkb, double dt, int64_t n);
/* MPIStateArrays.jl:837-871 (pack / unpack of face nodes; vmap 1-based) */
void orc_fillsendbuf(double *sendbuf, const double *buf, const int64_t *vmapsend, int64_t nvmap,
                     int Np, int nvar);
void orc_transferrecvbuf(double *buf, const double *recvbuf, const int64_t *vmaprecv,
                         int64_t nvmap, int Np, int nvar);
/* Grids.jl:1228-1333 kernel_min_neighbor_distance!: out (Np, nreal) */
void orc_min_neighbor_distance(const orc_grid *g, int direction, double *out);
/* DGModel_kernels.jl:3028-3096 kernel_local_courant!: pointwise (Np, nreal) holds dx on
 * entry and the local Courant number on return */
void orc_local_courant(const orc_physics *ph, const orc_grid *g, int kind, double *pointwise,
                       const double *Q, const double *aux, const double *gf, double dt,
                       double simtime, int direction);
/* ---- column (stack) integrals: DGModel_kernels.jl:1903-2104 ------------------------- */
/* integral_load/set_auxiliary_state!, reverse_integral_load/set_auxiliary_state! of a law
 * (BalanceLaws/interface.jl) */
typedef struct orc_integral_law {
    int nout, nrout; /* UpwardIntegrals / DownwardIntegrals variables */
    int ns, naux;
    const void *p;
    void (*load)(const void *p, double *integrand, const double *Q, const double *aux);
    void (*set)(const void *p, double *aux, const double *integral);
    void (*rload)(const void *p, double *integral, const double *Q, const double *aux);
    void (*rset)(const void *p, double *aux, const double *integral);
} orc_integral_law;
/* kernel_indefinite_stack_integral! over horizontal elements [h0, h1) (0-based); Imat is the
 * (Nq3, Nq3) column-major grid.Imat[dim]; JcV = 0-based vgeo column of _JcV */
void orc_indefinite_stack_integral(const orc_integral_law *law, const orc_grid *g, int nvertelem,
                                   const double *Q, double *aux, const double *Imat, int JcV,
                                   int64_t h0, int64_t h1);
/* kernel_reverse_indefinite_stack_integral! (Nq3 > 1) */
void orc_reverse_indefinite_stack_integral(const orc_integral_law *law, const orc_grid *g,
                                           int nvertelem, const double *Q, double *aux,
                                           int64_t h0, int64_t h1);
/* the reference's IntegralTestModel{3} (test/Numerics/DGMethods/integral_test.jl:37-140) and
 * the field-combination law that mirrors cmdg_stack_integral_desc */
orc_integral_law *orc_integral_test_law(void);
orc_integral_law *orc_integral_fields_law(int nout, const int *src_is_state, const int *src_col,
                                          const double *scale, const int *dst_col,
                                          const int *rsrc_col, const int *rdst_col, int ns,
                                          int naux);
void orc_integral_law_free(orc_integral_law *law);
/* ---- element filters (filter_oracle.c) ------------------------------------------- */
/* filter target: kind 0 FilterIndices (idx 1-based), 1 AtmosFilterPerturbations,
 * 2 AtmosSpecificFilterPerturbations (aux_ref_*: 0-based aux columns of ref_state.rho, .rho e) */
typedef struct orc_filter_target {
    int kind, nfs;
    int idx[ORC_MAXS];
    int aux_ref_rho, aux_ref_rhoe;
} orc_filter_target;
/* Filters.jl:651-794; `direction` is the kernel's direction argument */
void orc_apply_filter(int dim, const int *Nq, int direction, double *Q, int nstate,
                      const double *aux, int naux, const orc_filter_target *tg, const double *F,
                      int64_t nrealelem);
/* Filters.jl:796-884 */
void orc_apply_tmar_filter(int dim, const int *Nq, double *Q, int nstate,
                           const orc_filter_target *tg, const double *vgeo, int nvgeo, int Mcol,
                           int64_t nrealelem);
/* Filters.jl:893-1071 */
void orc_apply_mp_filter(int dim, const int *Nq, int direction, double *Q, int nstate,
                         const double *aux, int naux, const orc_filter_target *tg,
                         const double *F, const double *vgeo, int nvgeo, int Mcol,
                         int64_t nrealelem);
void orc_set_num_threads(int n);
int orc_get_max_threads(void);
#ifdef __cplusplus
}
#endif
#endif
