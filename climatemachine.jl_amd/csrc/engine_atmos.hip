// Engine instantiations for the dry atmosphere law (physics_atmos.h).
#include "engine.h"
#include "physics_atmos.h"

namespace cmdg {

int counts_atmos(const int32_t *ip, int32_t out[6])
{
    const bool orient = ip[0] != 0, ref = ip[1] != 0, hyp = ip[4] != 0, smag = ip[14] == 1;
    out[0] = 5;
    out[1] = 3 + (orient ? 4 : 0) + (ref ? 7 : 0) + (smag ? 1 : 0) + (hyp ? 1 : 0) + 2;
    out[2] = 4 + (smag ? 1 : 0) + (hyp ? 4 : 0);
    out[3] = 9 + (smag ? 1 : 0);
    out[4] = hyp ? 4 : 0;
    out[5] = hyp ? 12 : 0;
    return CMDG_OK;
}

template <int NQ>
static EngineBase *pick(const cmdg_desc *d, std::string &err)
{
    const bool orient = d->iparam[0] != 0, ref = d->iparam[1] != 0, hyp = d->iparam[4] != 0;
    bool mms = (d->iparam[5] & 8) != 0;
    for (int i = 0; i < d->iparam[6] && i < 7; ++i) mms = mms || d->iparam[7 + i] == 2;
    if (mms && (orient || ref || hyp || d->iparam[14] != 0)) {
        err = "DryAtmos: MMSSource / InitStateBC are compiled for NoOrientation, NoReferenceState, "
              "constant viscosity only";
        return nullptr;
    }
    if (d->nf_first >= NF_ROE) {  // the law's own Roe / HLLC methods
        if constexpr (NQ == 5) {
            if (!orient && !ref && !hyp && d->iparam[14] == 0 && !mms)
                return make_engine<DryAtmos<false, false, false, false, true>, NQ>(d);
            if (orient && ref && !hyp && d->iparam[14] == 0)
                return make_engine<DryAtmos<true, true, false, false, true>, NQ>(d);
        }
        err = "DryAtmos: Roe / HLLC / LMARS numerical fluxes are compiled for N = 4, constant "
              "viscosity, without orientation and reference state or with both";
        return nullptr;
    }
    if (d->iparam[14] == 1) {  // SmagorinskyLilly (AtmosLES configurations)
        if (orient && ref && !hyp) return make_engine<DryAtmos<true, true, false, true>, NQ>(d);
        err = "DryAtmos: SmagorinskyLilly is compiled with orientation + reference state, no hyperdiffusion";
        return nullptr;
    }
    if (d->iparam[14] != 0) {
        err = "DryAtmos: unknown turbulence closure";
        return nullptr;
    }
    if (!orient && !ref && !hyp) return make_engine<DryAtmos<false, false, false>, NQ>(d);
    if (orient && ref && hyp) return make_engine<DryAtmos<true, true, true>, NQ>(d);
    if (orient && ref && !hyp) return make_engine<DryAtmos<true, true, false>, NQ>(d);
    if (orient && !ref && !hyp) return make_engine<DryAtmos<true, false, false>, NQ>(d);
    err = "DryAtmos: this orientation/ref-state/hyperdiffusion combination is not compiled in";
    return nullptr;
}

EngineBase *make_engine_atmos(const cmdg_desc *d, std::string &err)
{
    if ((d->iparam[1] != 0 || d->iparam[4] != 0) && d->iparam[0] == 0) {
        err = "DryAtmos: reference state / hyperdiffusion need an orientation";
        return nullptr;
    }
    switch (d->N[0]) {  // element-per-workgroup kernels: one element's working set lives in LDS
                        // (N = 6 with hyperdiffusion: 90 KB of the CU's 160 KB)
    case 2: return pick<3>(d, err);
    case 3: return pick<4>(d, err);
    case 4: return pick<5>(d, err);
    case 5: return pick<6>(d, err);
    case 6: return pick<7>(d, err);
    default:
        err = "DryAtmos: polynomial order not compiled in (have N = 2..6)";
        return nullptr;
    }
}

}  // namespace cmdg
