// Engine instantiations for the moist LES law (physics_moist.h).
#include "engine.h"
#include "physics_moist.h"

namespace cmdg {

int counts_moist(const int32_t *ip, int32_t out[6])
{
    out[0] = 6;
    out[1] = 19;
    out[2] = 6;
    out[3] = 3 + (ip[0] == 2 ? 10 : 7) + 3;
    out[4] = out[5] = 0;
    return CMDG_OK;
}

template <int NQ>
static EngineBase *pick(const cmdg_desc *d, std::string &err)
{
    if (d->nf_first >= NF_ROE_MOIST) {  // the law's own RoeNumericalFluxMoist
        if constexpr (NQ == 5) {
            if (d->iparam[0] == 0) return make_engine<MoistAtmos<0, true>, NQ>(d);
        }
        err = "MoistAtmos: RoeNumericalFluxMoist is compiled for N = 4 with constant viscosity";
        return nullptr;
    }
    switch (d->iparam[0]) {
    case 0: return make_engine<MoistAtmos<0>, NQ>(d);
    case 1: return make_engine<MoistAtmos<1>, NQ>(d);
    case 2: return make_engine<MoistAtmos<2>, NQ>(d);
    default: err = "MoistAtmos: unknown turbulence closure"; return nullptr;
    }
}

EngineBase *make_engine_moist(const cmdg_desc *d, std::string &err)
{
    switch (d->N[0]) {
    case 4: return pick<5>(d, err);
    case 6: return pick<7>(d, err);  // BASELINE configs[3]
    default: err = "MoistAtmos: polynomial orders compiled in: N = 4, 6"; return nullptr;
    }
}

}  // namespace cmdg
