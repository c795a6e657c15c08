// Engine instantiations for the hydrostatic Boussinesq ocean law (physics_ocean.h).
#include "engine.h"
#include "physics_ocean.h"

namespace cmdg {

int counts_ocean(const int32_t *, int32_t out[6])
{
    out[0] = HydroBoussinesq::NS;
    out[1] = HydroBoussinesq::NAUX;
    out[2] = HydroBoussinesq::NGRAD;
    out[3] = HydroBoussinesq::NGF;
    out[4] = 0;
    out[5] = 0;
    return CMDG_OK;
}

EngineBase *make_engine_ocean(const cmdg_desc *d, std::string &err)
{
    switch (d->N[0]) {
    case 2: return make_engine<HydroBoussinesq, 3>(d);
    case 3: return make_engine<HydroBoussinesq, 4>(d);
    case 4: return make_engine<HydroBoussinesq, 5>(d);
    case 5: return make_engine<HydroBoussinesq, 6>(d);
    default:
        err = "HydrostaticBoussinesq: polynomial order not compiled in (have N = 2..5)";
        return nullptr;
    }
}

}  // namespace cmdg
