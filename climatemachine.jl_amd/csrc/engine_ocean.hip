// Engine instantiations for the hydrostatic Boussinesq ocean law (physics_ocean.h) and the
// PressureGradientModel used by the reference-state initialisation (physics_pgrad.h).
#include "engine.h"
#include "physics_ocean.h"
#include "physics_ocean01.h"
#include "physics_pgrad.h"
#include "physics_sw.h"

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

int counts_sw(const int32_t *ip, int32_t out[6])
{
    out[0] = 3;
    out[1] = 5;
    out[2] = ShallowWater::NGRAD;
    out[3] = ShallowWater::NGF;
    out[4] = out[5] = 0;
    (void)ip;
    return CMDG_OK;
}

EngineBase *make_engine_sw(const cmdg_desc *d, std::string &err)
{
    if (d->iparam[1] != 0) {
        err = "ShallowWaterModel: LinearDrag is not compiled in (ConstantViscosity only)";
        return nullptr;
    }
    if (d->N[2] != d->N[0]) {
        // the one-layer extrusion of the 2-D grid needs no resolution along the extrusion:
        // two nodes (N_v = 1) carry the same 2-D arithmetic at 2/5 of the work
        if (d->N[0] == 4 && d->N[2] == 1) return make_engine<ShallowWater, 5, 2>(d);
        err = "ShallowWaterModel: mixed polynomial orders compiled in: (4, 1)";
        return nullptr;
    }
    switch (d->N[0]) {
    case 2: return make_engine<ShallowWater, 3>(d);
    case 3: return make_engine<ShallowWater, 4>(d);
    case 4: return make_engine<ShallowWater, 5>(d);
    case 5: return make_engine<ShallowWater, 6>(d);
    default:
        err = "ShallowWaterModel: polynomial order not compiled in (have N = 2..5)";
        return nullptr;
    }
}

// src/Ocean/SplitExplicit01: OceanModel, Continuity3dModel, BarotropicModel (N = 4, the order
// of its reference tests; the barotropic model also with two nodes along the extrusion)
template <class P>
static void se01_counts(int32_t out[6])
{
    out[0] = P::NS;
    out[1] = P::NAUX;
    out[2] = P::NGRAD;
    out[3] = P::NGF;
    out[4] = out[5] = 0;
}
int counts_se01(int32_t physics_id, int32_t out[6])
{
    switch (physics_id) {
    case CMDG_PHYSICS_OCEAN_SE01: se01_counts<OceanSE01>(out); return CMDG_OK;
    case CMDG_PHYSICS_CONTINUITY3D_SE01: se01_counts<Continuity3dSE01>(out); return CMDG_OK;
    case CMDG_PHYSICS_BAROTROPIC_SE01: se01_counts<BarotropicSE01>(out); return CMDG_OK;
    default: return CMDG_ERR_UNSUPPORTED;
    }
}

EngineBase *make_engine_se01(const cmdg_desc *d, std::string &err)
{
    if (d->N[0] != 4) {
        err = "SplitExplicit01 laws: polynomial order not compiled in (have N = 4)";
        return nullptr;
    }
    if (d->physics_id == CMDG_PHYSICS_BAROTROPIC_SE01) {
        if (d->N[2] == 1) return make_engine<BarotropicSE01, 5, 2>(d);
        if (d->N[2] == 4) return make_engine<BarotropicSE01, 5>(d);
        err = "BarotropicModel: extrusion order 1 or 4";
        return nullptr;
    }
    if (d->N[2] != 4) {
        err = "SplitExplicit01 laws: one polynomial order in all directions";
        return nullptr;
    }
    if (d->physics_id == CMDG_PHYSICS_OCEAN_SE01) return make_engine<OceanSE01, 5>(d);
    return make_engine<Continuity3dSE01, 5>(d);
}

int counts_pgrad(const int32_t *, int32_t out[6])
{
    out[0] = 3;
    out[1] = 1;
    out[2] = out[3] = out[4] = out[5] = 0;
    return CMDG_OK;
}

EngineBase *make_engine_pgrad(const cmdg_desc *d, std::string &err)
{
    switch (d->N[0]) {
    case 1: return make_engine<PressureGradient, 2>(d);
    case 2: return make_engine<PressureGradient, 3>(d);
    case 3: return make_engine<PressureGradient, 4>(d);
    case 4: return make_engine<PressureGradient, 5>(d);
    case 5: return make_engine<PressureGradient, 6>(d);
    case 6: return make_engine<PressureGradient, 7>(d);
    case 7: return make_engine<PressureGradient, 8>(d);
    default:
        err = "PressureGradientModel: polynomial order not compiled in (have N = 1..7)";
        return nullptr;
    }
}

}  // namespace cmdg
