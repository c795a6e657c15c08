// Engine instantiations for the AdvectionDiffusion law (physics_advdiff.h).
#include "engine.h"
#include "physics_advdiff.h"

namespace cmdg {

int counts_advdiff(const int32_t *ip, int32_t out[6])
{
    if (ip[0] != 1) return CMDG_ERR_UNSUPPORTED;  // num_equations
    const bool adv = ip[1], diff = ip[2], hyp = ip[3];
    out[0] = 1;
    out[1] = 3 + (adv ? 3 : 0) + (diff ? 9 : 0) + (hyp ? 9 : 0);
    out[2] = (diff || hyp) ? 1 : 0;
    out[3] = diff ? 3 : 0;
    out[4] = hyp ? 1 : 0;
    out[5] = hyp ? 3 : 0;
    return CMDG_OK;
}

template <int NQ>
static EngineBase *pick(const cmdg_desc *d, std::string &err)
{
    const bool adv = d->iparam[1], diff = d->iparam[2], hyp = d->iparam[3];
    if (adv && diff && !hyp) return make_engine<AdvDiff<true, true, false>, NQ>(d);
    if (!adv && !diff && hyp) return make_engine<AdvDiff<false, false, true>, NQ>(d);
    if (adv && !diff && !hyp) return make_engine<AdvDiff<true, false, false>, NQ>(d);
    if (!adv && diff && !hyp) return make_engine<AdvDiff<false, true, false>, NQ>(d);
    if (!adv && diff && hyp) return make_engine<AdvDiff<false, true, true>, NQ>(d);
    err = "AdvectionDiffusion: this advection/diffusion/hyperdiffusion combination is not compiled in";
    return nullptr;
}

EngineBase *make_engine_advdiff(const cmdg_desc *d, std::string &err)
{
    if (d->iparam[0] != 1) {
        err = "AdvectionDiffusion: num_equations != 1 is not compiled in";
        return nullptr;
    }
    if (d->N[2] != d->N[0]) {
        // polynomialorder = (N_h, N_v): the pairs of variable_degree_advection_diffusion.jl:300
        const bool adv = d->iparam[1], diff = d->iparam[2], hyp = d->iparam[3];
        if (adv && diff && !hyp) {
            if (d->N[0] == 4 && d->N[2] == 2) return make_engine<AdvDiff<true, true, false>, 5, 3>(d);
            if (d->N[0] == 2 && d->N[2] == 4) return make_engine<AdvDiff<true, true, false>, 3, 5>(d);
        }
        err = "AdvectionDiffusion: mixed polynomial orders compiled in are (4,2) and (2,4), "
              "advection + diffusion";
        return nullptr;
    }
    switch (d->N[0]) {  // NQ = N + 1 is a template parameter of every kernel
    case 1: return pick<2>(d, err);
    case 2: return pick<3>(d, err);
    case 3: return pick<4>(d, err);
    case 4: return pick<5>(d, err);
    case 5: return pick<6>(d, err);
    case 6: return pick<7>(d, err);
    case 7: return pick<8>(d, err);
    default:
        err = "AdvectionDiffusion: polynomial order not compiled in (have N = 1..7)";
        return nullptr;
    }
}

}  // namespace cmdg
