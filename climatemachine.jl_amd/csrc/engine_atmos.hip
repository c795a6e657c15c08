// Engine instantiations for the dry atmosphere law (placeholder until physics_atmos.h lands).
#include "engine.h"

namespace cmdg {
int counts_atmos(const int32_t *, int32_t *) { return CMDG_ERR_UNSUPPORTED; }
EngineBase *make_engine_atmos(const cmdg_desc *, std::string &err)
{
    err = "dry atmosphere law is not compiled in";
    return nullptr;
}
}  // namespace cmdg
