// placeholder until the device graph build lands (next milestone)
#include "index.hpp"
namespace eps {
int32_t graph_build(Index& ix, int64_t, const eps_build_params&) {
  return ix.fail(EPS_NOT_IMPLEMENTED_ERROR, "device graph build not built yet");
}
}  // namespace eps
