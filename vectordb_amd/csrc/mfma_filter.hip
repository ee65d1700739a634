// placeholder until the fp16-MFMA filter engine lands (next milestone)
#include "index.hpp"
namespace eps {
struct HalfMirror {};
void half_mirror_free(HalfMirror* m) { delete m; }
bool flat_mfma_supported(const Index&, int64_t, int) { return false; }
int32_t flat_mfma_search(Index& ix, const float*, int64_t, int, u64*) {
  return ix.fail(EPS_NOT_IMPLEMENTED_ERROR, "MFMA flat engine not built");
}
}  // namespace eps
