// ORACLE BUILD SHIM (test infrastructure, not product code).
// Stands in for the reference's HTTP embedding client, which needs oatpp + libcurl
// (absent here). Every method reports NOT_IMPLEMENTED; the dense ANN path never calls it.
// Method signatures follow /root/reference/engine/services/embedding_service.hpp:87-110.
#pragma once
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "db/vector.hpp"
#include "logger/logger.hpp"
#include "utils/json.hpp"
#include "utils/status.hpp"

namespace vectordb {
namespace engine {

struct EmbeddingModel {
  std::string model;
  size_t dim;
  bool dense;
  bool dimensionReduction;
};

class EmbeddingService {
 public:
  explicit EmbeddingService(const std::string&) {}
  Status getSupportedModels(std::vector<EmbeddingModel>&) {
    return Status(NOT_IMPLEMENTED_ERROR, "embedding service stubbed out in oracle build");
  }
  Status denseEmbedDocuments(const std::string&, VariableLenAttrColumnContainer&, float*, size_t, size_t, size_t,
                             std::unordered_map<std::string, std::string>&, bool) {
    return Status(NOT_IMPLEMENTED_ERROR, "embedding service stubbed out in oracle build");
  }
  Status denseEmbedQuery(const std::string&, const std::string&, std::vector<engine::DenseVectorElement>&, size_t,
                         std::unordered_map<std::string, std::string>&, bool) {
    return Status(NOT_IMPLEMENTED_ERROR, "embedding service stubbed out in oracle build");
  }
};

}  // namespace engine
}  // namespace vectordb
