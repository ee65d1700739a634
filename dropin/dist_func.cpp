// Drop-in GetDistFunc (reference: engine/db/index/index.cpp:10-35).  TableMVP asks for a DistFunc per
// (field type, metric) and hands it to the executor constructor (table_mvp.cpp:70-82); on the device the metric is a
// kernel parameter, so the function pointers returned here are identity tags the executor maps back to EPS_METRIC_*.
// They are also correct host implementations of the three metrics (squared L2, 1 - dot, -dot) should a caller invoke
// them directly; they are never on the search path.
#include "dist_func.hpp"

#include "db/index/index.hpp"
#include "epsilla_gfx950.h"

namespace epsdrop {
static float Dot(const float* x, const float* y, size_t d) {
  float s = 0.f;
  for (size_t i = 0; i < d; ++i) s += x[i] * y[i];
  return s;
}
float TagL2Sqr(const void* a, const void* b, const void* dim_ptr) {
  const float *x = static_cast<const float*>(a), *y = static_cast<const float*>(b);
  const size_t d = *static_cast<const size_t*>(dim_ptr);
  float s = 0.f;
  for (size_t i = 0; i < d; ++i) {
    const float t = x[i] - y[i];
    s += t * t;
  }
  return s;
}
float TagCosineDistance(const void* a, const void* b, const void* dim_ptr) {
  return 1 - 1.0f * Dot(static_cast<const float*>(a), static_cast<const float*>(b), *static_cast<const size_t*>(dim_ptr));
}
float TagInnerProduct(const void* a, const void* b, const void* dim_ptr) {
  return -Dot(static_cast<const float*>(a), static_cast<const float*>(b), *static_cast<const size_t*>(dim_ptr));
}
int MetricOfDistFunc(const void* fn) {
  if (fn == reinterpret_cast<const void*>(&TagL2Sqr)) return EPS_METRIC_EUCLIDEAN;
  if (fn == reinterpret_cast<const void*>(&TagCosineDistance)) return EPS_METRIC_COSINE;
  if (fn == reinterpret_cast<const void*>(&TagInnerProduct)) return EPS_METRIC_DOT_PRODUCT;
  return -1;
}
}  // namespace epsdrop

namespace vectordb {

DistFunc GetDistFunc(engine::meta::FieldType fType, engine::meta::MetricType mType) {
  if (fType == engine::meta::FieldType::VECTOR_FLOAT || fType == engine::meta::FieldType::VECTOR_DOUBLE) {
    switch (mType) {
      case engine::meta::MetricType::COSINE: return &epsdrop::TagCosineDistance;
      case engine::meta::MetricType::DOT_PRODUCT: return &epsdrop::TagInnerProduct;
      default: return &epsdrop::TagL2Sqr;  // EUCLIDEAN and unknown metrics (index.cpp:18-19)
    }
  }
  // sparse vectors stay on the host DBMS: same functions the reference returns
  switch (mType) {
    case engine::meta::MetricType::COSINE: return engine::GetCosineDist;
    case engine::meta::MetricType::DOT_PRODUCT: return engine::GetInnerProductDist;
    default: return engine::GetL2DistSqr;
  }
}

}  // namespace vectordb
