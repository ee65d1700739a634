// Metric tags shared by dropin/dist_func.cpp (GetDistFunc) and dropin/vec_search_executor.cpp.
#pragma once
namespace epsdrop {
float TagL2Sqr(const void* a, const void* b, const void* dim_ptr);
float TagCosineDistance(const void* a, const void* b, const void* dim_ptr);
float TagInnerProduct(const void* a, const void* b, const void* dim_ptr);
// EPS_METRIC_* of a DistFunc produced by GetDistFunc, or -1 for a sparse-vector function
int MetricOfDistFunc(const void* fn);
}  // namespace epsdrop
