// Metric tags shared by dropin/dist_func.cpp (GetDistFunc) and dropin/vec_search_executor.cpp.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
namespace epsdrop {
float TagL2Sqr(const void* a, const void* b, const void* dim_ptr);
float TagCosineDistance(const void* a, const void* b, const void* dim_ptr);
float TagInnerProduct(const void* a, const void* b, const void* dim_ptr);
// EPS_METRIC_* of a DistFunc produced by GetDistFunc, or -1 for a sparse-vector function
int MetricOfDistFunc(const void* fn);
// The graph of one dense vector field, built ON the field's device mirror (the same HBM copy of the rows the field's executors
// search; created here if no executor has yet: rows cross PCIe once, on the mirror's own device(s) - EPS_DEVICES).  owner: the
// ANNGraphSegment the graph belongs to (executors constructed from it find the graph already on the device).  One device: the
// CSR comes back as new[] arrays in the reference's layout.  Hash-sharded mirror: every shard builds the graph of ITS rows and
// keeps it (a graph over the whole table cannot be split); the CSR returned is the n-node placeholder without edges.
// *keep: keeps the mirror alive (the caller stores it next to the graph).  Returns "" or the error text.
std::string BuildGraphOnMirror(const float* column, int64_t n, int64_t dim, int metric, const void* owner, int64_t** off, int64_t** nbr,
                               int64_t* nav, std::shared_ptr<void>* keep);
}  // namespace epsdrop
