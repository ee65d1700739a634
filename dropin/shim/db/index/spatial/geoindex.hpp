// ORACLE BUILD SHIM (test infrastructure). No-op geo index: Boost.Geometry is absent here
// and NEARBY() filters are outside the dense ANN hot path.
// API shape follows /root/reference/engine/db/index/spatial/geoindex.hpp:20-40.
#pragma once
#include <cstdint>
#include <utility>
#include <vector>

namespace vectordb {
namespace engine {
namespace index {

class GeospatialIndex {
 public:
  struct point_t {
    double a, b;
    point_t(double a_ = 0, double b_ = 0) : a(a_), b(b_) {}
  };
  typedef std::pair<point_t, int64_t> value_t;

  GeospatialIndex() {}
  ~GeospatialIndex() {}
  void insertPoint(double, double, int64_t) {}
  void deletePoint(double, double, int64_t) {}
  void searchWithinRadius(double, double, double, std::vector<value_t>&) const {}
  static double distance(const point_t&, const point_t&) { return 0.0; }
};

}  // namespace index
}  // namespace engine
}  // namespace vectordb
