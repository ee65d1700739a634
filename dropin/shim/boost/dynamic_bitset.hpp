// ORACLE BUILD SHIM: minimal boost::dynamic_bitset over std::vector<bool> (only what nsg.cpp uses).
#pragma once
#include <cstddef>
#include <vector>
namespace boost {
template <class Block = unsigned long>
class dynamic_bitset {
  std::vector<bool> v_;

 public:
  dynamic_bitset() {}
  dynamic_bitset(std::size_t n, unsigned long) : v_(n, false) {}
  std::vector<bool>::reference operator[](std::size_t i) { return v_[i]; }
  bool operator[](std::size_t i) const { return v_[i]; }
  void reset() { v_.assign(v_.size(), false); }
  std::size_t size() const { return v_.size(); }
};
}  // namespace boost
