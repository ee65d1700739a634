// ORACLE BUILD SHIM: boost::detail::spinlock on std::atomic_flag.
#pragma once
#include <atomic>
namespace boost {
namespace detail {
class spinlock {
  std::atomic_flag f_ = ATOMIC_FLAG_INIT;

 public:
  void lock() {
    while (f_.test_and_set(std::memory_order_acquire)) {
    }
  }
  void unlock() { f_.clear(std::memory_order_release); }
  bool try_lock() { return !f_.test_and_set(std::memory_order_acquire); }
};
}  // namespace detail
}  // namespace boost
