// ORACLE BUILD SHIM: boost::mt19937 == std::mt19937 (same engine parameters).
#pragma once
#include <random>
namespace boost {
using mt19937 = std::mt19937;
}
