// ORACLE BUILD SHIM for the python-binding build only: the one constant interface.cpp needs
// (/root/reference/engine/server/web_server/web_controller.hpp:38).
#pragma once
#include <cstdint>
namespace vectordb {
namespace server {
namespace web {
constexpr const int64_t InitTableScale = 150000;
}
}  // namespace server
}  // namespace vectordb
