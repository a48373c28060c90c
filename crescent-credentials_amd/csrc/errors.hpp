// The library's one exception type (code = a cg_status), free of the HIP runtime so that host-only headers can use it.
#pragma once
#include <stdexcept>
#include <string>

#include "../../include/crescent_gpu.h"

namespace cg {
struct HipError : std::runtime_error {
    int code;
    HipError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
}  // namespace cg
