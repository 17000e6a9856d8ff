// The library's exception type (never crosses the C ABI).  HIP-free, so that the host-side logic that uses it (api_guard.h,
// dispatcher_core.h) can also be built with g++ and the thread / address sanitizers by the CPU suite.
#pragma once
#include <stdexcept>
#include <string>

namespace kx {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

}  // namespace kx
