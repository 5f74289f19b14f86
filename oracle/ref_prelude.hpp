// ref_prelude.hpp — force-included (-include) when compiling the reference's own
// hot-path translation units *where they lie* under /root/reference (see
// oracle/Makefile). TEST INFRASTRUCTURE; never compiled into the product.
//
// Why it exists: the reference was written against libc++ (macOS) and relies on
// transitive standard-header includes that libstdc++-11 (this image) does not
// provide (M_PI in math_base.hpp:12, std::bit_cast :141, std::memcpy in
// rng.hpp:38, ...). Every #include below is a *standard* header.
//
// The one non-#include line is a stream inserter for std::chrono::milliseconds:
// reference src/core/bvh.hpp:207 prints the BVH build time with `std::cout <<
// buildTime`, a C++20 library feature libstdc++-11 lacks. It only formats a
// debug print; no algorithmic code of the reference is replaced, stubbed or
// altered by this file. (Disclosed in DESIGN.md "Oracle".)
#pragma once
#include <cmath>
#include <math.h>
#include <bit>
#include <cstring>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <array>
#include <optional>
#include <algorithm>
#include <numeric>
#include <limits>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <condition_variable>
#include <thread>
#include <iostream>
#include <sstream>
#include <chrono>
#include <concepts>
#include <ranges>
#include <span>
#include <functional>
#include <variant>
#include <string>
#include <utility>

namespace std { namespace chrono {
inline std::ostream& operator<<(std::ostream& o, const milliseconds& d) {
  return o << d.count() << "ms";
}
} }
