/* shim: thrust::sequence(first, last) writes 0, 1, 2, ... */
#pragma once
#include <numeric>

namespace thrust {
template <typename It>
void sequence(It first, It last) { std::iota(first, last, 0); }
}  // namespace thrust
