// binary/algorithm/all.hpp — umbrella include (reference: library/include/binary/algorithm/all.hpp).
#ifndef BINARY_AMD_ALGORITHM_ALL_HPP_
#define BINARY_AMD_ALGORITHM_ALL_HPP_
#include <binary/algorithm/interval_tree.hpp>
#include <binary/algorithm/rb_tree.hpp>
#endif  // BINARY_AMD_ALGORITHM_ALL_HPP_
