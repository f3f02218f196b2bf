// binary/algorithm/all.hpp
//
// One include for the tree algorithms of this drop-in: the host-side red-black tree (structure questions) and the
// device-backed interval tree (overlap queries through libbivx.so). The reference has an umbrella header of the
// same path (library/include/binary/algorithm/all.hpp); its experimental shared_ptr tree is not part of the
// interval-overlap path and has no counterpart here.
#ifndef BINARY_AMD_ALGORITHM_ALL_HPP_
#define BINARY_AMD_ALGORITHM_ALL_HPP_

#include <binary/concepts.hpp>

#include <binary/algorithm/rb_tree.hpp>

#include <binary/algorithm/interval_tree.hpp>

#endif  // BINARY_AMD_ALGORITHM_ALL_HPP_
