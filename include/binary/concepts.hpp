// binary/concepts.hpp — the small concept helpers the IntervalTree overload set relies on
// (reference: library/include/binary/concepts.hpp:11-19; same names and meaning, written from scratch).
#ifndef BINARY_AMD_CONCEPTS_HPP_
#define BINARY_AMD_CONCEPTS_HPP_

#include <concepts>
#include <type_traits>

namespace binary::concepts {

  /// T is exactly one of U...
  template <typename T, typename... U>
  concept IsAnyOf = (std::same_as<T, U> || ...);

  /// every U is exactly T
  template <typename T, typename... U>
  concept IsAllOf = (std::same_as<T, U> && ...);

  /// T can be built from Args..., and Args is not just "a T" (that case belongs to the non-variadic overload)
  template <typename T, typename... Args>
  concept ArgsConstructible = std::constructible_from<T, Args...> && !IsAnyOf<T, std::remove_cvref_t<Args>...>;

}  // namespace binary::concepts

#endif  // BINARY_AMD_CONCEPTS_HPP_
