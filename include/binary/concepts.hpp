// binary/concepts.hpp
//
// Three small constraints used by the IntervalTree overload set. The names and meaning are part of the
// reference's public interface (library/include/binary/concepts.hpp:11-19: IsAnyOf, IsAllOf, ArgsConstructible),
// so code written against it keeps compiling; the definitions below are this repository's own.
//
//   IsAnyOf<T, U...>           T is one of the U (exact type identity)
//   IsAllOf<T, U...>           every U is T
//   ArgsConstructible<T, A...> a T can be built from the arguments A..., and the argument list is not simply
//                              "one T": that call belongs to the overload taking T itself, and excluding it
//                              here is what keeps find_overlap(interval) and find_overlap(args...) apart.
#ifndef BINARY_AMD_CONCEPTS_HPP_
#define BINARY_AMD_CONCEPTS_HPP_

#include <concepts>
#include <type_traits>

namespace binary::concepts {

  namespace detail {
    template <typename T, typename... Candidates>
    inline constexpr bool occurs_in = std::disjunction_v<std::is_same<T, Candidates>...>;

    template <typename T, typename... Others>
    inline constexpr bool all_are = std::conjunction_v<std::is_same<T, Others>...>;
  }  // namespace detail

  template <typename T, typename... U>
  concept IsAnyOf = detail::occurs_in<T, U...>;

  template <typename T, typename... U>
  concept IsAllOf = detail::all_are<T, U...>;

  template <typename T, typename... Args>
  concept ArgsConstructible = requires {
    requires std::is_constructible_v<T, Args...>;
    requires !detail::occurs_in<T, std::remove_cvref_t<Args>...>;
  };

}  // namespace binary::concepts

#endif  // BINARY_AMD_CONCEPTS_HPP_
