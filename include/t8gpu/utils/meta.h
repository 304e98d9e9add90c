// t8gpu/utils/meta.h (MI355X backend) -- compile-time helpers used by Subgrid<> and the accessors.
// Same names and results as the reference's t8gpu/utils/meta.h:25-120, written with C++17 constexpr
// functions and fold expressions instead of recursive class templates.
#ifndef T8GPU_HIP_UTILS_META_H
#define T8GPU_HIP_UTILS_META_H

#include <cstddef>
#include <type_traits>
#include <utility>

namespace t8gpu::meta {

  namespace detail {
    template<typename First, typename... Rest>
    inline constexpr bool same_as_first = (std::is_same_v<std::remove_cv_t<First>, std::remove_cv_t<Rest>> && ...);

    template<typename From, typename To, typename = void>
    struct castable : std::false_type {};
    template<typename From, typename To>
    struct castable<From, To, std::void_t<decltype(static_cast<To>(std::declval<From>()))>> : std::true_type {};

    template<int... values>
    constexpr int pick(int index) {
      constexpr int table[] = {values...};
      return table[index];
    }
    // product of values[lo, hi)
    template<int... values>
    constexpr int product(int lo, int hi) {
      constexpr int table[] = {values...};
      int           p      = 1;
      for (int i = lo; i < hi && i < static_cast<int>(sizeof...(values)); i++) p *= table[i];
      return p;
    }
    constexpr std::size_t ilog2(std::size_t x) { return x <= 1 ? 0 : 1 + ilog2(x / 2); }
  }  // namespace detail

  /// true iff all types are equal after stripping cv (false for an empty pack), meta.h:25-36.
  template<typename... Ts>
  struct all_same : std::bool_constant<false> {};
  template<typename T, typename... Ts>
  struct all_same<T, Ts...> : std::bool_constant<detail::same_as_first<T, Ts...>> {};
  template<typename... Ts>
  inline constexpr bool all_same_v = all_same<Ts...>::value;

  /// true iff static_cast<U>(T) is well-formed (explicit conversions included), meta.h:47-56.
  template<typename T, typename U>
  struct is_explicitly_convertible_to : detail::castable<T, U> {};
  template<typename T, typename U>
  inline constexpr bool is_explicitly_convertible_to_v = is_explicitly_convertible_to<T, U>::value;

  /// pack element at `index`, meta.h:64-75.
  template<int index, int... args>
  struct argpack_at : std::integral_constant<int, detail::pick<args...>(index)> {
    static_assert(index >= 0 && index < static_cast<int>(sizeof...(args)), "argpack_at: index out of range");
  };
  template<int... args>
  inline constexpr int argpack_at_v = argpack_at<args...>::value;

  /// product of the pack elements from `index` on (1 past the end), meta.h:83-93.
  template<int index, int arg1, int... args>
  struct argpack_mul_from
      : std::integral_constant<int, detail::product<arg1, args...>(index, static_cast<int>(1 + sizeof...(args)))> {};
  template<int index, int arg1, int... args>
  inline constexpr int argpack_mul_from_v = argpack_mul_from<index, arg1, args...>::value;

  /// product of the pack elements before `index`, meta.h:100-111.
  template<int index, int arg1, int... args>
  struct argpack_mul_to : std::integral_constant<int, detail::product<arg1, args...>(0, index)> {};
  template<int index, int arg1, int... args>
  inline constexpr int argpack_mul_to_v = argpack_mul_to<index, arg1, args...>::value;

  /// integer log2, meta.h:113-120.
  template<std::size_t x>
  struct log2 : std::integral_constant<std::size_t, detail::ilog2(x)> {};
  template<std::size_t x>
  inline constexpr std::size_t log2_v = log2<x>::value;

}  // namespace t8gpu::meta

#endif  // T8GPU_HIP_UTILS_META_H
