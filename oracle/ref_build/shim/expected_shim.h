// ORACLE / TEST INFRASTRUCTURE ONLY — stand-ins for two C++23 library pieces the sliced REFERENCE
// sources use and libstdc++-11 (this image) does not have:
//
//   std::expected<T, E> / std::unexpected<E>   AccountMetaContainer.cpp:180,345,508,542,672,891 (run-limit checks return
//                                              std::expected<void, std::string>), CtldPublicDefs.h:339
//   std::ranges::to<std::vector>()             CtldPublicDefs.cpp:2132 (the step's craned ids out of a keys view)
//
// libstdc++-11 still declares C++98's function std::unexpected() (<exception>), so the class template is called
// std::crane_ref_unexpected and ref_harness.cpp renames the token `unexpected` around the slices (a macro rename
// like its `unordered_map` / `sort` ones: no slice line is edited).
//
// Written for this build from the standard's description ([expected], [range.utility.conv]); only the members
// the slices call.  Declaring them in namespace std is what lets the slices compile UNEDITED; nothing outside
// oracle/_ref links against this.
#pragma once
#include <ranges>
#include <string>
#include <type_traits>
#include <utility>
#include <variant>
#include <vector>

namespace std {

template <class E>
class crane_ref_unexpected {
 public:
  template <class U = E, class = std::enable_if_t<std::is_constructible_v<E, U>>>
  constexpr explicit crane_ref_unexpected(U&& e) : e_(std::forward<U>(e)) {}
  constexpr const E& error() const& { return e_; }
  constexpr E& error() & { return e_; }
  constexpr E&& error() && { return std::move(e_); }

 private:
  E e_;
};
template <class E>
crane_ref_unexpected(E) -> crane_ref_unexpected<E>;

template <class T, class E>
class expected {
 public:
  constexpr expected() : v_(std::in_place_index<0>) {}
  template <class U = T, class = std::enable_if_t<std::is_constructible_v<T, U> && !std::is_same_v<std::remove_cvref_t<U>, expected>>>
  constexpr expected(U&& v) : v_(std::in_place_index<0>, std::forward<U>(v)) {}
  template <class G, class = std::enable_if_t<std::is_constructible_v<E, const G&>>>
  constexpr expected(const crane_ref_unexpected<G>& u) : v_(std::in_place_index<1>, u.error()) {}
  template <class G, class = std::enable_if_t<std::is_constructible_v<E, G>>>
  constexpr expected(crane_ref_unexpected<G>&& u) : v_(std::in_place_index<1>, std::move(u).error()) {}
  constexpr bool has_value() const { return v_.index() == 0; }
  constexpr explicit operator bool() const { return has_value(); }
  constexpr T& value() { return std::get<0>(v_); }
  constexpr const T& value() const { return std::get<0>(v_); }
  constexpr T& operator*() { return std::get<0>(v_); }
  constexpr const T& operator*() const { return std::get<0>(v_); }
  constexpr T* operator->() { return &std::get<0>(v_); }
  constexpr const T* operator->() const { return &std::get<0>(v_); }
  constexpr E& error() { return std::get<1>(v_); }
  constexpr const E& error() const { return std::get<1>(v_); }

 private:
  std::variant<T, E> v_;
};

template <class E>
class expected<void, E> {
 public:
  constexpr expected() = default;
  template <class G, class = std::enable_if_t<std::is_constructible_v<E, const G&>>>
  expected(const crane_ref_unexpected<G>& u) : ok_(false), e_(u.error()) {}
  template <class G, class = std::enable_if_t<std::is_constructible_v<E, G>>>
  expected(crane_ref_unexpected<G>&& u) : ok_(false), e_(std::move(u).error()) {}
  bool has_value() const { return ok_; }
  explicit operator bool() const { return ok_; }
  void value() const {}
  E& error() { return e_; }
  const E& error() const { return e_; }

 private:
  bool ok_{true};
  E e_{};
};

namespace ranges {
namespace crane_ref_detail {
template <template <class...> class C>
struct to_closure {};
template <std::ranges::input_range R, template <class...> class C>
auto operator|(R&& r, to_closure<C>) {
  using V = std::ranges::range_value_t<R>;
  C<V> out;
  for (auto&& x : r) out.insert(out.end(), static_cast<V>(x));
  return out;
}
}  // namespace crane_ref_detail
template <template <class...> class C>
constexpr auto to() {
  return crane_ref_detail::to_closure<C>{};
}
}  // namespace ranges

}  // namespace std
