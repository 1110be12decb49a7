// ORACLE / TEST INFRASTRUCTURE ONLY — stand-in for <fpm/fixed.hpp> so that slices of the
// REFERENCE's own sources compile here (oracle/ref_build/extract.py).  The reference pins
// github.com/MikeLankamp/fpm @ b46537fe9697e1a598ac8a26f8ae43d8b286ac3f
// (dependencies/cmake/fpm/CMakeLists.txt:6-8), which is fetched from the network and is not
// in this image.  This file restates the published arithmetic of fpm::fixed for the
// operations CraneSched's `cpu_t = fpm::fixed<int64_t, __int128, 8>` performs
// (PublicHeader.h:44): raw = value * 2^F; integral construction multiplies, floating
// construction rounds half away from zero; conversion to an integer divides the raw value
// (truncation toward zero), to a floating type divides as that type; + - compare on raw;
// `*= integer` / `/= integer` act on raw; fixed * fixed and fixed / fixed round to nearest
// through the intermediate type.
#pragma once
#include <cassert>
#include <cstdint>
#include <functional>
#include <type_traits>

namespace fpm {

template <typename BaseType, typename IntermediateType, unsigned int FractionBits, bool EnableRounding = true>
class fixed {
  static_assert(std::is_integral<BaseType>::value, "BaseType must be an integral type");
  static constexpr BaseType FRACTION_MULT = BaseType(1) << FractionBits;
  struct raw_construct_tag {};
  constexpr fixed(BaseType val, raw_construct_tag) noexcept : m_value(val) {}

 public:
  fixed() noexcept = default;

  template <typename T, typename std::enable_if<std::is_integral<T>::value>::type* = nullptr>
  constexpr explicit fixed(T val) noexcept : m_value(static_cast<BaseType>(val * FRACTION_MULT)) {}

  template <typename T, typename std::enable_if<std::is_floating_point<T>::value>::type* = nullptr>
  constexpr explicit fixed(T val) noexcept
      : m_value(static_cast<BaseType>(EnableRounding ? (val >= T{0} ? (val * FRACTION_MULT + T{0.5}) : (val * FRACTION_MULT - T{0.5}))
                                                     : (val * FRACTION_MULT))) {}

  template <typename T, typename std::enable_if<std::is_floating_point<T>::value>::type* = nullptr>
  constexpr explicit operator T() const noexcept { return static_cast<T>(m_value) / FRACTION_MULT; }

  template <typename T, typename std::enable_if<std::is_integral<T>::value>::type* = nullptr>
  constexpr explicit operator T() const noexcept { return static_cast<T>(m_value / FRACTION_MULT); }

  constexpr BaseType raw_value() const noexcept { return m_value; }
  static constexpr fixed from_raw_value(BaseType value) noexcept { return fixed(value, raw_construct_tag{}); }

  constexpr fixed operator-() const noexcept { return from_raw_value(-m_value); }

  fixed& operator+=(const fixed& y) noexcept { m_value += y.m_value; return *this; }
  fixed& operator-=(const fixed& y) noexcept { m_value -= y.m_value; return *this; }
  template <typename I, typename std::enable_if<std::is_integral<I>::value>::type* = nullptr>
  fixed& operator+=(I y) noexcept { m_value += y * FRACTION_MULT; return *this; }
  template <typename I, typename std::enable_if<std::is_integral<I>::value>::type* = nullptr>
  fixed& operator-=(I y) noexcept { m_value -= y * FRACTION_MULT; return *this; }

  fixed& operator*=(const fixed& y) noexcept {
    if (EnableRounding) {
      auto value = (static_cast<IntermediateType>(m_value) * y.m_value) / (FRACTION_MULT / 2);
      m_value = static_cast<BaseType>((value / 2) + (value % 2));
    } else {
      auto value = (static_cast<IntermediateType>(m_value) * y.m_value) / FRACTION_MULT;
      m_value = static_cast<BaseType>(value);
    }
    return *this;
  }
  template <typename I, typename std::enable_if<std::is_integral<I>::value>::type* = nullptr>
  fixed& operator*=(I y) noexcept { m_value *= y; return *this; }

  fixed& operator/=(const fixed& y) noexcept {
    assert(y.m_value != 0);
    if (EnableRounding) {
      auto value = (static_cast<IntermediateType>(m_value) * FRACTION_MULT * 2) / y.m_value;
      m_value = static_cast<BaseType>((value / 2) + (value % 2));
    } else {
      auto value = (static_cast<IntermediateType>(m_value) * FRACTION_MULT) / y.m_value;
      m_value = static_cast<BaseType>(value);
    }
    return *this;
  }
  template <typename I, typename std::enable_if<std::is_integral<I>::value>::type* = nullptr>
  fixed& operator/=(I y) noexcept { m_value /= y; return *this; }

 private:
  BaseType m_value;
};

#define FPM_BIN(op)                                                                                                        \
  template <typename B, typename I, unsigned int F, bool R>                                                                \
  constexpr fixed<B, I, F, R> operator op(const fixed<B, I, F, R>& x, const fixed<B, I, F, R>& y) noexcept {               \
    return fixed<B, I, F, R>(x) op## = y;                                                                                  \
  }                                                                                                                        \
  template <typename B, typename I, unsigned int F, bool R, typename T, typename std::enable_if<std::is_integral<T>::value>::type* = nullptr> \
  constexpr fixed<B, I, F, R> operator op(const fixed<B, I, F, R>& x, T y) noexcept {                                      \
    return fixed<B, I, F, R>(x) op## = y;                                                                                  \
  }
FPM_BIN(+)
FPM_BIN(-)
FPM_BIN(*)
FPM_BIN(/)
#undef FPM_BIN

#define FPM_CMP(op)                                                                                       \
  template <typename B, typename I, unsigned int F, bool R>                                               \
  constexpr bool operator op(const fixed<B, I, F, R>& x, const fixed<B, I, F, R>& y) noexcept {           \
    return x.raw_value() op y.raw_value();                                                                \
  }
FPM_CMP(==)
FPM_CMP(!=)
FPM_CMP(<)
FPM_CMP(>)
FPM_CMP(<=)
FPM_CMP(>=)
#undef FPM_CMP

}  // namespace fpm

namespace std {
template <typename B, typename I, unsigned int F, bool R>
struct hash<fpm::fixed<B, I, F, R>> {
  size_t operator()(const fpm::fixed<B, I, F, R>& x) const noexcept { return std::hash<B>{}(x.raw_value()); }
};
}  // namespace std
