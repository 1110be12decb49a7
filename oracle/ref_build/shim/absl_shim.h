// ORACLE / TEST INFRASTRUCTURE ONLY — shim for building slices of the REFERENCE's own
// sources (oracle/ref_build/extract.py) without abseil, which is not in this image.
//
// Restates the few abseil types the sliced ranges use.  Not abseil code: written from
// abseil's documented semantics (absl/time/time.h, absl/container/flat_hash_map.h @ the
// version CraneSched pins in dependencies/cmake/abseil).
//
//   absl::Duration  signed count of quarter-nanosecond ticks (abseil's resolution) with
//                   +/- infinity; integer division truncates toward zero at tick
//                   granularity, so PreemptSegTree's `st + (ed - st) / 2`
//                   (JobScheduler.h:896) splits exactly as in the reference.
//   absl::Time      Duration since the Unix epoch; InfiniteFuture() = +inf.
//   flat_hash_map   iteration order of the real container is unspecified (and
//   flat_hash_set   randomised per process).  The shims iterate in KEY order, and a map
//                   whose mapped type is a NodeState (has `time_avail_res_map`) places
//                   its elements in one array indexed by the key's RANK
//                   (crane_ref::key_rank: the dense node index the harness encodes in
//                   the craned id).  Consequence: `NodeState*` address order == dense
//                   node index order, which is the canonical cost tie-break of
//                   SURVEY.md §7 (the reference breaks cost ties on the address,
//                   JobScheduler.h:594).  Element addresses are stable, as the
//                   reference's use of `&it->second` requires.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <limits>
#include <map>
#include <new>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace crane_ref {
// Dense index encoded in a craned id by the harness ("n%u"), SIZE_MAX if none.
inline size_t key_rank(const std::string& k) {
  if (k.size() < 2 || k[0] != 'n') return SIZE_MAX;
  size_t v = 0;
  for (size_t i = 1; i < k.size(); ++i) {
    if (k[i] < '0' || k[i] > '9') return SIZE_MAX;
    v = v * 10 + size_t(k[i] - '0');
  }
  return v;
}
// Number of slots an arena-backed map allocates on first use (set by the harness).
inline size_t g_arena_slots = 0;
}  // namespace crane_ref

namespace absl {

class Duration {
 public:
  using rep = __int128;
  static constexpr rep kTicksPerSecond = 4000000000LL;  // quarter nanoseconds
  static constexpr rep kInf = rep(1) << 100;
  constexpr Duration() : t_(0) {}
  static constexpr Duration FromTicks(rep t) { Duration d; d.t_ = t; return d; }
  constexpr rep ticks() const { return t_; }
  constexpr bool is_pos_inf() const { return t_ == kInf; }
  constexpr bool is_neg_inf() const { return t_ == -kInf; }
  constexpr bool is_inf() const { return is_pos_inf() || is_neg_inf(); }

  constexpr Duration& operator+=(Duration o) {
    if (is_inf()) return *this;
    if (o.is_inf()) { t_ = o.t_; return *this; }
    t_ += o.t_;
    return *this;
  }
  constexpr Duration& operator-=(Duration o) {
    if (is_inf()) return *this;
    if (o.is_inf()) { t_ = -o.t_; return *this; }
    t_ -= o.t_;
    return *this;
  }
  constexpr Duration& operator/=(int64_t n) {
    if (is_inf() || n == 0) { t_ = ((t_ < 0) != (n < 0)) ? -kInf : kInf; return *this; }
    t_ /= n;  // truncates toward zero, as abseil's integer division of Durations
    return *this;
  }
  constexpr Duration& operator*=(int64_t n) {
    if (is_inf()) { if (n < 0) t_ = -t_; return *this; }
    t_ *= n;
    return *this;
  }
  constexpr Duration operator-() const { return FromTicks(-t_); }
  friend constexpr bool operator==(Duration a, Duration b) { return a.t_ == b.t_; }
  friend constexpr bool operator!=(Duration a, Duration b) { return a.t_ != b.t_; }
  friend constexpr bool operator<(Duration a, Duration b) { return a.t_ < b.t_; }
  friend constexpr bool operator<=(Duration a, Duration b) { return a.t_ <= b.t_; }
  friend constexpr bool operator>(Duration a, Duration b) { return a.t_ > b.t_; }
  friend constexpr bool operator>=(Duration a, Duration b) { return a.t_ >= b.t_; }

 private:
  rep t_;
};
constexpr Duration operator+(Duration a, Duration b) { return a += b; }
constexpr Duration operator-(Duration a, Duration b) { return a -= b; }
constexpr Duration operator/(Duration a, int64_t n) { return a /= n; }
constexpr Duration operator*(Duration a, int64_t n) { return a *= n; }
constexpr Duration operator*(int64_t n, Duration a) { return a *= n; }

constexpr Duration InfiniteDuration() { return Duration::FromTicks(Duration::kInf); }
constexpr Duration ZeroDuration() { return Duration(); }
constexpr Duration Seconds(int64_t n) { return Duration::FromTicks(Duration::rep(n) * Duration::kTicksPerSecond); }
constexpr Duration Minutes(int64_t n) { return Seconds(n * 60); }
constexpr Duration Hours(int64_t n) { return Seconds(n * 3600); }
constexpr Duration Nanoseconds(int64_t n) { return Duration::FromTicks(Duration::rep(n) * 4); }
constexpr int64_t ToInt64Seconds(Duration d) {
  if (d.is_pos_inf()) return std::numeric_limits<int64_t>::max();
  if (d.is_neg_inf()) return std::numeric_limits<int64_t>::min();
  return int64_t(d.ticks() / Duration::kTicksPerSecond);  // toward zero
}

class Time {
 public:
  constexpr Time() = default;
  static constexpr Time FromDuration(Duration d) { Time t; t.d_ = d; return t; }
  constexpr Duration since_epoch() const { return d_; }
  constexpr Time& operator+=(Duration d) { d_ += d; return *this; }
  constexpr Time& operator-=(Duration d) { d_ -= d; return *this; }
  friend constexpr bool operator==(Time a, Time b) { return a.d_ == b.d_; }
  friend constexpr bool operator!=(Time a, Time b) { return a.d_ != b.d_; }
  friend constexpr bool operator<(Time a, Time b) { return a.d_ < b.d_; }
  friend constexpr bool operator<=(Time a, Time b) { return a.d_ <= b.d_; }
  friend constexpr bool operator>(Time a, Time b) { return a.d_ > b.d_; }
  friend constexpr bool operator>=(Time a, Time b) { return a.d_ >= b.d_; }

 private:
  Duration d_;
};
constexpr Time operator+(Time t, Duration d) { return t += d; }
constexpr Time operator-(Time t, Duration d) { return t -= d; }
constexpr Duration operator-(Time a, Time b) {
  if (a.since_epoch().is_inf()) return a.since_epoch();
  if (b.since_epoch().is_inf()) return -b.since_epoch();
  return a.since_epoch() - b.since_epoch();
}
constexpr Time InfiniteFuture() { return Time::FromDuration(InfiniteDuration()); }
constexpr Time InfinitePast() { return Time::FromDuration(-InfiniteDuration()); }
constexpr Time UnixEpoch() { return Time(); }
constexpr Time FromUnixSeconds(int64_t s) { return Time::FromDuration(Seconds(s)); }
// absl::Now(): the step scheduler stamps a step's start time with it (CtldPublicDefs.cpp:2041); the harness sets the clock.
inline int64_t g_ref_now_sec = 0;
inline Time Now() { return FromUnixSeconds(g_ref_now_sec); }
constexpr int64_t ToUnixSeconds(Time t) {
  // abseil floors; whole-second inputs only on this path
  const Duration d = t.since_epoch();
  if (d.is_inf()) return ToInt64Seconds(d);
  Duration::rep q = d.ticks() / Duration::kTicksPerSecond;
  if (d.ticks() % Duration::kTicksPerSecond < 0) --q;
  return int64_t(q);
}

class Mutex {};
class MutexLock {
 public:
  explicit MutexLock(Mutex*) {}
};

// ---------------------------------------------------------------------------------------
// containers
// ---------------------------------------------------------------------------------------
namespace ref_detail {

// NodeSelect keeps every NodeState, selector and cost in LOCALS (JobScheduler.cpp:6563,6721-6722) that die when it
// returns.  To let the harness read the final time maps and costs without editing a line of the slice, a shim map
// offers its elements to an observer right before it destroys them.
template <class K, class V>
struct DtorHook {
  static inline std::function<void(const void* map, const K&, V&)> fn;
};

template <class V>
concept NodeStateLike = requires(V& v) { v.time_avail_res_map; };

// Key-ordered, address-stable map (std::map underneath).
template <class K, class V>
class OrderedMap {
  using Impl = std::map<K, V>;
  Impl m_;

 public:
  using key_type = K;
  using mapped_type = V;
  using value_type = typename Impl::value_type;
  using iterator = typename Impl::iterator;
  using const_iterator = typename Impl::const_iterator;
  using size_type = size_t;

  OrderedMap() = default;
  OrderedMap(const OrderedMap&) = default;
  OrderedMap(OrderedMap&&) = default;
  OrderedMap& operator=(const OrderedMap&) = default;
  OrderedMap& operator=(OrderedMap&&) = default;
  ~OrderedMap() {
    if (DtorHook<K, V>::fn) {
      try { for (auto& [k, v] : m_) DtorHook<K, V>::fn(this, k, v); } catch (...) {}
    }
  }

  iterator begin() { return m_.begin(); }
  iterator end() { return m_.end(); }
  const_iterator begin() const { return m_.begin(); }
  const_iterator end() const { return m_.end(); }
  size_t size() const { return m_.size(); }
  bool empty() const { return m_.empty(); }
  void clear() { m_.clear(); }
  void reserve(size_t) {}
  template <class Q> iterator find(const Q& k) { return m_.find(K(k)); }
  template <class Q> const_iterator find(const Q& k) const { return m_.find(K(k)); }
  template <class Q> bool contains(const Q& k) const { return m_.count(K(k)) != 0; }
  template <class Q> size_t count(const Q& k) const { return m_.count(K(k)); }
  template <class Q> V& at(const Q& k) { return m_.at(K(k)); }
  template <class Q> const V& at(const Q& k) const { return m_.at(K(k)); }
  template <class Q> V& operator[](const Q& k) { return m_[K(k)]; }
  template <class... A> std::pair<iterator, bool> emplace(A&&... a) { return m_.emplace(std::forward<A>(a)...); }
  template <class Q, class... A> std::pair<iterator, bool> try_emplace(const Q& k, A&&... a) {
    return m_.try_emplace(K(k), std::forward<A>(a)...);
  }
  std::pair<iterator, bool> insert(const value_type& v) { return m_.insert(v); }
  std::pair<iterator, bool> insert(value_type&& v) { return m_.insert(std::move(v)); }
  template <class Q> size_t erase(const Q& k) { return m_.erase(K(k)); }
  iterator erase(iterator it) { return m_.erase(it); }
  iterator erase(const_iterator it) { return m_.erase(it); }
};

// Map of NodeState-like values: element for key k lives at slot key_rank(k) of one array,
// so that element ADDRESS order is dense node index order.  Iterates in slot order.
template <class K, class V>
class ArenaMap {
 public:
  using key_type = K;
  using mapped_type = V;
  using value_type = std::pair<const K, V>;
  using size_type = size_t;

 private:
  value_type* slots_ = nullptr;
  size_t cap_ = 0;
  std::vector<uint8_t> used_;
  size_t size_ = 0;

  void ensure() {
    if (slots_) return;
    cap_ = crane_ref::g_arena_slots;
    if (cap_ == 0) throw std::logic_error("crane_ref: arena map used before g_arena_slots was set");
    slots_ = static_cast<value_type*>(::operator new(cap_ * sizeof(value_type), std::align_val_t(alignof(value_type))));
    used_.assign(cap_, 0);
  }
  size_t slot_of(const K& k) const {
    const size_t r = crane_ref::key_rank(k);
    if (r == SIZE_MAX || (slots_ && r >= cap_)) throw std::out_of_range("crane_ref: craned id without a dense index");
    return r;
  }
  size_t next_used(size_t i) const {
    while (i < cap_ && !used_[i]) ++i;
    return i;
  }

  template <bool Const>
  class Iter {
    using Owner = std::conditional_t<Const, const ArenaMap, ArenaMap>;
    Owner* o_ = nullptr;
    size_t i_ = 0;
    friend class ArenaMap;

   public:
    using value_type = ArenaMap::value_type;
    using reference = std::conditional_t<Const, const value_type&, value_type&>;
    using pointer = std::conditional_t<Const, const value_type*, value_type*>;
    using difference_type = std::ptrdiff_t;
    using iterator_category = std::forward_iterator_tag;
    Iter() = default;
    Iter(Owner* o, size_t i) : o_(o), i_(i) {}
    template <bool C2, class = std::enable_if_t<Const && !C2>>
    Iter(const Iter<C2>& x) : o_(x.o_), i_(x.i_) {}
    reference operator*() const { return o_->slots_[i_]; }
    pointer operator->() const { return &o_->slots_[i_]; }
    Iter& operator++() { i_ = o_->next_used(i_ + 1); return *this; }
    Iter operator++(int) { Iter t = *this; ++*this; return t; }
    friend bool operator==(const Iter& a, const Iter& b) { return a.i_ == b.i_; }
    friend bool operator!=(const Iter& a, const Iter& b) { return a.i_ != b.i_; }
    template <bool> friend class Iter;
  };

 public:
  using iterator = Iter<false>;
  using const_iterator = Iter<true>;

  ArenaMap() = default;
  ArenaMap(const ArenaMap&) = delete;
  ArenaMap& operator=(const ArenaMap&) = delete;
  ArenaMap(ArenaMap&& o) noexcept : slots_(o.slots_), cap_(o.cap_), used_(std::move(o.used_)), size_(o.size_) {
    o.slots_ = nullptr; o.cap_ = 0; o.size_ = 0;
  }
  ArenaMap& operator=(ArenaMap&& o) noexcept {
    if (this != &o) { destroy(); slots_ = o.slots_; cap_ = o.cap_; used_ = std::move(o.used_); size_ = o.size_; o.slots_ = nullptr; o.cap_ = 0; o.size_ = 0; }
    return *this;
  }
  ~ArenaMap() { destroy(); }
  void destroy() {
    if (!slots_) return;
    if (DtorHook<K, V>::fn) {
      try { for (size_t i = 0; i < cap_; ++i) if (used_[i]) DtorHook<K, V>::fn(this, slots_[i].first, slots_[i].second); } catch (...) {}
    }
    for (size_t i = 0; i < cap_; ++i) if (used_[i]) slots_[i].~value_type();
    ::operator delete(slots_, std::align_val_t(alignof(value_type)));
    slots_ = nullptr; cap_ = 0; size_ = 0; used_.clear();
  }

  iterator begin() { return iterator(this, next_used(0)); }
  iterator end() { return iterator(this, cap_); }
  const_iterator begin() const { return const_iterator(this, next_used(0)); }
  const_iterator end() const { return const_iterator(this, cap_); }
  size_t size() const { return size_; }
  bool empty() const { return size_ == 0; }
  void reserve(size_t) {}

  iterator find(const K& k) {
    if (!slots_) return end();
    const size_t r = crane_ref::key_rank(k);
    return (r < cap_ && used_[r]) ? iterator(this, r) : end();
  }
  const_iterator find(const K& k) const {
    if (!slots_) return end();
    const size_t r = crane_ref::key_rank(k);
    return (r < cap_ && used_[r]) ? const_iterator(this, r) : end();
  }
  bool contains(const K& k) const { return find(k) != end(); }
  V& at(const K& k) {
    auto it = find(k);
    if (it == end()) throw std::out_of_range("flat_hash_map::at");
    return it->second;
  }
  const V& at(const K& k) const {
    auto it = find(k);
    if (it == end()) throw std::out_of_range("flat_hash_map::at");
    return it->second;
  }
  template <class KK, class VV>
  std::pair<iterator, bool> emplace(KK&& k, VV&& v) {
    ensure();
    const K key(std::forward<KK>(k));
    const size_t r = slot_of(key);
    if (r >= cap_) throw std::out_of_range("crane_ref: dense index beyond the arena");
    if (used_[r]) return {iterator(this, r), false};
    new (&slots_[r]) value_type(std::piecewise_construct, std::forward_as_tuple(key), std::forward_as_tuple(std::forward<VV>(v)));
    used_[r] = 1;
    ++size_;
    return {iterator(this, r), true};
  }
};

template <class K, class V>
struct MapChoice { using type = OrderedMap<K, V>; };
template <class K, NodeStateLike V>
struct MapChoice<K, V> { using type = ArenaMap<K, V>; };

}  // namespace ref_detail

namespace container_internal {
template <class K> using hash_default_hash = std::hash<K>;
template <class K> using hash_default_eq = std::equal_to<K>;
}  // namespace container_internal

template <class K, class V, class Hash = void, class Eq = void, class Alloc = void>
class flat_hash_map : public ref_detail::MapChoice<K, V>::type {};

template <class K, class V, class Cmp = std::less<K>>
using btree_map = std::map<K, V, Cmp>;

// Key-ordered set.  For std::variant<PdJobInScheduler*, RnJobInScheduler*> elements the order is
// (alternative, address); the harness allocates the jobs of each kind in one array, so the
// order within a kind is the job's position in its vector.
template <class T, class Hash = void, class Eq = void, class Alloc = void>
class flat_hash_set {
  using Impl = std::set<T>;
  Impl s_;

 public:
  using value_type = T;
  using iterator = typename Impl::const_iterator;
  using const_iterator = typename Impl::const_iterator;
  flat_hash_set() = default;
  flat_hash_set(std::initializer_list<T> il) : s_(il) {}
  iterator begin() const { return s_.begin(); }
  iterator end() const { return s_.end(); }
  size_t size() const { return s_.size(); }
  bool empty() const { return s_.empty(); }
  void clear() { s_.clear(); }
  void reserve(size_t) {}
  template <class Q> bool contains(const Q& v) const { return s_.count(T(v)) != 0; }
  template <class Q> size_t count(const Q& v) const { return s_.count(T(v)); }
  template <class Q> iterator find(const Q& v) const { return s_.find(T(v)); }
  std::pair<iterator, bool> insert(const T& v) { return s_.insert(v); }
  std::pair<iterator, bool> insert(T&& v) { return s_.insert(std::move(v)); }
  template <class It> void insert(It a, It b) { s_.insert(a, b); }
  template <class... A> std::pair<iterator, bool> emplace(A&&... a) { return s_.emplace(std::forward<A>(a)...); }
  template <class Q> size_t erase(const Q& v) { return s_.erase(T(v)); }
  // abseil's erase(iterator) returns void; `erase(it++)` (JobScheduler.cpp:6551) works with both
  void erase(iterator it) { s_.erase(it); }
};

}  // namespace absl

#define ABSL_ASSERT(cond) CRANE_REF_CHECK(cond, "ABSL_ASSERT(" #cond ")")
