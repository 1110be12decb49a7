// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it,
// and only as the checker.  The product path (cranesched_amd/) never links or calls it.
//
// CPU restatement of CraneSched's per-node resource algebra, in two forms:
//   LitAlgebra  — container-literal: std::set core ids, name->type->std::set slot
//                 maps, exactly the containers of the reference
//                 (src/Utilities/PublicHeader/include/crane/PublicHeader.h:425-494,555-639);
//   MaskAlgebra — the canonical integer form of SURVEY.md Appendix A (bit masks).
// tests/ prove Lit == Mask on random inputs and pin both against the reference's own
// known-answer vectors (test/Utilities/dedicated_resource_test.cpp:27-251).
//
// PARITY STATUS: PINNED.  The reference's own build system cannot run here (C++23 stdlib,
// protobuf/abseil/fpm fetched from the network — SURVEY.md §8c) and it ships no test for
// GetFeasibleResourceInNode / Ckmin / NodeSelect, but the path's own sources compile:
// oracle/_ref (oracle/ref_build/, built by `make -C oracle _ref` from slices of
// /root/reference cut at build time) is the reference's code, and tests/test_ref_pin.py +
// tests/test_ref_pin_limits_steps.py hold every restatement under oracle/ to it, exactly
// (6 000 random algebra calls, whole NodeSelect cycles incl. final costs and time maps,
// priorities, run-limit admission, step scheduling).
//
// Third-party arithmetic restated here: fpm::fixed<int64_t,__int128,8>
// (github.com/MikeLankamp/fpm @ b46537fe9697e1a598ac8a26f8ae43d8b286ac3f,
// dependencies/cmake/fpm/CMakeLists.txt:6-8): raw = value*256; compare/add/sub on raw;
// `*= integer` multiplies raw; static_cast<int64_t> = raw/256 (truncating);
// static_cast<double> = double(raw)/256.0.
#pragma once
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <iterator>
#include <map>
#include <set>
#include <unordered_map>

#include "../include/crane_gpu/node_select.h"

namespace ora {

using i64 = int64_t;
using u64 = uint64_t;
using u32 = uint32_t;

// ResourceView restricted to what reaches this path (PublicHeader.h:695-761):
// cpu, mem and the per-name GresCount {total, specified[type]} (PublicHeader.h:505-523).
// `type` is the dense class id of (name,type); class_name[] maps it back to its name.
struct ReqView {
  i64 cpu = 0;
  u64 mem = 0;
  uint8_t gtot[CNS_MAX_GRES_NAMES] = {0, 0, 0, 0};
  uint8_t gspec[CNS_MAX_GRES_CLASSES] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// req_node_res_view + req_task_res_view * n  (PublicHeader.cpp:473-481,601-611).
// req_task_res_view carries no GRES (CtldPublicDefs.cpp:1797-1824), so the GRES part is
// the node view's.
inline ReqView ComposeView(const ReqView& node, const ReqView& task, u32 n) {
  ReqView r = node;
  r.cpu = node.cpu + task.cpu * static_cast<i64>(n);
  r.mem = node.mem + task.mem * static_cast<u64>(n);
  return r;
}

struct GresLayout {
  u32 num_classes = 0;
  uint8_t class_name[CNS_MAX_GRES_CLASSES] = {};
  uint8_t class_shift[CNS_MAX_GRES_CLASSES] = {};
  uint8_t class_width[CNS_MAX_GRES_CLASSES] = {};
  u64 class_mask(u32 g) const {
    u64 w = class_width[g] >= 64 ? ~0ull : ((1ull << class_width[g]) - 1ull);
    return w << class_shift[g];
  }
  u64 name_mask(u32 a) const {
    u64 m = 0;
    for (u32 g = 0; g < num_classes; ++g)
      if (class_name[g] == a) m |= class_mask(g);
    return m;
  }
  int class_of_bit(int b) const {
    for (u32 g = 0; g < num_classes; ++g)
      if (b >= class_shift[g] && b < class_shift[g] + class_width[g]) return (int)g;
    return -1;
  }
};

// Canonical parity record of a ResourceInNodeV3.
struct MaskRes {
  i64 cpu = 0;
  u64 mem = 0;
  u64 clo = 0, chi = 0;     // core ids 0..63, 64..127
  u64 gres = 0;
  u64 c2 = 0, c3 = 0;       // core ids 128..191, 192..255 (ABI 3)
  bool operator==(const MaskRes& o) const {
    return cpu == o.cpu && mem == o.mem && clo == o.clo && chi == o.chi && c2 == o.c2 && c3 == o.c3 && gres == o.gres;
  }
  bool any_core() const { return (clo | chi | c2 | c3) != 0; }
  u64 core_word(int w) const { return w == 0 ? clo : w == 1 ? chi : w == 2 ? c2 : c3; }
  u64& core_word(int w) { return w == 0 ? clo : w == 1 ? chi : w == 2 ? c2 : c3; }
};

// ----------------------------------------------------------------------------------
// Mask algebra
// ----------------------------------------------------------------------------------
struct MaskAlgebra {
  using Res = MaskRes;
  const GresLayout* L;
  explicit MaskAlgebra(const GresLayout* l) : L(l) {}

  static int popc(u64 x) { return __builtin_popcountll(x); }
  static u64 lowest_n(u64 x, int n) {  // the n lowest set bits of x (all of x if fewer)
    u64 y = x;
    for (int i = 0; i < n && y; ++i) y &= y - 1;
    return x ^ y;
  }

  Res from_mask(const MaskRes& m) const { return m; }
  MaskRes to_mask(const Res& r) const { return r; }
  void set_zero(Res& r) const { r = Res{}; }
  // ResourceInNodeV3::IsZero, PublicHeader.cpp:798-801 (memory_sw is not modelled on this path)
  bool is_zero(const Res& r) const { return r.cpu == 0 && !r.any_core() && r.mem == 0 && r.gres == 0; }

  // ResourceView::GetFeasibleResourceInNode, PublicHeader.cpp:519-599
  bool feasible(const ReqView& q, const Res& a, Res* out) const {
    if (q.cpu > a.cpu) return false;  // :522
    if (q.mem > a.mem) return false;  // :523
    Res c;
    i64 req_int = q.cpu / 256;  // static_cast<int64_t>(cpu_t) :528
    bool is_int = (req_int * 256 == q.cpu) && a.any_core();  // :529-530
    if (is_int) {
      u32 n = static_cast<u32>(req_int);
      if (static_cast<u32>(popc(a.clo) + popc(a.chi) + popc(a.c2) + popc(a.c3)) < n) return false;  // :534
      int left = (int)n;   // the n lowest core ids (:536-541: the first n of an ordered set)
      for (int w = 0; w < 4 && left > 0; ++w) {
        c.core_word(w) = lowest_n(a.core_word(w), left);
        left -= popc(c.core_word(w));
      }
    }
    c.cpu = q.cpu;
    c.mem = q.mem;
    for (u32 name = 0; name < CNS_MAX_GRES_NAMES; ++name) {  // :549
      u64 spec_sum = 0;
      bool present = q.gtot[name] != 0;
      for (u32 g = 0; g < L->num_classes; ++g)
        if (L->class_name[g] == name && q.gspec[g]) {
          spec_sum += q.gspec[g];
          present = true;
        }
      if (!present) continue;
      u64 nm = L->name_mask(name);
      if ((a.gres & nm) == 0) return false;  // :550-551 name absent
      u64 untyped = q.gtot[name] > spec_sum ? q.gtot[name] - spec_sum : 0;  // :556-559
      for (u32 g = 0; g < L->num_classes; ++g) {  // specified types, canonical ascending :564
        if (L->class_name[g] != name || !q.gspec[g]) continue;
        u64 slots = a.gres & L->class_mask(g);
        if (slots == 0) return false;                    // :566 type absent
        if ((u64)popc(slots) < q.gspec[g]) return false;  // :569
        u64 take = lowest_n(slots, q.gspec[g]);
        u64 rest = slots ^ take;
        u64 extra = lowest_n(rest, (int)std::min<u64>(untyped, 64));  // :577-578
        untyped -= popc(extra);
        c.gres |= take | extra;
      }
      if (untyped > 0) {  // :582-592 other types, canonical ascending
        for (u32 g = 0; g < L->num_classes && untyped > 0; ++g) {
          if (L->class_name[g] != name || q.gspec[g]) continue;
          u64 slots = a.gres & L->class_mask(g);
          u64 extra = lowest_n(slots, (int)std::min<u64>(untyped, 64));
          untyped -= popc(extra);
          c.gres |= extra;
        }
      }
      if (untyped != 0) return false;  // :594
    }
    *out = c;
    return true;
  }

  // ResourceInNodeV3::Ckmin, PublicHeader.cpp:815-827
  void ckmin(Res& a, const Res& b) const {
    a.cpu = std::min(a.cpu, b.cpu);
    if (a.any_core() && b.any_core()) {
      a.clo &= b.clo;
      a.chi &= b.chi;
      a.c2 &= b.c2;
      a.c3 &= b.c3;
    }
    a.mem = std::min(a.mem, b.mem);
    a.gres &= b.gres;
  }
  // operator<=(ResourceInNodeV3, ResourceInNodeV3), PublicHeader.cpp:886-890,159-169,334-343
  bool le(const Res& a, const Res& b) const {
    if (a.cpu > b.cpu) return false;
    if (a.mem > b.mem) return false;
    return (a.gres & ~b.gres) == 0;
  }
  // operator+= / -=, PublicHeader.cpp:781-796,752-766,196-217,309-328
  void add(Res& a, const Res& b) const {
    a.clo |= b.clo;
    a.chi |= b.chi;
    a.c2 |= b.c2;
    a.c3 |= b.c3;
    a.cpu += b.cpu;
    a.mem += b.mem;
    a.gres |= b.gres;
  }
  void sub(Res& a, const Res& b) const {
    a.clo &= ~b.clo;  // tolerant erase :758-762
    a.chi &= ~b.chi;
    a.c2 &= ~b.c2;
    a.c3 &= ~b.c3;
    a.cpu -= b.cpu;
    a.mem -= b.mem;
    a.gres &= ~b.gres;
  }
};

// ----------------------------------------------------------------------------------
// Literal algebra — the reference's containers
// ----------------------------------------------------------------------------------
struct LitRes {
  i64 cpu = 0;                // CpuSet::cpu_count raw (PublicHeader.h:555-573)
  std::set<u32> cores;        // CpuSet::core_ids
  u64 mem = 0;
  // DedicatedResourceInNode::name_type_slots_map (PublicHeader.h:493): name -> type -> slots.
  // The type level is a std::map (reference: unordered_map) so that the
  // "iteration order unspecified" spots of GetFeasibleResourceInNode
  // (PublicHeader.cpp:564,583) get the canonical ascending-type order.
  std::unordered_map<int, std::map<int, std::set<int>>> gres;
};

struct LitAlgebra {
  using Res = LitRes;
  const GresLayout* L;
  explicit LitAlgebra(const GresLayout* l) : L(l) {}

  Res from_mask(const MaskRes& m) const {
    Res r;
    r.cpu = m.cpu;
    r.mem = m.mem;
    for (int b = 0; b < 64; ++b) {
      if ((m.clo >> b) & 1) r.cores.insert((u32)b);
      if ((m.chi >> b) & 1) r.cores.insert((u32)(64 + b));
      if ((m.c2 >> b) & 1) r.cores.insert((u32)(128 + b));
      if ((m.c3 >> b) & 1) r.cores.insert((u32)(192 + b));
      if ((m.gres >> b) & 1) {
        int g = L->class_of_bit(b);
        assert(g >= 0);
        r.gres[L->class_name[g]][g].insert(b);
      }
    }
    return r;
  }
  MaskRes to_mask(const Res& r) const {
    MaskRes m;
    m.cpu = r.cpu;
    m.mem = r.mem;
    for (u32 c : r.cores) {
      m.core_word((int)(c >> 6)) |= 1ull << (c & 63);
    }
    for (const auto& [name, tm] : r.gres)
      for (const auto& [type, slots] : tm)
        for (int s : slots) m.gres |= 1ull << s;
    return m;
  }
  void set_zero(Res& r) const {
    r.cpu = 0;
    r.cores.clear();
    r.mem = 0;
    r.gres.clear();
  }
  bool is_zero(const Res& r) const { return r.cpu == 0 && r.cores.empty() && r.mem == 0 && r.gres.empty(); }

  // ResourceView::GetFeasibleResourceInNode, PublicHeader.cpp:519-599
  bool feasible(const ReqView& q, const Res& avail, Res* out) const {
    if (q.cpu > avail.cpu) return false;
    if (q.mem > avail.mem) return false;
    Res cand;
    i64 req_int = q.cpu / 256;
    bool is_integer_req = (req_int * 256 == q.cpu) && !avail.cores.empty();
    if (is_integer_req) {
      u32 n = static_cast<u32>(req_int);
      if (avail.cores.size() < n) return false;
      auto it = avail.cores.begin();
      for (u32 i = 0; i < n; ++i, ++it) cand.cores.insert(*it);
      cand.cpu = q.cpu;
    } else {
      cand.cpu = q.cpu;
    }
    cand.mem = q.mem;

    // m_gres_map_: name -> GresCount{total, specified}
    for (int name = 0; name < (int)CNS_MAX_GRES_NAMES; ++name) {
      std::map<int, u64> specified;
      for (u32 g = 0; g < L->num_classes; ++g)
        if (L->class_name[g] == name && q.gspec[g]) specified[(int)g] = q.gspec[g];
      u64 total = q.gtot[name];
      if (total == 0 && specified.empty()) continue;  // name not in the request map

      auto dres_avail_it = avail.gres.find(name);
      if (dres_avail_it == avail.gres.end()) return false;
      const auto& dres_avail = dres_avail_it->second;

      u64 specified_sum = 0;
      for (const auto& [_, cnt] : specified) specified_sum += cnt;
      u64 untyped_cnt = total > specified_sum ? total - specified_sum : 0;

      auto& feasible_res_dev_name = cand.gres[name];
      for (const auto& [dev_type, typed_cnt] : specified) {
        auto avail_slots_it = dres_avail.find(dev_type);
        if (avail_slots_it == dres_avail.end()) return false;
        const auto& avail_slots = avail_slots_it->second;
        if (avail_slots.size() < typed_cnt) return false;
        auto& feasible_res_dev_name_type = feasible_res_dev_name[dev_type];
        auto it = avail_slots.begin();
        for (size_t i = 0; i < typed_cnt; ++i, ++it) feasible_res_dev_name_type.emplace(*it);
        for (; untyped_cnt > 0 && it != avail_slots.end(); ++it, --untyped_cnt)
          feasible_res_dev_name_type.emplace(*it);
      }
      if (untyped_cnt > 0) {
        for (const auto& [type, slots] : dres_avail) {
          if (specified.count(type)) continue;
          auto it = slots.begin();
          for (; untyped_cnt > 0 && it != slots.end(); ++it, --untyped_cnt)
            feasible_res_dev_name[type].emplace(*it);
          if (untyped_cnt == 0) break;
        }
      }
      if (untyped_cnt != 0) return false;
    }
    *out = std::move(cand);
    return true;
  }

  // PublicHeader.cpp:815-827 with Intersection :176-190,345-359
  void ckmin(Res& a, const Res& b) const {
    a.cpu = std::min(a.cpu, b.cpu);
    if (!a.cores.empty() && !b.cores.empty()) {
      std::set<u32> inter;
      std::set_intersection(a.cores.begin(), a.cores.end(), b.cores.begin(), b.cores.end(),
                            std::inserter(inter, inter.begin()));
      a.cores = std::move(inter);
    }
    a.mem = std::min(a.mem, b.mem);
    std::unordered_map<int, std::map<int, std::set<int>>> result;
    for (const auto& [lname, ltm] : a.gres) {
      auto rit = b.gres.find(lname);
      if (rit == b.gres.end()) continue;
      std::map<int, std::set<int>> tm;
      for (const auto& [ltype, lslots] : ltm) {
        auto rt = rit->second.find(ltype);
        if (rt == rit->second.end()) continue;
        std::set<int> temp;
        std::set_intersection(lslots.begin(), lslots.end(), rt->second.begin(),
                              rt->second.end(), std::inserter(temp, temp.begin()));
        if (!temp.empty()) tm[ltype] = std::move(temp);
      }
      if (!tm.empty()) result[lname] = std::move(tm);
    }
    a.gres = std::move(result);
  }

  // PublicHeader.cpp:886-890,159-169,334-343
  bool le(const Res& a, const Res& b) const {
    if (a.cpu > b.cpu) return false;
    if (a.mem > b.mem) return false;
    for (const auto& [lname, ltm] : a.gres) {
      auto rit = b.gres.find(lname);
      if (rit == b.gres.end()) return false;
      for (const auto& [ltype, lslots] : ltm) {
        auto rt = rit->second.find(ltype);
        if (rt == rit->second.end()) return false;
        if (!std::includes(rt->second.begin(), rt->second.end(), lslots.begin(), lslots.end()))
          return false;
      }
    }
    return true;
  }

  // PublicHeader.cpp:781-787,752-756,196-202,309-314
  void add(Res& a, const Res& b) const {
    a.cores.insert(b.cores.begin(), b.cores.end());
    a.cpu += b.cpu;
    a.mem += b.mem;
    for (const auto& [name, tm] : b.gres)
      for (const auto& [type, slots] : tm) a.gres[name][type].insert(slots.begin(), slots.end());
  }
  // PublicHeader.cpp:789-796,758-766,204-217,316-328.  The reference's `.at()` on a
  // missing type throws; every caller on this path subtracts a subset, so the oracle
  // treats a missing name/type as "nothing to erase".
  void sub(Res& a, const Res& b) const {
    for (u32 id : b.cores) a.cores.erase(id);
    a.cpu -= b.cpu;
    a.mem -= b.mem;
    for (const auto& [name, tm] : b.gres) {
      auto it = a.gres.find(name);
      if (it == a.gres.end()) continue;
      for (const auto& [type, slots] : tm) {
        auto tt = it->second.find(type);
        if (tt == it->second.end()) continue;
        std::set<int> temp;
        std::set_difference(tt->second.begin(), tt->second.end(), slots.begin(), slots.end(),
                            std::inserter(temp, temp.begin()));
        if (temp.empty()) it->second.erase(tt);
        else tt->second = std::move(temp);
      }
      if (it->second.empty()) a.gres.erase(it);
    }
  }
};

}  // namespace ora
