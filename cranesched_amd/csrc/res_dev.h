// Resource algebra of the node-selection engine on bit masks, usable from HIP device code
// (gfx950) and — for the CPU unit tests of these helpers — from plain host C++.
//
// Reference semantics implemented (paths relative to the CraneSched tree):
//   ResourceView::GetFeasibleResourceInNode   src/Utilities/PublicHeader/PublicHeader.cpp:519-599
//   ResourceInNodeV3::Ckmin                   src/Utilities/PublicHeader/PublicHeader.cpp:815-827
//   operator<=(ResourceInNodeV3, ...)         src/Utilities/PublicHeader/PublicHeader.cpp:886-890
//   ResourceInNodeV3 += / -=                  src/Utilities/PublicHeader/PublicHeader.cpp:781-796
//   get_max_tasks lambda                      src/CraneCtld/JobScheduler.cpp:6171-6186
// in the canonical integer model of include/crane_gpu/node_select.h.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define CNS_HD __host__ __device__ __forceinline__
#else
#define CNS_HD inline
#endif

namespace cns {

typedef int64_t i64;
typedef uint64_t u64;
typedef uint32_t u32;

constexpr int kMaxClasses = 8;
constexpr int kMaxNames = 4;

// ResourceInNodeV3 in mask form: cpu raw (x256), mem bytes, 256 core bits (ids 0..63, 64..127, 128..191, 192..255:
// CpuSet::core_ids is an unbounded set, PublicHeader.h:555-573; 256 ids are what the engine carries), 64 GRES slot bits.
struct Res {
  i64 cpu;
  u64 mem;
  u64 clo, chi;
  u64 gres;
  u64 c2, c3;
};

// One entry of a node's time -> available-resource map (std::map<absl::Time, ResourceInNodeV3>,
// src/CraneCtld/JobScheduler.h:245) — the REGISTER form.
struct TlEntry {
  i64 t;
  Res r;
};
// ... and how it sits in HBM: a sorted array of 48-byte records (time, cpu, mem, core ids 0..127, GRES) plus, behind it, a
// parallel array of 16-byte records with core ids 128..255 that is read and written ONLY when the snapshot has a node with
// such ids (TlMap::wide): clusters without them move exactly the bytes they moved before ABI 3.
struct alignas(16) TlMem {
  i64 t;
  i64 cpu;
  u64 mem;
  u64 clo, chi;
  u64 gres;
};
struct alignas(16) TlExt {
  u64 c2, c3;
};
struct TlSlot {
  TlMem* m;
  TlExt* x;
  u32 wide;
  CNS_HD TlEntry get() const {
    const TlMem v = *m;
    TlEntry e;
    e.t = v.t; e.r.cpu = v.cpu; e.r.mem = v.mem; e.r.clo = v.clo; e.r.chi = v.chi; e.r.gres = v.gres;
    e.r.c2 = 0; e.r.c3 = 0;
    if (wide) { const TlExt w = *x; e.r.c2 = w.c2; e.r.c3 = w.c3; }
    return e;
  }
  CNS_HD operator TlEntry() const { return get(); }
  CNS_HD i64 t() const { return m->t; }
  CNS_HD Res r() const { return get().r; }
  CNS_HD void operator=(const TlEntry& e) const {
    TlMem v;
    v.t = e.t; v.cpu = e.r.cpu; v.mem = e.r.mem; v.clo = e.r.clo; v.chi = e.r.chi; v.gres = e.r.gres;
    *m = v;
    if (wide) { TlExt w; w.c2 = e.r.c2; w.c3 = e.r.c3; *x = w; }
  }
  CNS_HD void operator=(const TlSlot& o) const { *this = o.get(); }
  CNS_HD void set_t(i64 t) const { m->t = t; }
  CNS_HD void set_r(const Res& r) const { TlEntry e; e.t = m->t; e.r = r; *this = e; }
};
struct TlMap {
  TlMem* m;
  TlExt* x;
  u32 wide;
  CNS_HD TlSlot operator[](u32 i) const { return TlSlot{m + i, x + i, wide}; }
  CNS_HD TlMap operator+(u32 k) const { return TlMap{m + k, x + k, wide}; }
};

// ResourceView of a request: cpu raw, mem, per-name GresCount.total (4 x u8) and per-class
// GresCount.specified (8 x u8), PublicHeader.h:505-523,695-761.
struct Req {
  i64 cpu;
  u64 mem;
  u32 gtot;   // byte a = total of name a
  u64 gspec;  // byte g = specified count of class g
};

// Device copy of cns_gres_layout with the masks precomputed.
struct GresDev {
  u32 num_classes;
  u32 class_name_packed;  // nibble g = name id of class g
  u64 class_mask[kMaxClasses];
  u64 name_mask[kMaxNames];
  u64 name_bytes[kMaxNames];  // 0xFF in byte g for every class g of the name
};

CNS_HD int popc64(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popcll(x);
#else
  return __builtin_popcountll(x);
#endif
}

// The n lowest set bits of x (all of x if it has fewer).
CNS_HD u64 lowest_n(u64 x, int n) {
  u64 y = x;
  for (int i = 0; i < n && y; ++i) y &= y - 1;
  return x ^ y;
}

CNS_HD u32 byte_of(u64 v, int i) { return (u32)((v >> (8 * i)) & 0xFF); }
CNS_HD bool cores_empty(const Res& r) { return (r.clo | r.chi | r.c2 | r.c3) == 0; }
CNS_HD u32 cores_count(const Res& r) { return (u32)(popc64(r.clo) + popc64(r.chi) + popc64(r.c2) + popc64(r.c3)); }
CNS_HD void cores_clear(Res& r) { r.clo = 0; r.chi = 0; r.c2 = 0; r.c3 = 0; }
CNS_HD void cores_fill(Res& r) { r.clo = ~0ull; r.chi = ~0ull; r.c2 = ~0ull; r.c3 = ~0ull; }
CNS_HD void cores_copy(Res& r, const Res& s) { r.clo = s.clo; r.chi = s.chi; r.c2 = s.c2; r.c3 = s.c3; }
CNS_HD void cores_and(Res& r, const Res& s) { r.clo &= s.clo; r.chi &= s.chi; r.c2 &= s.c2; r.c3 &= s.c3; }

CNS_HD void res_sub(Res& a, const Res& b) {  // PublicHeader.cpp:789-796,758-766 (tolerant core erase)
  a.clo &= ~b.clo;
  a.chi &= ~b.chi;
  a.c2 &= ~b.c2;
  a.c3 &= ~b.c3;
  a.cpu -= b.cpu;
  a.mem -= b.mem;
  a.gres &= ~b.gres;
}
CNS_HD void res_add(Res& a, const Res& b) {  // PublicHeader.cpp:781-787
  a.clo |= b.clo;
  a.chi |= b.chi;
  a.c2 |= b.c2;
  a.c3 |= b.c3;
  a.cpu += b.cpu;
  a.mem += b.mem;
  a.gres |= b.gres;
}
CNS_HD bool res_le(const Res& a, const Res& b) {  // PublicHeader.cpp:886-890 (core ids not compared)
  return a.cpu <= b.cpu && a.mem <= b.mem && (a.gres & ~b.gres) == 0;
}
CNS_HD void res_ckmin(Res& a, const Res& b) {  // PublicHeader.cpp:815-827
  a.cpu = a.cpu < b.cpu ? a.cpu : b.cpu;
  if (!cores_empty(a) && !cores_empty(b)) cores_and(a, b);
  a.mem = a.mem < b.mem ? a.mem : b.mem;
  a.gres &= b.gres;
}

// req_node_res_view + req_task_res_view * n (PublicHeader.cpp:473-481,601-611); the task view has no GRES.
CNS_HD Req compose(const Req& node, i64 task_cpu, u64 task_mem, u32 n) {
  Req r = node;
  r.cpu = node.cpu + task_cpu * (i64)n;
  r.mem = node.mem + task_mem * (u64)n;
  return r;
}

// GetFeasibleResourceInNode, PublicHeader.cpp:519-599.
CNS_HD bool feasible(const Req& q, const Res& a, Res& out, const GresDev& L) {
  if (q.cpu > a.cpu) return false;  // :522
  if (q.mem > a.mem) return false;  // :523
  Res c;
  c.cpu = q.cpu;
  c.mem = q.mem;
  cores_clear(c);
  c.gres = 0;
  i64 req_int = q.cpu / 256;                                        // :528
  bool is_int = (req_int * 256 == q.cpu) && !cores_empty(a);        // :529-530
  if (is_int) {
    if (cores_count(a) < (u32)req_int) return false;                // :534
    int left = (int)req_int;                                        // the n lowest core ids (:536-541)
    c.clo = lowest_n(a.clo, left); left -= popc64(c.clo);
    if (left > 0) { c.chi = lowest_n(a.chi, left); left -= popc64(c.chi); }
    if (left > 0) { c.c2 = lowest_n(a.c2, left); left -= popc64(c.c2); }
    if (left > 0) c.c3 = lowest_n(a.c3, left);
  }
  if (q.gtot | q.gspec) {
    for (int name = 0; name < kMaxNames; ++name) {                  // :549
      u32 tot = (q.gtot >> (8 * name)) & 0xFF;
      u64 spec_b = q.gspec & L.name_bytes[name];
      if (tot == 0 && spec_b == 0) continue;
      if ((a.gres & L.name_mask[name]) == 0) return false;          // :550-551
      u32 spec_sum = 0;
      for (int g = 0; g < kMaxClasses; ++g) spec_sum += byte_of(spec_b, g);
      u32 untyped = tot > spec_sum ? tot - spec_sum : 0;             // :556-559
      for (int g = 0; g < (int)L.num_classes; ++g) {                // specified types, ascending :564
        u32 cnt = byte_of(spec_b, g);
        if (!cnt) continue;
        u64 slots = a.gres & L.class_mask[g];
        if (slots == 0) return false;                                // :566
        if ((u32)popc64(slots) < cnt) return false;                  // :569
        u64 take = lowest_n(slots, (int)cnt);
        u64 rest = slots ^ take;
        u64 extra = lowest_n(rest, (int)(untyped < 64 ? untyped : 64));  // :577-578
        untyped -= (u32)popc64(extra);
        c.gres |= take | extra;
      }
      if (untyped > 0) {                                             // :582-592 other types ascending
        for (int g = 0; g < (int)L.num_classes && untyped > 0; ++g) {
          if (((L.class_name_packed >> (4 * g)) & 0xF) != (u32)name) continue;
          if (byte_of(spec_b, g)) continue;
          u64 slots = a.gres & L.class_mask[g];
          u64 extra = lowest_n(slots, (int)(untyped < 64 ? untyped : 64));
          untyped -= (u32)popc64(extra);
          c.gres |= extra;
        }
      }
      if (untyped != 0) return false;                                // :594
    }
  }
  out = c;
  return true;
}

CNS_HD u32 byte_sum(u64 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_sad_u8((u32)v, 0u, __builtin_amdgcn_sad_u8((u32)(v >> 32), 0u, 0u));
#else
  u32 s = 0;
  for (int i = 0; i < 8; ++i) s += (u32)((v >> (8 * i)) & 0xFF);
  return s;
#endif
}

// Truth value of feasible() from COUNTS only: cpu, mem, number of free cores and the per-class slot
// popcounts (byte g of `cnt`).  GetFeasibleResourceInNode fails exactly when (PublicHeader.cpp:522-594)
//   cpu or mem is short, a whole-number request finds fewer free core ids than it needs, a specified
//   type has fewer slots than asked, or the name's slots cannot cover max(total, sum(specified)).
CNS_HD bool feasible_counts(const Req& q, i64 cpu, u64 mem, u32 ncores, u64 cnt, const GresDev& L) {
  if (q.cpu > cpu) return false;
  if (q.mem > mem) return false;
  const i64 req_int = q.cpu / 256;
  if (req_int * 256 == q.cpu && ncores != 0 && ncores < (u32)req_int) return false;
  if (q.gtot | q.gspec) {
    const u64 H8 = 0x8080808080808080ull;
    if (q.gspec & H8) return false;  // a class holds <= 64 slots
    if ((((cnt | H8) - q.gspec) & H8) != H8) return false;  // bytewise spec_g <= cnt_g
    for (int name = 0; name < kMaxNames; ++name) {
      const u32 tot = (q.gtot >> (8 * name)) & 0xFF;
      const u64 spec_b = q.gspec & L.name_bytes[name];
      if (tot == 0 && spec_b == 0) continue;
      const u32 have = byte_sum(cnt & L.name_bytes[name]);
      const u32 ssum = byte_sum(spec_b);
      if (have < (tot > ssum ? tot : ssum)) return false;
    }
  }
  return true;
}

// get_max_tasks, JobScheduler.cpp:6171-6186: 0 if the minimum view does not fit, else tpn_min plus
// the number of further single tasks that fit one at a time, capped at tpn_max.
CNS_HD int max_tasks(const Req& min_view, i64 task_cpu, u64 task_mem, u32 tpn_min, u32 tpn_max,
                     const Res& on_node, const GresDev& L) {
  Res f;
  if (!feasible(min_view, on_node, f, L)) return 0;
  int n = (int)tpn_min;
  if (n >= (int)tpn_max) return n;
  Res left = on_node;
  res_sub(left, f);
  Req task;
  task.cpu = task_cpu;
  task.mem = task_mem;
  task.gtot = 0;
  task.gspec = 0;
  while (n < (int)tpn_max && feasible(task, left, f, L)) {
    ++n;
    res_sub(left, f);
  }
  return n;
}

}  // namespace cns
