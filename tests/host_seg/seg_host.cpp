// The two forms of TryPreempt_'s segment tree — cranesched_amd/csrc/preempt_dev.inc, the DEVICE source, compiled for the
// host (CNS_PRE_HOST: "LDS" is a static array, a wave is one thread, lane 0) — run on the same random operation sequences:
// the reference's recursion written out here (RTree), the device's node-for-node tree (LDS write-back cache over a pool in
// memory) and its compressed form must report the same `satisfied` after every operation.  Test infrastructure (tests/test_seg_host.py builds and runs it); nothing here ships.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../cranesched_amd/csrc/engine_params.h"

#define __device__
#define __forceinline__ inline
#define __noinline__
#define CNS_PRE_HOST 1
struct { unsigned x; } threadIdx = {0};

namespace cns {
static inline u32 uni32(u32 v) { return v; }
static inline void drain_stores() {}
template <class T> static inline const T* as_global(const T* p) { return p; }
static inline Res res_zero() { Res r; r.cpu = 0; r.mem = 0; cores_clear(r); r.gres = 0; return r; }
#include "../../cranesched_amd/csrc/preempt_dev.inc"
}  // namespace cns

using namespace cns;

// ---- the reference's tree as the reference writes it: recursion, nodes on the heap (JobScheduler.h:867-980) ---------------------
struct RNode { i64 st, ed; RNode *ls = nullptr, *rs = nullptr; bool sat = false; Res res, add, sub; };
struct RTree {
  Res target;
  std::vector<RNode*> all;
  RNode* root;
  RNode* node(i64 st, i64 ed, bool sat, const Res& res) {
    RNode* n = new RNode; n->st = st; n->ed = ed; n->sat = sat; n->res = res; n->add = res_zero(); n->sub = res_zero(); all.push_back(n); return n;
  }
  RTree(i64 st, i64 ed, const Res& t) : target(t) { root = node(st, ed, false, res_zero()); }
  ~RTree() { for (RNode* n : all) delete n; }
  void apply(RNode* n, const Res& r, bool plus) {
    if (plus) res_add(n->res, r); else res_sub(n->res, r);
    n->sat = res_le(target, n->res);
    if (n->ls) res_add(plus ? n->add : n->sub, r);
  }
  void down(RNode* n) {
    if (!n->ls) {
      const i64 mid = n->st + (n->ed - n->st) / 2;
      n->ls = node(n->st, mid, n->sat, n->res); n->rs = node(mid, n->ed, n->sat, n->res);
      return;
    }
    if (!pre_is_zero(n->add)) { apply(n->ls, n->add, true); apply(n->rs, n->add, true); n->add = res_zero(); }
    if (!pre_is_zero(n->sub)) { apply(n->ls, n->sub, false); apply(n->rs, n->sub, false); n->sub = res_zero(); }
  }
  void walk(RNode* n, i64 st, i64 ed, const Res& r, bool plus) {
    if (n->ed <= st || ed <= n->st) return;
    if (st <= n->st && n->ed <= ed) { apply(n, r, plus); return; }
    down(n);
    walk(n->ls, st, ed, r, plus); walk(n->rs, st, ed, r, plus);
    n->sat = n->ls->sat && n->rs->sat;
  }
};

static uint64_t rng_state = 1;
static uint64_t rnd() {
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static Res rand_res(int scale) {
  Res r = res_zero();
  if (rnd() % 20 == 0) return r;
  r.cpu = (i64)(rnd() % scale) * 256;
  r.mem = (rnd() % scale) << 30;
  if (rnd() & 1) r.clo = rnd() & 0x3F;
  if (rnd() % 5 < 2) r.gres = rnd() & 0xF;
  return r;
}

// one case; returns 0 if both forms agree after every operation.  cap: records the compressed form may use (0: all)
static int run_case(uint64_t seed, i64 L, i64 unit, int n_ends, int n_ops, u32 cap, long* visits_lit, long* visits_cmp, long* fell_back) {
  rng_state = seed * 2654435761ull + 12345;
  const i64 seg_end = L * unit;
  std::vector<i64> pool_ends = {0, seg_end};
  for (int i = 0; i < n_ends; ++i) pool_ends.push_back(((i64)(rnd() % (uint64_t)(L + 7)) - 3) * unit);
  Res target = res_zero();
  target.cpu = (i64)(1 + rnd() % 4) * 256; target.mem = (rnd() % 4) << 30; if (rnd() & 1) target.gres = rnd() & 7;

  static std::vector<PreNode> pool(1u << 16);
  static u32 lit_lds[kPreCacheDw], cmp_lds[kPreCacheDw];
  PreTree T;
  T.pool = pool.data(); T.cap = (u32)pool.size(); T.used = 0; T.overflow = false;
  T.cdat = lit_lds; T.ctag = T.cdat + kPreCacheN * kPreNodeDw;
  for (u32 x = 0; x < kPreCacheN; ++x) T.ctag[x] = 0;
  pre_new(T, 0, seg_end, 0u, res_zero(), 0);
  PreC C;
  C.base = cmp_lds; C.stack = C.base + kPcCap * kPcDw; C.used = 0; C.overflow = false; C.cap = cap ? cap : kPcCap;
  pc_leaf(C, 0, seg_end, res_zero(), 0u, 0);
  RTree RT(0, seg_end, target);

  struct Op { i64 a, b; Res r; };
  std::vector<Op> done;
  for (int i = 0; i < n_ops; ++i) {
    Op o; bool plus;
    if (!done.empty() && rnd() % 10 < 3) { const size_t k = rnd() % done.size(); o = done[k]; done.erase(done.begin() + (long)k); plus = false; }
    else {
      i64 a = pool_ends[rnd() % pool_ends.size()], b = pool_ends[rnd() % pool_ends.size()];
      if (a == b) continue;
      if (a > b) std::swap(a, b);
      o.a = a; o.b = b; o.r = rand_res(4); plus = rnd() % 100 < 85;
      if (plus) done.push_back(o);
    }
    if (getenv("SEG_DUMP")) printf("OP %lld %lld %d %lld %llu %llu %llu | target %lld %llu %llu seg_end %lld\n", (long long)o.a, (long long)o.b, plus ? 1 : 0, (long long)o.r.cpu, (unsigned long long)o.r.mem, (unsigned long long)o.r.clo, (unsigned long long)o.r.gres, (long long)target.cpu, (unsigned long long)target.mem, (unsigned long long)target.gres, (long long)seg_end);
    RT.walk(RT.root, o.a, o.b, o.r, plus);
    pre_range(T, 0, target, o.a, o.b, o.r, plus, 0);
    if (T.overflow) return 2;
    if ((pre_sat(T, 0, 0) != 0) != RT.root->sat) {
      fprintf(stderr, "seed %llu op %d: the device's node-for-node tree %u, the reference's recursion %d\n", (unsigned long long)seed, i, pre_sat(T, 0, 0), (int)RT.root->sat);
      return 3;
    }
    if (!C.overflow) C = pre_crange_v(C, 0, target, o.a, o.b, o.r, plus);
    if (C.overflow) { ++*fell_back; continue; }     // (on the device the whole call starts again node for node)
    const u32 sl = pre_sat(T, 0, 0), sc = pre_r32(pc_at(C, 0) + kPcSat);
    if ((sl != 0) != (sc != 0)) {
      fprintf(stderr, "seed %llu op %d: [%lld, %lld) %c: node for node %u, compressed %u\n", (unsigned long long)seed, i, (long long)o.a, (long long)o.b, plus ? '+' : '-', sl, sc);
      return 1;
    }
  }
  return 0;
}

int main(int argc, char** argv) {
  const int cases = argc > 1 ? atoi(argv[1]) : 2000;
  static const i64 Ls[] = {1, 2, 3, 5, 8, 16, 37, 64, 100, 255, 256, 257, 1000, 3600, 86400, 1 << 20};
  static const i64 units[] = {1, 1, 2, 3, 4000000000ll, 4000000000ll, 1ll << 32, 1000003};
  static const int ends[] = {2, 3, 5, 9, 20, 60}, ops[] = {10, 40, 150};
  long vl = 0, vc = 0, fb = 0, ran = 0;
  const int only = getenv("SEG_ONLY") ? atoi(getenv("SEG_ONLY")) : -1;
  for (int c = 0; c < cases; ++c) {
    if (only >= 0 && c != only) continue;
    rng_state = (uint64_t)c * 7 + 1;
    const i64 L = Ls[rnd() % 16], unit = units[rnd() % 8];
    const int ne = ends[rnd() % 6], no = ops[rnd() % 3];
    const u32 cap = c % 5 == 4 ? 12u : 0u;     // every fifth case: so few records that the compressed form runs out
    const int rc = run_case((uint64_t)c + 1000000, L, unit, ne, no, cap, &vl, &vc, &fb);
    if (rc) { printf("FAIL case %d rc %d\n", c, rc); return 1; }
    ++ran;
  }
  printf("ok: %ld cases, compressed form out of records in %ld operations (counted, not compared)\n", ran, fb);
  return 0;
}
