// CPU unit test of the engine's host-compilable device helpers (cranesched_amd/csrc/res_dev.h,
// pq_emul.h) against the oracle's MaskAlgebra and the real std::priority_queue.
// Build: g++ -O1 -std=c++20 tests/cpp/test_dev_helpers.cpp -o tests/cpp/test_dev_helpers
#include <cstdio>
#include <cstdlib>
#include <queue>
#include <random>
#include <vector>

#include "../../cranesched_amd/csrc/pq_emul.h"
#include "../../oracle/res_algebra.hpp"

using namespace cns;

static GresDev make_dev(const ora::GresLayout& L) {
  GresDev d{};
  d.num_classes = L.num_classes;
  for (u32 c = 0; c < L.num_classes; ++c) {
    d.class_mask[c] = L.class_mask(c);
    d.class_name_packed |= (u32)L.class_name[c] << (4 * c);
    d.name_mask[L.class_name[c]] |= L.class_mask(c);
    d.name_bytes[L.class_name[c]] |= 0xFFull << (8 * c);
  }
  return d;
}

struct node_info {  // JobScheduler.cpp:6157-6164
  int ntasks_on_node;
  int id;
  bool operator<(const node_info& o) const { return ntasks_on_node > o.ntasks_on_node; }
};

int main() {
  std::mt19937_64 rng(12345);
  ora::GresLayout L;
  L.num_classes = 3;
  L.class_name[0] = 0; L.class_shift[0] = 0; L.class_width[0] = 4;
  L.class_name[1] = 0; L.class_shift[1] = 4; L.class_width[1] = 4;
  L.class_name[2] = 1; L.class_shift[2] = 8; L.class_width[2] = 8;
  GresDev D = make_dev(L);
  ora::MaskAlgebra A(&L);
  long nfeas = 0, nfail = 0;
  for (int it = 0; it < 400000; ++it) {
    Res a;
    a.cpu = (i64)(rng() % 40) * 128;
    a.mem = rng() % 64;
    a.clo = (rng() % 4 == 0) ? 0 : (rng() & rng() & 0xFFFF);
    a.chi = (rng() % 8 == 0) ? (rng() & 0xF) : 0;
    a.c2 = (rng() % 8 == 0) ? (rng() & 0x3F) : 0;      // core ids 128..255 (ABI 3)
    a.c3 = (rng() % 16 == 0) ? (rng() & 0x7) : 0;
    a.gres = rng() & rng() & 0xFFFF;
    Req q;
    q.cpu = (i64)(rng() % 12) * 128;
    q.mem = rng() % 48;
    q.gtot = 0; q.gspec = 0;
    if (rng() % 2) {
      q.gtot = (u32)(rng() % 5) | ((u32)(rng() % 6) << 8);
      if (rng() % 2) q.gspec = (rng() % 3) | ((rng() % 3) << 8) | ((rng() % 4) << 16);
    }
    ora::ReqView v;
    v.cpu = q.cpu; v.mem = q.mem;
    for (int i = 0; i < 4; ++i) v.gtot[i] = (q.gtot >> (8 * i)) & 0xFF;
    for (int i = 0; i < 8; ++i) v.gspec[i] = (q.gspec >> (8 * i)) & 0xFF;
    ora::MaskRes am; am.cpu = a.cpu; am.mem = a.mem; am.clo = a.clo; am.chi = a.chi; am.gres = a.gres; am.c2 = a.c2; am.c3 = a.c3;
    ora::MaskRes om;
    Res od;
    bool r1 = A.feasible(v, am, &om);
    bool r2 = feasible(q, a, od, D);
    u64 cnt = 0;
    for (int g = 0; g < 3; ++g) cnt |= (u64)__builtin_popcountll(a.gres & D.class_mask[g]) << (8 * g);
    bool r3 = feasible_counts(q, a.cpu, a.mem, cores_count(a), cnt, D);
    if (r1 != r2 || r1 != r3) { printf("FAIL feasible truth it=%d %d %d %d\n", it, r1, r2, r3); return 1; }
    if (r1 && !(om.cpu == od.cpu && om.mem == od.mem && om.clo == od.clo && om.chi == od.chi && om.c2 == od.c2 && om.c3 == od.c3 && om.gres == od.gres)) {
      printf("FAIL feasible alloc it=%d\n", it);
      return 1;
    }
    (r1 ? nfeas : nfail)++;
  }
  // priority_queue emulation: random push / pop-when-over-k sequences with many ties
  for (int it = 0; it < 20000; ++it) {
    int k = 1 + (int)(rng() % 9);
    std::priority_queue<node_info> pq;
    std::vector<HeapEnt> H(k + 2);
    int hs = 0;
    int n = 1 + (int)(rng() % 40);
    for (int i = 0; i < n; ++i) {
      int cap = 1 + (int)(rng() % 3);
      pq.push(node_info{cap, i});
      H[hs] = HeapEnt{}; H[hs].ntasks = cap; H[hs].node = (u32)i; ++hs; pq_push(H.data(), hs);
      if ((int)pq.size() > k) {
        if (pq.top().id != (int)H[0].node) { printf("FAIL pq top it=%d\n", it); return 1; }
        pq.pop();
        pq_pop(H.data(), hs); --hs;
      }
    }
    while (!pq.empty()) {
      if (pq.top().id != (int)H[0].node || pq.top().ntasks_on_node != H[0].ntasks) { printf("FAIL pq drain it=%d\n", it); return 1; }
      pq.pop();
      pq_pop(H.data(), hs); --hs;
    }
  }
  printf("ok feasible=%ld infeasible=%ld\n", nfeas, nfail);
  return 0;
}
